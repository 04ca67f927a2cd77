"""Same-process A/B of engine.stream_bf16_tokens (the streamed block kernel hands patch_unembed bf16 tokens) on BASELINE configs[1]:
per-stage event times and the whole forward, alternating settings.   python scripts/ab_bf16_tokens.py"""
import contextlib, importlib, os, sys
from collections import defaultdict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import engine
from transformerupscaler_amd.weights import deterministic_state_dict

m = importlib.import_module("models.FastTransformer.model").TransformerModel()
m.load_state_dict(deterministic_state_dict(0), strict=False)
m = m.cuda().eval()
x = torch.rand(8, 3, 720, 1280).cuda()
ev = defaultdict(list)

@contextlib.contextmanager
def timer(name):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); yield; e.record()
    ev[(engine.stream_bf16_tokens, name)].append((s, e))

outs = {}
with torch.no_grad():
    for flag in (False, True):
        engine.stream_bf16_tokens = flag
        for _ in range(3):
            outs[flag] = m(x, res_out=(1080, 1920))
    torch.cuda.synchronize()
    print("outputs identical:", bool(torch.equal(outs[False], outs[True])))
    engine.stage_timer = timer
    tot = defaultdict(list)
    for rnd in range(6):
        for flag in (False, True):
            engine.stream_bf16_tokens = flag
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                m(x, res_out=(1080, 1920))
            t1.record()
            torch.cuda.synchronize()
            tot[flag].append(t0.elapsed_time(t1) / 5)
for flag in (False, True):
    t = sorted(tot[flag])
    print(f"bf16 tokens {flag}: forward median {t[len(t) // 2]:.3f} ms (min {t[0]:.3f}); blocks "
          f"{sum(s.elapsed_time(e) for s, e in ev[(flag, 'blocks')]) / len(ev[(flag, 'blocks')]):.3f} ms, unembed "
          f"{sum(s.elapsed_time(e) for s, e in ev[(flag, 'unembed')]) / len(ev[(flag, 'unembed')]):.3f} ms")
names = sorted({k[1] for k in ev})
for n in names:
    a = sum(s.elapsed_time(e) for s, e in ev[(False, n)]) / len(ev[(False, n)]); b = sum(s.elapsed_time(e) for s, e in ev[(True, n)]) / len(ev[(True, n)])
    print(f"  {n:12s} {a:.3f} -> {b:.3f} ms")
