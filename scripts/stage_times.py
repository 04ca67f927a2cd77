"""Untraced per-stage event timing of the inference forward (run on the MI355X box): rocprofv3 inflates some kernels."""
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
on = [False]

@contextlib.contextmanager
def timer(name):
    if not on[0]:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); yield; e.record()
    ev[name].append((s, e))

engine.stage_timer = timer
with torch.no_grad():
    for _ in range(3):
        m(x, res_out=(1080, 1920))
    on[0] = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        m(x, res_out=(1080, 1920))
    t1.record()
torch.cuda.synchronize()
print("forward %.3f ms" % (t0.elapsed_time(t1) / 10))
for k, v in ev.items():
    print("  %-10s %.3f ms" % (k, sum(s.elapsed_time(e) for s, e in v) / 10))
