"""Timing of the streamed 32x32x16 whole-block kernel (tup_blocks_stream_fwd) against the 16x16x32 one (tup_fused_blocks32_fwd) in ONE
process on the MI355X box: six blocks in one launch at 1,920 windows (BASELINE configs[1]), interleaved rounds, same data.
    python scripts/ab_stream.py [nwin] [rounds]"""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from transformerupscaler_amd import ops
import test_hip_kernels as T

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 15
sets = [T._block_operands("cuda", nwin, seed=40 + i) for i in range(3)]
x0 = sets[0][0]["x"].to("cuda")
t32 = ops.block_table([tuple(sets[i % 3][1]) for i in range(6)])
st = [T._stream_operands("cuda", s[0]) for s in sets]
tst = ops.stream_table([st[i % 3] for i in range(6)])
runs = {"blocks32 (16x16x32)": lambda x: ops.fused_blocks32(x, t32), "stream (32x32x16)": lambda x: ops.blocks_stream(x, tst)}
x = x0.clone()
outs = {}
for k, f in runs.items():
    for _ in range(2):
        x.copy_(x0); f(x)
    torch.cuda.synchronize()
    outs[k] = x.clone()
a, b = list(outs.values())
print("finite:", bool(torch.isfinite(b).all()), " stream vs blocks32: max |diff|", (a - b).abs().max().item(), "mean", (a - b).abs().mean().item())
times = {k: [] for k in runs}
for r in range(rounds):
    for k, f in runs.items():
        x.copy_(x0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(x); e.record(); torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) * 1e3)
gf = 86.1e9 * nwin / 240
for k, t in times.items():
    t = sorted(t)
    med = t[len(t) // 2]
    print(f"{k}: median {med:.1f} us  min {t[0]:.1f} us per six-block launch = {gf / med / 1e6 / 2500:.3f} of the 2.5 PFLOP/s MFMA peak (median)")
