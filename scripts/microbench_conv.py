"""Micro-benchmark of the 64->64 3x3 conv kernel variants (run on the MI355X box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops, packing

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

B, H, W = 8, 720, 1280
x = (torch.rand((B, H, W, 64), device="cuda") - 0.5).to(torch.bfloat16)
w = torch.rand((64, 64, 3, 3)) - 0.5
b = torch.rand((64,))
wp, bp = packing.pack_conv_c64(w, b, 1)
wp, bp = wp.cuda(), bp.cuda()
gf = 2.0 * B * H * W * 64 * 576 / 1e9
for name, fn in [("bias+relu", lambda: ops.conv_c64(x, wp, bp, 1, relu=True)),
                 ("no bias", lambda: ops.conv_c64(x, wp, None, 1, relu=False))]:
    ms = timeit(fn)
    print(f"{name:12s} {ms:.3f} ms  {gf / ms:.0f} TFLOP/s", flush=True)
