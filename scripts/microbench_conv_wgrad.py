"""conv3x3_wgrad_c64_kernel timing at the training shapes (4 x 720p, gr = 1); run on the MI355X box.  TUP_LIB_PATH selects a build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops
B, H, W = 4, 720, 1280
x = (torch.randn(B, H, W, 64, device="cuda") * 0.5).bfloat16()
g = (torch.randn(B, H, W, 64, device="cuda") * 0.1).bfloat16()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]
us = t(lambda: ops.conv_c64_wgrad(x, g, 1))
print("conv_c64_wgrad 4x720p: %.1f us (incl. two small zero fills)  %.0f TFLOP/s" % (us, 2 * B * H * W * 64 * 576 / us / 1e6))
d = (torch.randn(B, H, W, 64, device="cuda") * 0.5).bfloat16()
gp = torch.randn(B, 3, H, W, device="cuda") * 0.1
us = t(lambda: ops.conv_thin_wgrad(d, gp, True))
print("conv_thin_wgrad 4x720p: %.1f us  %.2f TB/s" % (us, (d.numel() * 2 + gp.numel() * 4) / us / 1e6))
