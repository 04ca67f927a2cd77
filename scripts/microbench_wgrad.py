"""gemm_wgrad timing per Linear-layer shape of a FastTransformer training step (4 images); run on the MI355X box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops
M = 4 * 15360
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for NI, NJ in ((576, 192), (192, 192), (768, 192), (192, 768)):
    p = torch.randn(M, NI, device="cuda").to(torch.bfloat16)
    q = torch.randn(M, NJ, device="cuda").to(torch.bfloat16)
    out = torch.zeros(NI, NJ, device="cuda")
    us = t(lambda: ops.gemm_wgrad(p, q, out=out))
    print("NI %4d NJ %4d: %.1f us  %.0f TFLOP/s  %.2f TB/s (operands once)" % (NI, NJ, us, 2 * M * NI * NJ / us / 1e6, (p.numel() + q.numel()) * 2 / us / 1e6))

# the two patch weights (192 x 4096, M = every token of 4 x 720p): the wide-tile kernel against the 64 x 64-tile one
B, H, W = 4, 720, 1280
fmap = (torch.randn(B, H, W, 64, device="cuda") * 0.5).to(torch.bfloat16)
p = torch.randn(B * 12 * 20 * 64, 192, device="cuda")
for wide in (True, False, True, False):
    ops.PATCH_WGRAD_WIDE = wide
    us = t(lambda: ops.patch_wgrad(p, fmap, True), n=10)
    print("patch_wgrad (incl. the zero fill%s): %s %.1f us" % (" and the bf16 cast" if wide else "", "wide" if wide else "64x64", us))
ops.PATCH_WGRAD_WIDE = True
