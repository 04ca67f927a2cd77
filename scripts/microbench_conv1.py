"""conv1 (3 -> 64, 720p, 8 images) kernel time; run on the MI355X box.  TUP_CONV1_ONE_TILE=1 selects the one-tile kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops, packing
x = torch.rand(8, 3, 720, 1280, device="cuda")
w = packing.pack_conv1(torch.randn(64, 3, 3, 3) * 0.1).cuda()
b = torch.zeros(64, device="cuda")
for _ in range(5): y = ops.conv1(x, w, b, relu=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): y = ops.conv1(x, w, b, relu=True)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 50
print("conv1 %.1f us  (%.2f TB/s written)" % (ms * 1e3, y.numel() * 2 / ms / 1e9))
