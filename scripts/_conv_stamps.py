"""Timing experiment: phase timeline (s_memtime) of the persistent ping-pong conv kernel, workgroup 100."""
import os, sys, ctypes
os.environ["TUP_CONV_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformerupscaler_amd import ops, packing, _lib
B = 8
x = (torch.randn(B, 720, 1280, 64, device="cuda") * 0.5).to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3) * 0.05
wp, bp = packing.pack_conv_c64(w, torch.zeros(64), 1)
wp, bp = wp.cuda(), bp.cuda()
for _ in range(3):
    y = ops.conv_c64(x, wp, bp, 1, relu=True)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 160)()
lib.tup_debug_conv_stamps.restype = ctypes.c_int
print("rc", lib.tup_debug_conv_stamps(buf))
t0 = min(buf[i] for i in range(160) if buf[i])
for grp in range(2):
    print("group", grp)
    for ph in range(2, 12):
        v = [buf[(grp * 16 + ph) * 5 + i] for i in range(5)]
        role = "K-loop" if (ph & 1) == grp else "store+DMA"
        if role == "K-loop":
            print(f"  ph {ph:2d} {role:10s} start {v[0]-t0:7d}  K loop {v[1]-v[0]:6d}  wait {v[3]-v[1]:5d}  barrier {v[4]-v[3]:6d}")
        else:
            print(f"  ph {ph:2d} {role:10s} start {v[0]-t0:7d}  DMA issue {v[1]-v[0]:6d}  stores {v[2]-v[1]:6d}  vmcnt wait {v[3]-v[2]:6d}  barrier {v[4]-v[3]:6d}")
