import os, sys, ctypes
os.environ["TUP_MLP_ABLATE"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformerupscaler_amd import ops, _lib
M = 122880
x = torch.randn(M, 192, device="cuda")
gm = torch.ones(192, device="cuda"); bt = torch.zeros(192, device="cuda")
w1 = (torch.randn(768, 192, device="cuda") * 0.05).to(torch.bfloat16); b1 = torch.zeros(768, device="cuda")
w2 = (torch.randn(192, 768, device="cuda") * 0.05).to(torch.float16); b2 = torch.zeros(192, device="cuda")
for _ in range(3):
    ops.fused_mlp(x, gm, bt, w1, b1, w2, b2)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 256)()
lib.tup_debug_mlp_stamps.restype = ctypes.c_int
print("rc", lib.tup_debug_mlp_stamps(buf))
names = ["start", "prologue"]
for j in range(3):
    names += [f"c{j} top", f"c{j} vmcnt", f"c{j} barrier", f"c{j} FC1a", f"c{j} W2bar", f"c{j} GELUa", f"c{j} FC1b(+FC2a)", f"c{j} GELUb"]
names += ["loop done", "stores retired"]
for wg in range(4):
    t = [buf[wg * 64 + i] for i in range(len(names))]
    print("WG slot", wg, "total", t[-1] - t[0])
    print("  " + "  ".join(f"{n}:{t[i] - t[i - 1] if i else 0}" for i, n in enumerate(names)))
