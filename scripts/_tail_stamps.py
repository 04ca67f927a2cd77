"""Timing experiment: stage timeline (s_memtime) of the fused output tail, one workgroup."""
import os, sys, ctypes, importlib
os.environ["TUP_TAIL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformerupscaler_amd import _lib
from transformerupscaler_amd.weights import deterministic_state_dict
m = importlib.import_module("models.FastTransformer.model").TransformerModel()
m.load_state_dict(deterministic_state_dict(0), strict=False)
m = m.cuda().eval()
x = torch.rand(8, 3, 720, 1280).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x, res_out=(1080, 1920))
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
lib.tup_debug_tail_stamps.restype = ctypes.c_int
print("rc", lib.tup_debug_tail_stamps(buf))
names = ["start", "A loads+write", "barrier", "zero t1+barrier", "stage B", "barrier", "stage C (+ui loads)", "barrier", "stage D"]
print("  ".join(f"{n}:{buf[i] - buf[i - 1] if i else 0}" for i, n in enumerate(names)), " total", buf[8] - buf[0])
