"""Per-phase cycle shares of the streamed whole-block kernel (diagnostic library: `make -C transformerupscaler_amd/csrc diag`,
TUP_LIB_PATH=transformerupscaler_amd/libtupscale_hip_diag.so TUP_BS_STAMPS=1 python scripts/bs_stamps.py)."""
import os, sys, ctypes
os.environ["TUP_BS_STAMPS"] = "1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TUP_LIB_PATH", os.path.join(root, "transformerupscaler_amd", "libtupscale_hip_diag.so"))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from transformerupscaler_amd import ops, _lib
import test_hip_kernels as T
nwin = 1920
raw, _ = T._block_operands("cuda", nwin)
x = raw["x"].cuda()
tab = ops.stream_table([T._stream_operands("cuda", raw)] * 6)
for _ in range(3):
    ops.blocks_stream(x.clone(), tab)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (2 * 8 * 20))()
lib.tup_debug_bs_stamps.restype = ctypes.c_int
print("rc", lib.tup_debug_bs_stamps(buf))
names = ["PRO", "bar_pro", "A0", "bar_a0", "SLOTS(rest)", "bar_slot", "PROJ_PRE", "bar_x", "PROJ", "LN2", "bar_y", "MLP(rest)", "bar_mlp", "TOTAL", "slot_dma", "slot_loop", "slot_pv", "mlp_dma"]
for rec in range(2):
    for w in range(8):
        v = [buf[(rec * 8 + w) * 20 + k] for k in range(18)]
        print(f"wg{rec} wave{w}: " + "  ".join(f"{n} {x / 6 / 1000:.1f}k" for n, x in zip(names, v)) + "   (cycles per block)")
