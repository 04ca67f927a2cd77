"""Per-phase cycle shares of the whole-block kernel (diagnostic build, TUP_B32_STAMPS=1)."""
import os, sys, ctypes
os.environ["TUP_B32_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from transformerupscaler_amd import ops, _lib
import test_hip_kernels as T
nwin = 1920
raw, args = T._block_operands("cuda", nwin)
x = raw["x"].cuda()
six = len(sys.argv) > 1 and sys.argv[1] == "6"            # the last of six blocks in one launch instead of a single block
table6 = ops.block_table([tuple(args)] * 6)
for _ in range(3):
    if six:
        ops.fused_blocks32(x.clone(), table6)
    else:
        ops.fused_block(x.clone(), *args)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4 * 16))()
lib.tup_debug_block32_stamps.restype = ctypes.c_int
print("rc", lib.tup_debug_block32_stamps(buf))
names = ["LN1", "SYNC0", "QKV", "HBAR", "ATT", "PROJ", "LN2", "MTOP", "FC1", "W2BAR", "GELU", "FC2", "STORE", "TOTAL"]
for rec in range(8):
    for wave in (0, 3):
        v = [buf[(rec * 4 + wave) * 16 + k] for k in range(len(names))]
        tot = v[-1]
        print(f"wg {rec} wave {wave} total {tot}: " + "  ".join(f"{n}:{v[i]} ({100.0 * v[i] / max(tot, 1):.0f}%)" for i, n in enumerate(names[:-1])))
