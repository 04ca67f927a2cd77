"""A/B of builds of the whole-block kernel in ONE process on the MI355X box (interleaved rounds, same GPU, same data):
    python scripts/ab_block.py name=path/to/lib.so [name=...]      (scripts/_ab_build.sh makes the libraries)
Per library: 6 launches of tup_fused_block_fwd, and tup_fused_blocks32_fwd (6 blocks, one launch) where exported."""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from transformerupscaler_amd import ops
import test_hip_kernels as T

nwin, rounds = 1920, 15
libs = dict(a.split("=") for a in sys.argv[1:])
raw, args = T._block_operands("cuda", nwin)
x0 = raw["x"].to("cuda")
# libraries built before ABI 9 (flag "@abi8") take gamma / beta explicitly: 13 pointers per block instead of 9
from transformerupscaler_amd import packing
_wh, _bh = packing.pack_qkv_heads(raw["w"], raw["b"])
_w1, _b1 = packing.pack_fc1_fused_q(raw["w1"], raw["b1"])
args8 = [raw["gm1"].cuda(), raw["bt1"].cuda(), _wh.cuda(), _bh.cuda(), args[2], args[3], args[4], raw["gm2"].cuda(), raw["bt2"].cuda(),
         _w1.cuda(), _b1.cuda(), args[7], args[8]]
P, I = ctypes.c_void_p, ctypes.c_int
runs = {}
keep = []
for name, path in libs.items():
    path, _, flag = path.partition("@")
    L = ctypes.CDLL(os.path.join(root, path))
    a = args8 if flag == "abi8" else args
    L.tup_fused_block_fwd.argtypes = [P] * (1 + len(a)) + [I, P]
    ptrs = [t.data_ptr() for t in a]
    runs[name + "/6 launches"] = (lambda x, L=L, ptrs=ptrs: [L.tup_fused_block_fwd(x.data_ptr(), *ptrs, nwin, None) for _ in range(6)])
    L.tup_fused_blocks32_fwd.argtypes = [P, P, I, I, P]
    tab = (ctypes.c_void_p * (6 * len(a)))(*(ptrs * 6))
    keep.append(tab)
    runs[name + "/1 launch"] = (lambda x, L=L, tab=tab: L.tup_fused_blocks32_fwd(x.data_ptr(), tab, 6, nwin, None))
x = x0.clone()
ref = None
for k, f in runs.items():
    for _ in range(2):
        x.copy_(x0); f(x)
    torch.cuda.synchronize()
    ref = x.clone() if ref is None else ref
    print(k, "max |diff| vs first variant", (x - ref).abs().max().item())
times = {k: [] for k in runs}
st = torch.cuda.current_stream()
assert st.cuda_stream == 0, "the libraries are launched on the null stream"
for r in range(rounds):
    for k, f in runs.items():
        x.copy_(x0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(x); e.record(); torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) / 6 * 1e3)
for k, t in times.items():
    t = sorted(t)
    print(f"{k}: median {t[len(t) // 2]:.1f} us  min {t[0]:.1f} us per block")
