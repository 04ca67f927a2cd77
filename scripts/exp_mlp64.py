"""EXPERIMENT: the MLP half of the streamed block kernel with one wave per SIMD and 64 tokens per wave (csrc/experiments/mlp64_exp.hip,
`make -C transformerupscaler_amd/csrc exp`), checked against torch and timed on the MI355X box.
    python scripts/exp_mlp64.py [nwin] [reps] [rounds]"""
import ctypes, os, sys
import torch
import torch.nn.functional as F
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from transformerupscaler_amd import ops  # noqa: F401  (loads the runtime the way the package does)
import test_hip_kernels as T

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 15
lib = ctypes.CDLL(os.path.join(root, "transformerupscaler_amd", "libtupscale_mlp64_exp%s.so" % os.environ.get("M64_VARIANT", "")))
P = ctypes.c_void_p
lib.tup_exp_mlp64.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_int, P]
raw, _ = T._block_operands("cuda", nwin)
wqk, wv, wproj, w1, w2, tab, sbias = T._stream_operands("cuda", raw)
x0 = raw["x"].to("cuda")


def run(x, n):
    err = lib.tup_exp_mlp64(x.data_ptr(), w1.data_ptr(), w2.data_ptr(), tab.data_ptr(), nwin, n, torch.cuda.current_stream().cuda_stream)
    assert err == 0, err


def ref(x):
    bf = lambda t: t.to(torch.bfloat16).float()
    y = bf(F.layer_norm(x, (192,), raw["gm2"].cuda(), raw["bt2"].cuda(), 1e-5))
    hid = F.gelu(y @ bf(raw["w1"].cuda()).t() + raw["b1"].cuda())
    return x + hid.half().float() @ raw["w2"].cuda().half().float().t() + raw["b2"].cuda()


for n in (1, 2):
    x = x0.clone(); run(x, n); torch.cuda.synchronize()
    want = x0
    for _ in range(n): want = ref(want)
    d = (x - want).abs()
    print(f"reps {n}: finite {bool(torch.isfinite(x).all())}  vs torch max {d.max().item():.3e} mean {d.mean().item():.3e}  (|ref| max {want.abs().max().item():.2f})")
ts = []
x = x0.clone()
for r in range(rounds):
    x.copy_(x0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(x, reps); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3)
ts.sort()
med = ts[len(ts) // 2]
fl = 2 * 2 * 192 * 768 * 64 * nwin * reps
print(f"mlp64 x{reps}: median {med:.1f} us min {ts[0]:.1f} us = {med / reps:.1f} us per MLP half; {fl / med / 1e6 / 2500:.3f} of the 2.5 PFLOP/s peak")
