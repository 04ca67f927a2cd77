"""LayerNorm backward at the training bench's size (61,440 rows x 192): time per launch.  TUP_LN_BWD_BLOCKS caps the grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops
M = 61440
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(M, 192, device="cuda", generator=g)
gy = torch.randn(M, 192, device="cuda", generator=g).bfloat16()
gres = torch.randn(M, 192, device="cuda", generator=g)
gm = torch.ones(192, device="cuda")
_, mean, rstd = ops.layernorm(x, gm, torch.zeros(192, device="cuda"), save_stats=True)
ts = []
for r in range(14):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.layernorm_bwd(gy, x, mean, rstd, gm, gres=gres); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3)
ts = sorted(ts[2:])
print(f"blocks cap {os.environ.get('TUP_LN_BWD_BLOCKS', 'default')}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us (incl. two memsets + launch)")
