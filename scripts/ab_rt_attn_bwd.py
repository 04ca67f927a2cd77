"""ResidualTransformer attention backward (prep + dQ + dK/dV launches) with dropout 0.1 at the bench size (B = 2, N = 3600), one
library per process:    TUP_LIB_PATH=build/ab_x.so python scripts/ab_rt_attn_bwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops
B, N = 2, 3600
qkv = (torch.randn(B * N, 384, device="cuda") * 0.5).to(torch.bfloat16)
go = (torch.randn(B * N, 128, device="cuda") * 0.1).to(torch.bfloat16)
o, lse = ops.rt_attention(qkv, B, N, save_lse=True, drop_p=0.1, drop_seed=5)
def t(fn, n=14):
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts[3:])[len(ts[3:]) // 2]
print(f"{os.environ.get('TUP_LIB_PATH', 'default').split('/')[-1]}: fwd {t(lambda: ops.rt_attention(qkv, B, N, save_lse=True, drop_p=0.1, drop_seed=5)):.1f} us, "
      f"bwd {t(lambda: ops.rt_attention_bwd(qkv, o, go, lse, B, N, drop_p=0.1, drop_seed=5)):.1f} us")
