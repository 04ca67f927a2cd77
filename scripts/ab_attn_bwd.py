"""Time of the window-attention backward (tup_window_attn_bwd + its two small reduce kernels) at the training bench's size:
960 windows (B = 4 at 720p), 12 heads, dropout 0.1 and 0.    python scripts/ab_attn_bwd.py"""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from transformerupscaler_amd import ops
nwin = 960
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(nwin * 64, 576, device="cuda", generator=g) * 0.5).bfloat16()
gout = (torch.randn(nwin * 64, 192, device="cuda", generator=g) * 0.1).bfloat16()
table = torch.randn(225, 12, device="cuda", generator=g) * 0.2
ft, fn = ops.relpos_bias_expand(table), ops.relpos_bias_expand_n(table)
for p in (0.1, 0.0):
    att, lse = ops.window_attn(qkv, ft, p, 1234, save_lse=True)
    tf, tb = [], []
    for r in range(12):
        s, m, e = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s.record(); ops.window_attn(qkv, ft, p, 1234, save_lse=True); m.record(); ops.window_attn_bwd(qkv, gout, att, lse, fn, p, 1234); e.record()
        torch.cuda.synchronize()
        tf.append(s.elapsed_time(m) * 1e3); tb.append(m.elapsed_time(e) * 1e3)
    tf, tb = sorted(tf[2:]), sorted(tb[2:])
    print(f"dropout {p}: forward median {tf[len(tf) // 2]:.1f} us, backward (3 launches) median {tb[len(tb) // 2]:.1f} us  min {tb[0]:.1f} us")
