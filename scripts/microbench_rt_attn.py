"""Timing of the ResidualTransformer attention kernels alone (N = 3600, 8 heads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformerupscaler_amd import ops
for B in (2, 8):
    N = 3600
    qkv = (torch.randn(B * N, 384, device="cuda") * 0.5).to(torch.bfloat16)
    go = (torch.randn(B * N, 128, device="cuda") * 0.1).to(torch.bfloat16)
    o, lse = ops.rt_attention(qkv, B, N, save_lse=True)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print(f"B={B}: fwd {t(lambda: ops.rt_attention(qkv, B, N)):.1f} us, fwd+lse {t(lambda: ops.rt_attention(qkv, B, N, save_lse=True)):.1f} us, "
          f"bwd {t(lambda: ops.rt_attention_bwd(qkv, o, go, lse, B, N)):.1f} us", flush=True)
