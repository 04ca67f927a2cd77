"""Micro-benchmark of the token-path kernels (run on the MI355X box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops, packing

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

M = 8 * 240 * 64
dev = "cuda"
x = torch.randn((M, 192), device=dev)
gm, bt = torch.ones(192, device=dev), torch.zeros(192, device=dev)
w1raw = torch.randn(768, 192) * 0.05
w1, b1 = packing.pack_linear(w1raw).to(dev), torch.zeros(768, device=dev)
w1f, b1q = [t.to(dev) for t in packing.pack_fc1_fused_q(w1raw, torch.zeros(768))]
w2raw = torch.randn(192, 768) * 0.05
w2, b2 = packing.pack_linear(w2raw).to(dev), torch.zeros(192, device=dev)
w2h = packing.pack_fc2_h4(w2raw).to(dev)
wq, bq = packing.pack_linear(torch.randn(576, 192) * 0.05).to(dev), torch.zeros(576, device=dev)
wp, bp = packing.pack_linear(torch.randn(192, 192) * 0.05).to(dev), torch.zeros(192, device=dev)
frag = ops.relpos_bias_expand(torch.zeros(225, 12, device=dev))
print("fused_mlp   %.1f us" % timeit(lambda: ops.fused_mlp(x, gm, bt, w1f, b1q, w2h, b2)))
def unfused_mlp():
    y = ops.layernorm(x, gm, bt); h = ops.gemm_tokens(y, w1, b1, "gelu"); ops.gemm_tokens(h, w2, b2, "res", res=x, out=x)
print("unfused_mlp %.1f us" % timeit(unfused_mlp))
def unfused_attn():
    y = ops.layernorm(x, gm, bt); q = ops.gemm_tokens(y, wq, bq, "bf16"); a = ops.window_attn(q, frag); ops.gemm_tokens(a, wp, bp, "res", res=x, out=x)
print("unfused_attn %.1f us" % timeit(unfused_attn))
if hasattr(ops, "fused_attn"):
    print("fused_attn  %.1f us" % timeit(lambda: ops.fused_attn(x, gm, bt, wq, bq, frag, wp, bp)))
print("ln+qkv panel %.1f us" % timeit(lambda: ops.ln_gemm(x, gm, bt, wq, bq)))
y_ = ops.layernorm(x, gm, bt)
print("qkv gemm    %.1f us" % timeit(lambda: ops.gemm_tokens(y_, wq, bq, "bf16")))
a_ = ops.window_attn(ops.gemm_tokens(y_, wq, bq, "bf16"), frag)
print("attn core   %.1f us" % timeit(lambda: ops.window_attn(ops.gemm_tokens(y_, wq, bq, "bf16"), frag)))
print("proj+res    %.1f us" % timeit(lambda: ops.gemm_tokens(a_, wp, bp, "res", res=x, out=x)))
print("fc1 gelu    %.1f us" % timeit(lambda: ops.gemm_tokens(y_, w1, b1, "gelu")))
from transformerupscaler_amd import packing as _pk
_wh, _bh = _pk.pack_qkv_heads(torch.randn(576, 192) * 0.07, torch.randn(576) * 0.1)
_wh, _bh = _wh.cuda(), _bh.cuda()
print("fused ln+qkv+attn %.1f us" % timeit(lambda: ops.fused_qkv_attn(x, gm, bt, _wh, _bh, frag)))
