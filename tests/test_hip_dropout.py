"""Dropout on the HIP training path (reference model.py:80-82,127,132,150: attn_drop, proj_drop, MLP Dropout,
p = 0.1 in .train()).  The masks come from a stateless hash (csrc/common.h drop_hash) restated here in numpy,
so forward AND backward can be checked exactly against torch math using the same mask."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fast_transformer_oracle as O

pytestmark = pytest.mark.gpu


def drop_hash(seed, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    h = (idx * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def keep_mask(seed, n, p):
    thresh = np.uint64(int(p * 4294967296.0))
    return torch.from_numpy((drop_hash(seed, np.arange(n)) >= thresh).astype(np.float32))


def keep_mask_pairs(seed, n, p):
    """Attention-probability masks (csrc/common.h drop_pair*): one hash per PAIR of neighbouring keys, the even key takes the low 16
    bits, the odd key the high 16 bits; keep iff field >= round(p * 65536).  Returns (mask, 1 / keep probability of the mask)."""
    t = min(max(int(p * 65536.0 + 0.5), 1), 65535)
    h = drop_hash(seed, np.arange(n) >> 1)
    field = np.where((np.arange(n) & 1) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return torch.from_numpy((field >= np.uint64(t)).astype(np.float32)), 65536.0 / (65536.0 - t)


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(shape, seed, scale=1.0):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * scale


def test_gemm_residual_dropout_and_backward_mask():
    from transformerupscaler_amd import ops, packing
    M, N, K, p, seed = 320, 192, 768, 0.1, 12345
    a, w, b, res = rnd((M, K), 1), rnd((N, K), 2, 0.08), rnd((N,), 3, 0.2), rnd((M, N), 4)
    mask = keep_mask(seed, M * N, p).view(M, N)
    assert abs(mask.mean().item() - 0.9) < 0.01
    ref = (F.linear(bf(a), bf(w), b) * mask / (1 - p)) + res
    got = ops.gemm_tokens(a.to(torch.bfloat16).cuda(), packing.pack_linear(w).cuda(), b.cuda(), "res", res=res.cuda(),
                          drop_p=p, drop_seed=seed).cpu()
    assert (got - ref).abs().max() <= 3e-4 + 1e-4 * ref.abs().max()
    g = rnd((M, N), 5)
    gd = ops.dropout_bwd(g.cuda(), p, seed).float().cpu()
    assert (gd - bf(g * mask / (1 - p))).abs().max() == 0


def test_attention_dropout_forward_backward(det_sd):
    from transformerupscaler_amd import ops
    nwin, p, seed = 9, 0.1, 777
    qkv = bf(rnd((nwin, 64, 576), 6, 1.5)).requires_grad_(True)
    table = det_sd["window_blocks.0.attn.relative_position_bias_table"].clone().requires_grad_(True)
    idx = O.relative_position_index(8)
    q, k, v = qkv.view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    attn = ((q * 0.25) @ k.transpose(-2, -1) + table[idx.view(-1)].view(64, 64, 12).permute(2, 0, 1).unsqueeze(0)).softmax(-1)
    mask, inv_keep = keep_mask_pairs(seed, nwin * 12 * 64 * 64, p)              # index ((win*12+h)*64+q)*64+k
    mask = mask.view(nwin, 12, 64, 64)
    assert abs(mask.mean().item() - 0.9) < 0.005 and abs(inv_keep - 1 / 0.9) < 1e-4
    out = ((attn * mask * inv_keep) @ v).transpose(1, 2).reshape(nwin * 64, 192)
    gout = bf(rnd((nwin * 64, 192), 7))
    out.backward(gout)
    tb = table.detach().cuda()
    ft, fn = ops.relpos_bias_expand(tb), ops.relpos_bias_expand_n(tb)
    qd = qkv.detach().view(-1, 576).to(torch.bfloat16).cuda()
    att, lse = ops.window_attn(qd, ft, p, seed, save_lse=True)
    got = att.float().cpu()
    assert (got - out.detach()).abs().max() <= 2e-2
    gq, dt = ops.window_attn_bwd(qd, gout.to(torch.bfloat16).cuda(), att, lse, fn, p, seed)
    assert (gq.float().cpu() - qkv.grad.view(-1, 576)).abs().max() <= 3e-2 + 2e-2 * qkv.grad.abs().max()
    assert (dt.cpu() - table.grad).abs().max() <= 5e-2 + 2e-2 * table.grad.abs().max()


def test_module_train_mode_uses_dropout(det_sd):
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda()
    x = torch.rand((1, 3, 40, 56), generator=torch.Generator().manual_seed(1)).cuda()
    m.eval()
    with torch.no_grad():
        y_eval = m(x, upscale_factor=2)
    y_eval_grad = m(x, upscale_factor=2)                 # eval graph with grads: no dropout, same numbers up to bf16 paths
    assert (y_eval_grad.detach() - y_eval).abs().max() <= 2.5e-2
    m.train()
    y1 = m(x, upscale_factor=2)
    y2 = m(x, upscale_factor=2)
    assert not torch.equal(y1, y2), "train mode must draw a fresh dropout mask per call"
    assert (y1.detach() - y_eval).abs().mean() < 0.05          # same function in expectation
    y1.sum().backward()
    grads = [p.grad for p in m.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)
    torch.manual_seed(5); m._dropout_calls = 0
    ya = m(x, upscale_factor=2).detach().clone()
    torch.manual_seed(5); m._dropout_calls = 0
    yb = m(x, upscale_factor=2).detach()
    assert torch.equal(ya, yb), "same seed -> same masks"
