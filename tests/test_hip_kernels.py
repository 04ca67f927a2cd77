"""Per-kernel parity on the MI355X: every C-ABI entry point against the torch-CPU restatement of the
aten op it replaces (oracle side), on bf16-rounded operands.  Integer/index work (window layout,
pixel shuffle, padding) is checked exactly through values; floating point within bf16 tolerances
stated per test."""
import numpy as np
import pytest
import os
import torch
import torch.nn.functional as F

from oracle import fast_transformer_oracle as O

pytestmark = pytest.mark.gpu


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale + shift


def close(a, b, atol, rtol, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    assert bool((err <= bound).all()), f"{what}: max err {err.max().item():.3e}, worst ratio {(err / bound).max().item():.2f}"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from transformerupscaler_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 3, 37, 70), (1, 3, 8, 32), (1, 3, 64, 96)])
def test_conv1(dev, shape):
    from transformerupscaler_amd import ops, packing
    x = rnd(shape, 1, 0.5, 0.5)
    w, b = rnd((64, 3, 3, 3), 2, 0.3), rnd((64,), 3, 0.2)
    ref = F.relu(F.conv2d(bf(x), bf(w), b, padding=1)).permute(0, 2, 3, 1)
    got = ops.conv1(x.to(dev), packing.pack_conv1(w).to(dev), b.to(dev), relu=True)
    close(got, ref, 1e-2, 1e-2, "conv1")


@pytest.mark.parametrize("r,hw", [(1, (19, 45)), (2, (16, 64)), (3, (9, 33)), (6, (8, 20)), (1, (8, 32))])
def test_conv_c64_pixelshuffle(dev, r, hw):
    from transformerupscaler_amd import ops, packing
    H, W = hw
    x = bf(rnd((2, 64, H, W), 4))
    w, b = rnd((64 * r * r, 64, 3, 3), 5, 0.06), rnd((64 * r * r,), 6, 0.2)
    ref = F.pixel_shuffle(F.conv2d(x, bf(w), b, padding=1), r).permute(0, 2, 3, 1)
    wp, bp = packing.pack_conv_c64(w, b, r)
    got = ops.conv_c64(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev), wp.to(dev), bp.to(dev), r, relu=False)
    assert tuple(got.shape) == (2, H * r, W * r, 64)
    close(got, ref, 1.5e-2, 1e-2, f"conv_c64 r={r}")
    got = ops.conv_c64(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev), wp.to(dev), bp.to(dev), r, relu=True)
    close(got, F.relu(ref), 1.5e-2, 1e-2, f"conv_c64 relu r={r}")


@pytest.mark.parametrize("co,bias,relu", [(3, False, True), (3, True, False)])
def test_conv_c64_thin(dev, co, bias, relu):
    from transformerupscaler_amd import ops, packing
    x = bf(rnd((2, 64, 21, 50), 7))
    w = rnd((co, 64, 3, 3), 8, 0.06)
    b = rnd((co,), 9, 0.2) if bias else None
    ref = F.conv2d(x, bf(w), b, padding=1)
    if relu:
        ref = F.relu(ref)
    got = ops.conv_c64_thin(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev), packing.pack_conv_c64_thin(w).to(dev),
                            None if b is None else b.to(dev), co, relu=relu)
    close(got, ref, 2e-4, 1e-4, "thin conv (fp32 out)")


@pytest.mark.parametrize("r", [1, 2, 3, 6])
def test_conv_planar(dev, r):
    from transformerupscaler_amd import ops, packing
    x = rnd((2, 3, 23, 70), 10)
    w, b = rnd((3 * r * r, 3, 3, 3), 11, 0.3), rnd((3 * r * r,), 12, 0.2)
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), r)
    got = ops.conv_planar(x.to(dev), packing.pack_planar(w).to(dev), b.to(dev), r)
    close(got, ref, 2e-6, 1e-5, "planar conv")
    add = rnd(tuple(ref.shape), 13)
    got = ops.conv_planar(x.to(dev), packing.pack_planar(w).to(dev), b.to(dev), r, add=add.to(dev), clamp=True)
    close(got, (ref + add).clamp(0, 1), 2e-6, 1e-5, "planar conv + add + clamp")


@pytest.mark.parametrize("hw", [(23, 72), (5, 4), (9, 260), (33, 1024)])
def test_conv_planar_r1_four_pixels_per_thread(dev, hw):
    """The r = 1, W % 4 == 0 kernel (four pixels per thread, final_upscale_conv at HR and its input gradient) against torch: image
    borders, widths that are not a multiple of the 256-pixel block, a single 4-pixel column; with and without add + clamp."""
    from transformerupscaler_amd import ops, packing
    x = rnd((2, 3) + hw, 20)
    w, b = rnd((3, 3, 3, 3), 21, 0.3), rnd((3,), 22, 0.2)
    ref = F.conv2d(x, w, b, padding=1)
    close(ops.conv_planar(x.to(dev), packing.pack_planar(w).to(dev), b.to(dev), 1), ref, 2e-6, 1e-5, "planar conv r1x4")
    close(ops.conv_planar(x.to(dev), packing.pack_planar(w).to(dev), None, 1), F.conv2d(x, w, None, padding=1), 2e-6, 1e-5, "no bias")
    add = rnd(tuple(ref.shape), 23)
    got = ops.conv_planar(x.to(dev), packing.pack_planar(w).to(dev), b.to(dev), 1, add=add.to(dev), clamp=True)
    close(got, (ref + add).clamp(0, 1), 2e-6, 1e-5, "planar conv r1x4 + add + clamp")


@pytest.mark.parametrize("sizes", [((72, 96), (54, 72)), ((64, 64), (48, 48)), ((30, 40), (45, 47)), ((144, 256), (108, 192))])
def test_resize_aa(dev, sizes):
    from transformerupscaler_amd import ops
    (h, w), (oh, ow) = sizes
    x = rnd((2, 3, h, w), 14, 0.8, 0.5)
    got = ops.resize_aa(x.to(dev), (oh, ow), clamp=False)
    close(got, O.aa_resize(x, (oh, ow)), 3e-6, 0, "resize")
    got = ops.resize_aa(x.to(dev), (oh, ow), clamp=True)
    close(got, O.aa_resize(x, (oh, ow)).clamp(0, 1), 3e-6, 0, "resize+clamp")
    close(ops.clamp01(x.to(dev)), x.clamp(0, 1), 0, 0, "clamp")


def test_layernorm(dev):
    from transformerupscaler_amd import ops
    x = rnd((200, 192), 15, 2.0, 0.3)
    gm, bt = rnd((192,), 16, 0.1, 1.0), rnd((192,), 17, 0.1)
    y, mean, rstd = ops.layernorm(x.to(dev), gm.to(dev), bt.to(dev), save_stats=True)
    close(y, F.layer_norm(x, (192,), gm, bt, 1e-5), 1e-2, 8e-3, "layernorm")
    close(mean, x.mean(1), 1e-5, 1e-5, "mean")
    close(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5), 1e-5, 1e-5, "rstd")


@pytest.mark.parametrize("M,N,K", [(192, 576, 192), (320, 192, 768), (128, 768, 192)])
def test_gemm_tokens(dev, M, N, K):
    from transformerupscaler_amd import ops, packing
    a, w, b = rnd((M, K), 18), rnd((N, K), 19, 0.08), rnd((N,), 20, 0.2)
    wp = packing.pack_linear(w).to(dev)
    ref = F.linear(bf(a), bf(w), b)
    close(ops.gemm_tokens(a.to(torch.bfloat16).to(dev), wp, b.to(dev), "bf16"), ref, 1e-2, 1e-2, "gemm bf16")
    close(ops.gemm_tokens(a.to(torch.bfloat16).to(dev), wp, b.to(dev), "gelu"), F.gelu(ref), 1e-2, 1e-2, "gemm gelu")
    res = rnd((M, N), 21)
    got = ops.gemm_tokens(a.to(torch.bfloat16).to(dev), wp, b.to(dev), "res", res=res.to(dev))
    close(got, ref + res, 2e-4, 1e-4, "gemm res fp32")
    xs = res.to(dev).clone()        # in-place residual update, as the engine uses it
    ops.gemm_tokens(a.to(torch.bfloat16).to(dev), wp, b.to(dev), "res", res=xs, out=xs)
    close(xs, ref + res, 2e-4, 1e-4, "gemm res in place")


def test_window_attention(dev, det_sd):
    from transformerupscaler_amd import ops
    nwin = 5
    qkv = bf(rnd((nwin, 64, 576), 22, 1.5))
    table = det_sd["window_blocks.2.attn.relative_position_bias_table"]
    idx = O.relative_position_index(8)
    q, k, v = qkv.view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    attn = (q * 0.25) @ k.transpose(-2, -1) + table[idx.view(-1)].view(64, 64, 12).permute(2, 0, 1).unsqueeze(0)
    ref = (attn.softmax(-1) @ v).transpose(1, 2).reshape(nwin * 64, 192)
    frag = ops.relpos_bias_expand(table.to(dev))
    got = ops.window_attn(qkv.view(nwin * 64, 576).to(torch.bfloat16).to(dev), frag)
    close(got, ref, 1.5e-2, 1e-2, "window attention")


@pytest.mark.parametrize("hw", [(20, 28), (68, 84), (64, 64)])
def test_patch_embed_unembed(dev, hw):
    from transformerupscaler_amd import ops, packing
    H, W = hw
    B = 2
    feat = bf(rnd((B, 64, H, W), 23))
    w, b = rnd((192, 64, 8, 8), 24, 0.02), rnd((192,), 25, 0.2)
    ph, pw = (8 - H % 8) % 8, (8 - W % 8) % 8
    fp = F.pad(feat, (0, pw, 0, ph), mode="reflect") if (ph or pw) else feat
    tok = F.conv2d(fp, bf(w), b, stride=8).permute(0, 2, 3, 1)
    ht, wt = tok.shape[1:3]
    pb, pr = (8 - ht % 8) % 8, (8 - wt % 8) % 8
    tokp = F.pad(tok.permute(0, 3, 1, 2), (0, pr, 0, pb)).permute(0, 2, 3, 1).contiguous()
    ref = O.window_partition(tokp, 8).reshape(-1, 192)
    nhwc = feat.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    got = ops.patch_embed(nhwc, packing.pack_patch_embed(w).to(dev), b.to(dev))
    close(got, ref, 2e-3, 1e-3, "patch embed")
    pad_rows = (ref.abs().sum(1) == 0)
    assert bool((got.cpu()[pad_rows] == 0).all()), "zero-padded tokens must be exact zeros"

    # unembed: x (window layout, fp32) -> NHWC map + skip
    xw = rnd(tuple(ref.shape), 26)
    wu, bu = rnd((192, 64, 8, 8), 27, 0.05), rnd((64,), 28, 0.2)
    t = O.window_reverse(bf(xw).view(B, -1, 64, 192), 8, ht + pb, wt + pr)[:, :ht, :wt, :].permute(0, 3, 1, 2)
    refu = (F.conv_transpose2d(t, bf(wu), bu, stride=8)[:, :, :H, :W] + feat).permute(0, 2, 3, 1)
    gotu = ops.patch_unembed(xw.to(dev), packing.pack_patch_unembed(wu).to(dev), bu.to(dev), nhwc)
    close(gotu, refu, 2e-2, 1e-2, "patch unembed + skip")


# ------------------------------------------------------------------------------------------------
# backward kernels vs torch autograd on the same (bf16-rounded) operands
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,NI,NJ,pd,qd", [(200, 192, 64, "bf16", "bf16"), (1000, 576, 192, "bf16", "bf16"),
                                           (333, 192, 768, "f32", "bf16"), (64, 64, 64, "f32", "f32")])
def test_gemm_wgrad_and_colsum(dev, M, NI, NJ, pd, qd):
    from transformerupscaler_amd import ops
    P, Q = rnd((M, NI), 30), rnd((M, NJ), 31)
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    ref = bf(P).t() @ bf(Q)
    got = ops.gemm_wgrad(P.to(dt[pd]).to(dev), Q.to(dt[qd]).to(dev))
    close(got, ref, 2e-3 * (M ** 0.5), 1e-3, "wgrad")
    src = P.to(dt[pd])
    close(ops.colsum(src.to(dev)), src.float().sum(0), 1e-3, 1e-4, "colsum")
    # weight and bias gradient in one pass: the column sums ride on an MFMA against a ones operand (bf16 rounding of an
    # fp32 P happens before the sum, as for the weight gradient)
    gw, gb = ops.gemm_wgrad_bias(P.to(dt[pd]).to(dev), Q.to(dt[qd]).to(dev))
    close(gw, ref, 2e-3 * (M ** 0.5), 1e-3, "wgrad (fused)")
    close(gb, bf(P).sum(0), 2e-3 * (M ** 0.5), 1e-3, "bias grad (fused)")


def test_gemm_gelu_bwd_and_fp32_a(dev):
    from transformerupscaler_amd import ops, packing
    M, N, K = 192, 768, 192
    a, w, pre = rnd((M, K), 32), rnd((N, K), 33, 0.08), rnd((M, N), 34, 2.0)
    wp = packing.pack_linear(w).to(dev)
    prer = bf(pre).requires_grad_(True)
    torch.nn.functional.gelu(prer).backward(F.linear(bf(a), bf(w)))
    got = ops.gemm_tokens(a.to(dev), wp, None, "gelu_bwd", aux=pre.to(torch.bfloat16).to(dev))      # fp32 A
    close(got, prer.grad, 1e-2, 1e-2, "gelu bwd epilogue")
    close(ops.gemm_tokens(a.to(dev), wp, None, "bf16"), F.linear(bf(a), bf(w)), 1e-2, 1e-2, "fp32 A, no bias")


def test_layernorm_bwd(dev):
    from transformerupscaler_amd import ops
    M = 300
    x = rnd((M, 192), 35, 2.0, 0.3).requires_grad_(True)
    gm, bt = rnd((192,), 36, 0.1, 1.0).requires_grad_(True), rnd((192,), 37, 0.1).requires_grad_(True)
    gy, gres = bf(rnd((M, 192), 38)), rnd((M, 192), 39)
    F.layer_norm(x, (192,), gm, bt, 1e-5).backward(gy)
    _, mean, rstd = ops.layernorm(x.detach().to(dev), gm.detach().to(dev), bt.detach().to(dev), save_stats=True)
    dx, dg, db = ops.layernorm_bwd(gy.to(torch.bfloat16).to(dev), x.detach().to(dev), mean, rstd, gm.detach().to(dev), gres.to(dev))
    close(dx, x.grad + gres, 2e-5, 1e-4, "ln dx")
    close(dg, gm.grad, 2e-4, 1e-4, "ln dgamma")
    close(db, bt.grad, 2e-4, 1e-4, "ln dbeta")
    # the fused form also writes dx through the next Dropout's mask (what tup_dropout_bwd makes of dx): identical bits
    dx2, dg2, db2, gd = ops.layernorm_bwd(gy.to(torch.bfloat16).to(dev), x.detach().to(dev), mean, rstd, gm.detach().to(dev), gres.to(dev),
                                          drop=(0.1, 4242))
    assert torch.equal(dx2, dx) and torch.equal(gd, ops.dropout_bwd(dx, 0.1, 4242))
    assert (gd == 0).float().mean().item() > 0.05          # the mask drops ~10 %


def test_window_attention_bwd(dev, det_sd):
    from transformerupscaler_amd import ops
    nwin = 150          # > 128 slots: exercises the persistent loop
    qkv = bf(rnd((nwin, 64, 576), 40, 1.5)).requires_grad_(True)
    table = det_sd["window_blocks.1.attn.relative_position_bias_table"].clone().requires_grad_(True)
    idx = O.relative_position_index(8)
    q, k, v = qkv.view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    attn = (q * 0.25) @ k.transpose(-2, -1) + table[idx.view(-1)].view(64, 64, 12).permute(2, 0, 1).unsqueeze(0)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(nwin * 64, 192)
    gout = bf(rnd((nwin * 64, 192), 41))
    out.backward(gout)
    tb = table.detach().to(dev)
    qd = qkv.detach().view(-1, 576).to(torch.bfloat16).to(dev)
    att, lse = ops.window_attn(qd, ops.relpos_bias_expand(tb), save_lse=True)       # the backward rebuilds P from the saved row log-sum-exp
    ref_lse = torch.logsumexp(attn.detach(), dim=-1)                                # [nwin][12][64]
    assert (lse.cpu() - ref_lse).abs().max().item() <= 2e-2
    gqkv, dtable = ops.window_attn_bwd(qd, gout.to(torch.bfloat16).to(dev), att, lse, ops.relpos_bias_expand_n(tb))
    close(gqkv, qkv.grad.view(-1, 576), 2e-2, 2e-2, "attention dqkv")
    close(dtable, table.grad, 5e-2, 2e-2, "dtable")


@pytest.mark.parametrize("hw", [(20, 28), (68, 84), (64, 64)])
def test_patch_embed_unembed_bwd(dev, hw):
    from transformerupscaler_amd import ops, packing
    H, W = hw
    B = 2
    # ---- patch_embed backward: d feat (padded map) and d weight ----
    feat = bf(rnd((B, 64, H, W), 42)).requires_grad_(True)
    w = bf(rnd((192, 64, 8, 8), 43, 0.02)).requires_grad_(True)
    ph, pw = (8 - H % 8) % 8, (8 - W % 8) % 8
    fp = F.pad(feat, (0, pw, 0, ph), mode="reflect") if (ph or pw) else feat
    fp.retain_grad()
    tok = F.conv2d(fp, w, None, stride=8).permute(0, 2, 3, 1)
    ht, wt = tok.shape[1:3]
    pb, pr = (8 - ht % 8) % 8, (8 - wt % 8) % 8
    tokp = F.pad(tok.permute(0, 3, 1, 2), (0, pr, 0, pb)).permute(0, 2, 3, 1).contiguous()
    xw = O.window_partition(tokp, 8).reshape(-1, 192)
    gx = rnd(tuple(xw.shape), 44)
    xw.backward(gx)
    wt_bwd = packing.pack_linear(w.detach().permute(2, 3, 1, 0).reshape(4096, 192)).to(dev)     # rows (i,j,c), cols n
    gmap = ops.patch_embed_bwd(gx.to(dev), wt_bwd, B, H, W)
    ref_pad = fp.grad if (ph or pw) else feat.grad
    close(gmap, ref_pad.permute(0, 2, 3, 1), 1e-2, 1e-2, "patch_embed d(padded feat)")
    nhwc = feat.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    dw = ops.patch_wgrad(gx.to(dev), nhwc, reflect=True).cpu().view(192, 8, 8, 64).permute(0, 3, 1, 2)
    close(dw, w.grad, 3e-2, 2e-2, "patch_embed dW")
    # the wide-tile kernel (default) against the 64 x 64-tile one: same bf16 operands, fp32 sums in another order
    ops.PATCH_WGRAD_WIDE = False
    try:
        dw_narrow = ops.patch_wgrad(gx.to(dev), nhwc, reflect=True).cpu().view(192, 8, 8, 64).permute(0, 3, 1, 2)
    finally:
        ops.PATCH_WGRAD_WIDE = True
    close(dw, dw_narrow, 2e-3, 1e-4, "patch_embed dW, wide vs 64x64 tiles")

    # ---- patch_unembed backward: d tokens and d weight ----
    wu = bf(rnd((192, 64, 8, 8), 45, 0.05)).requires_grad_(True)
    xs = bf(rnd(tuple(xw.shape), 46)).requires_grad_(True)
    t = O.window_reverse(xs.view(B, -1, 64, 192), 8, ht + pb, wt + pr)[:, :ht, :wt, :].permute(0, 3, 1, 2)
    y = F.conv_transpose2d(t, wu, None, stride=8)[:, :, :H, :W]
    gy = bf(rnd(tuple(y.shape), 47))
    y.backward(gy)
    gnhwc = gy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    wt_ub = packing.pack_linear(wu.detach().permute(0, 2, 3, 1).reshape(192, 4096)).to(dev)      # rows k, cols (i,j,o)
    gxs = ops.patch_unembed_bwd(gnhwc, wt_ub)
    close(gxs, xs.grad, 2e-2, 1e-2, "patch_unembed d tokens")
    dwu = ops.patch_wgrad(xs.detach().to(dev), gnhwc, reflect=False).cpu().view(192, 8, 8, 64).permute(0, 3, 1, 2)
    close(dwu, wu.grad, 3e-2, 2e-2, "patch_unembed dW")


@pytest.mark.parametrize("hw,two", [((64, 64), True), ((24, 40), False), ((72, 128), True)])
def test_patch_embed_bwd_merge_equals_separate_passes(dev, hw, two):
    """tup_patch_embed_bwd_merge (patch_embed's input gradient with the gradient merge at `feat` + conv2's ReLU gate in its epilogue,
    and the column sums of add1 riding along) against the three separate passes it replaces: tup_patch_embed_bwd,
    tup_feat_grad_combine, tup_colsum.  The merged form keeps the GEMM value in fp32 where the separate one read it back as bf16."""
    from transformerupscaler_amd import ops, packing
    H, W = hw
    B = 2
    _, _, nwy, nwx = ops.window_geometry(H, W)
    gx = rnd((B * nwy * nwx * 64, 192), 81).to(dev)
    w = bf(rnd((192, 64, 8, 8), 82, 0.02))
    wt_bwd = packing.pack_linear(w.permute(2, 3, 1, 0).reshape(4096, 192)).to(dev)
    add1 = rnd((B, H, W, 64), 83).to(torch.bfloat16).to(dev)
    add2 = rnd((B, H, W, 64), 84).to(torch.bfloat16).to(dev) if two else None
    feat = rnd((B, H, W, 64), 85).to(torch.bfloat16).to(dev)          # about half of it <= 0: the ReLU gate
    ref = ops.feat_grad_combine(add1, add2, ops.patch_embed_bwd(gx, wt_bwd, B, H, W), feat).float()
    got, cs = ops.patch_embed_bwd_merge(gx, wt_bwd, add1, add2, feat, want_add1_colsum=True)
    assert torch.equal(got == 0, ref == 0) or ((got == 0) != (ref == 0)).float().mean().item() < 1e-3      # same gate
    close(got, ref, 2e-2, 1e-2, "merged gradient at feat")
    close(cs, ops.colsum(add1.view(-1, 64)), 5e-2, 2e-3, "column sums of add1")
    got2 = ops.patch_embed_bwd_merge(gx, wt_bwd, add1, add2, feat)
    assert torch.equal(got2, got)


def test_planar_conv_and_resize_write_gate_and_clamped_output_together(dev):
    """clamp01 = 2 of tup_conv3x3_planar_fwd / tup_resize_aa_fwd (training): [unclamped | clamped] from one pass == the two separate calls."""
    from transformerupscaler_amd import ops, packing
    x = (rnd((2, 3, 36, 48), 86) * 1.5 + 0.5).to(dev)
    w, b = rnd((3, 3, 3, 3), 87, 0.4), rnd((3,), 88, 0.2)
    w28 = packing.pack_planar(w).to(dev)
    add = rnd((2, 3, 36, 48), 89).to(dev)
    pre, out = ops.conv_planar(x, w28, b.to(dev), 1, add=add, clamp="both")
    assert torch.equal(pre, ops.conv_planar(x, w28, b.to(dev), 1, add=add, clamp=False))
    assert torch.equal(out, ops.conv_planar(x, w28, b.to(dev), 1, add=add, clamp=True))
    assert (out != pre).float().mean().item() > 0.05          # the clamp did something
    pre2, out2 = ops.resize_aa(x, (27, 36), clamp="both")
    assert torch.equal(pre2, ops.resize_aa(x, (27, 36), clamp=False)) and torch.equal(out2, ops.resize_aa(x, (27, 36), clamp=True))


# ------------------------------------------------------------------------------------------------
# conv-side backward kernels
# ------------------------------------------------------------------------------------------------
def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


@pytest.mark.parametrize("r,hw", [(1, (19, 45)), (2, (16, 40)), (3, (9, 33))])
def test_conv_c64_backward(dev, r, hw):
    from transformerupscaler_amd import ops, packing
    H, W = hw
    x = bf(rnd((2, 64, H, W), 50)).requires_grad_(True)
    w = bf(rnd((64 * r * r, 64, 3, 3), 51, 0.06)).requires_grad_(True)
    b = rnd((64 * r * r,), 52, 0.2).requires_grad_(True)
    y = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), r)
    gy = bf(rnd(tuple(y.shape), 53))
    y.backward(gy)
    gyd = nhwc(gy).to(dev)
    gx = ops.conv_c64(gyd, packing.pack_conv_c64_dgrad(w.detach(), r).to(dev), None, 1, in_r=r)
    close(gx, x.grad.permute(0, 2, 3, 1), 3e-2, 2e-2, f"conv dgrad r={r}")
    dwp, db = ops.conv_c64_wgrad(nhwc(x.detach()).to(dev), gyd, r)
    dw, dbb = packing.unpack_conv_c64_wgrad(dwp, db, r)
    close(dw, w.grad, 3e-2 * (H * W) ** 0.5 / 10, 2e-2, f"conv wgrad r={r}")
    close(dbb, b.grad, 2e-2, 1e-2, f"conv dbias r={r}")
    if r == 1:       # fused "+ add" and ReLU-backward mask epilogue
        add, z = bf(rnd(tuple(x.shape), 54)), bf(rnd(tuple(x.shape), 55))
        got = ops.conv_c64(gyd, packing.pack_conv_c64_dgrad(w.detach(), 1).to(dev), None, 1, add=nhwc(add).to(dev), mask=nhwc(z).to(dev))
        close(got, ((x.grad + add) * (z > 0)).permute(0, 2, 3, 1), 3e-2, 2e-2, "dgrad + add + mask")


def test_conv_thin_backward(dev):
    from transformerupscaler_amd import ops, packing
    x = bf(rnd((2, 64, 21, 50), 56)).requires_grad_(True)
    w = bf(rnd((3, 64, 3, 3), 57, 0.06)).requires_grad_(True)
    b = rnd((3,), 58, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1)
    gy = rnd(tuple(y.shape), 59)
    y.backward(gy)
    dwp, db = ops.conv_thin_wgrad(nhwc(x.detach()).to(dev), gy.to(dev), True)
    close(dwp.permute(0, 2, 1).reshape(3, 64, 3, 3), w.grad, 5e-2, 2e-2, "thin wgrad")
    close(db, b.grad, 1e-3, 1e-4, "thin dbias")
    z = bf(rnd(tuple(x.shape), 60))
    m = rnd(tuple(gy.shape), 61)
    gx = ops.conv1(gy.to(dev), packing.pack_conv_thin_dgrad(w.detach()).to(dev), None, relu=False)
    close(gx, x.grad.permute(0, 2, 3, 1), 2e-2, 2e-2, "thin dgrad")
    # masks: input *= (m > 0), output *= (z > 0)
    x.grad = None
    F.conv2d(x, w, b, padding=1).backward(gy * (m > 0))
    gx = ops.conv1(gy.to(dev), packing.pack_conv_thin_dgrad(w.detach()).to(dev), None, relu=False, in_mask=m.to(dev), out_mask=nhwc(z).to(dev))
    close(gx, (x.grad * (z > 0)).permute(0, 2, 3, 1), 2e-2, 2e-2, "thin dgrad masked")


def test_conv1_wgrad(dev):
    from transformerupscaler_amd import ops
    x = rnd((2, 3, 37, 70), 62, 0.5, 0.5)
    w = rnd((64, 3, 3, 3), 63, 0.3).requires_grad_(True)
    b = rnd((64,), 64, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1)
    gy = bf(rnd(tuple(y.shape), 65))
    y.backward(gy)
    dw, db = ops.conv1_wgrad_direct(x.to(dev), nhwc(gy).to(dev))            # fp32 VALU kernel
    close(dw, w.grad, 2e-3, 1e-3, "conv1 wgrad (direct)")
    close(db, b.grad, 2e-3, 1e-3, "conv1 dbias (direct)")
    dw, db = ops.conv1_wgrad(x.to(dev), nhwc(gy).to(dev))                   # MFMA path (x in bf16, as in the forward)
    # x enters the MFMA in bf16 (2^-9 relative): a sum over 5180 pixels of |g x| ~ 0.3 carries ~0.05 of rounding noise
    # whatever the size of the element, so the absolute tolerance is set from the scale of the sums (max |dw| ~ 25)
    close(dw, w.grad, 0.2, 5e-3, "conv1 wgrad")
    close(db, b.grad, 2e-3, 1e-3, "conv1 dbias")


@pytest.mark.parametrize("r", [1, 2, 3, 6])
def test_conv_planar_backward(dev, r):
    from transformerupscaler_amd import ops, packing
    x = rnd((2, 3, 23, 70), 66).requires_grad_(True)
    w = rnd((3 * r * r, 3, 3, 3), 67, 0.3).requires_grad_(True)
    b = rnd((3 * r * r,), 68, 0.2).requires_grad_(True)
    y = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), r)
    gy = rnd(tuple(y.shape), 69)
    y.backward(gy)
    dw, db = ops.conv_planar_wgrad(x.detach().to(dev), gy.to(dev), r)
    close(dw, w.grad, 2e-3, 1e-4, "planar wgrad")
    close(db, b.grad, 2e-3, 1e-4, "planar dbias")
    close(ops.conv_planar_dgrad(gy.to(dev), w.detach().to(dev), r), x.grad, 1e-5, 1e-5, "planar dgrad")
    if r == 1:
        close(ops.conv_planar(gy.to(dev), packing.pack_planar_dgrad(w.detach()).to(dev), None, 1), x.grad, 1e-5, 1e-5, "planar dgrad via fwd kernel")


@pytest.mark.parametrize("hw", [(23, 72), (5, 4), (9, 260), (33, 512)])
def test_conv_planar_dgrad_r2_four_pixels_per_thread(dev, hw):
    """The r = 2, W % 4 == 0 input-gradient kernel (four LR pixels per thread) against torch autograd through conv + PixelShuffle:
    image borders, one 4-pixel column, widths that are not a multiple of the 256-pixel block."""
    from transformerupscaler_amd import ops
    x = rnd((2, 3) + hw, 80).requires_grad_(True)
    w = rnd((12, 3, 3, 3), 81, 0.3)
    y = F.pixel_shuffle(F.conv2d(x, w, None, padding=1), 2)
    gy = rnd(tuple(y.shape), 82)
    y.backward(gy)
    close(ops.conv_planar_dgrad(gy.to(dev), w.to(dev), 2), x.grad, 1e-5, 1e-5, "planar dgrad r2x4")


@pytest.mark.parametrize("sizes", [((72, 96), (54, 72)), ((30, 40), (45, 47)), ((144, 256), (108, 192))])
def test_resize_and_clamp_backward(dev, sizes):
    from transformerupscaler_amd import ops
    (h, w), (oh, ow) = sizes
    x = rnd((2, 3, h, w), 70, 0.8, 0.5).requires_grad_(True)
    pre = O.aa_resize(x, (oh, ow))
    gy = rnd(tuple(pre.shape), 71)
    pre.clamp(0, 1).backward(gy)
    close(ops.resize_aa_bwd(gy.to(dev), (h, w), pre=pre.detach().to(dev)), x.grad, 3e-6, 1e-5, "resize+clamp bwd")
    x.grad = None
    O.aa_resize(x, (oh, ow)).backward(gy)
    close(ops.resize_aa_bwd(gy.to(dev), (h, w)), x.grad, 3e-6, 1e-5, "resize bwd")
    relu_src = rnd(tuple(gy.shape), 72)
    close(ops.mask_bwd(gy.to(dev), pre=pre.detach().to(dev), relu_src=relu_src.to(dev)),
          gy * ((pre >= 0) & (pre <= 1)) * (relu_src > 0), 0, 0, "mask bwd")


@pytest.mark.parametrize("hw", [(20, 28), (64, 64), (23, 41)])
def test_feat_grad_combine(dev, hw):
    from transformerupscaler_amd import ops
    H, W = hw
    B = 2
    hp, wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
    a, b, feat = bf(rnd((B, 64, H, W), 73)), bf(rnd((B, 64, H, W), 74)), bf(rnd((B, 64, H, W), 75))
    gpe = bf(rnd((B, 64, hp, wp), 76))
    f = feat.clone().requires_grad_(True)
    fp = F.pad(f, (0, wp - W, 0, hp - H), mode="reflect") if (hp > H or wp > W) else f
    fp.backward(gpe)
    ref = (a + b + f.grad) * (feat > 0)
    got = ops.feat_grad_combine(nhwc(a).to(dev), nhwc(b).to(dev), nhwc(gpe).to(dev), nhwc(feat).to(dev))
    close(got, ref.permute(0, 2, 3, 1), 2e-2, 1e-2, "feat grad combine")


@pytest.mark.parametrize("M", [128, 320, 64 * 7])
def test_fused_mlp(dev, M):
    """Inference fusion LN2 -> mlp.0 -> GELU -> mlp.2 -> +residual against torch and the unfused kernels."""
    from transformerupscaler_amd import ops, packing
    x = rnd((M, 192), 80, 2.0, 0.3)
    gm, bt = rnd((192,), 81, 0.1, 1.0), rnd((192,), 82, 0.1)
    w1, b1 = rnd((768, 192), 83, 0.08), rnd((768,), 84, 0.2)
    w2, b2 = rnd((192, 768), 85, 0.05), rnd((192,), 86, 0.2)
    y = bf(F.layer_norm(x, (192,), gm, bt, 1e-5))
    ref = x + F.linear(bf(F.gelu(F.linear(y, bf(w1), b1))), bf(w2), b2)
    w1p, w2p = packing.pack_linear(w1).to(dev), packing.pack_linear(w2).to(dev)
    w1q, b1q = packing.pack_fc1_fused_q(w1, b1)                 # mlp.0 / 4 (exact); mlp.2 as 4 W2 in fp16: the GELU runs in packed fp16
    got = ops.fused_mlp(x.to(dev).clone(), gm.to(dev), bt.to(dev), w1q.to(dev), b1q.to(dev), packing.pack_fc2_h4(w2).to(dev), b2.to(dev))
    close(got, ref, 2e-2, 1e-2, "fused mlp vs torch")
    # against torch with the hidden tile NOT rounded to bf16 (the fp16 tile is the more exact of the two): tighter
    ref16 = x + F.linear(F.gelu(F.linear(y, bf(w1), b1)), w2.half().float(), b2)
    close(got, ref16, 8e-3, 4e-3, "fused mlp vs torch (fp32 hidden, fp16 W2)")
    xu = x.to(dev).clone()
    yl = ops.layernorm(xu, gm.to(dev), bt.to(dev))
    hid = ops.gemm_tokens(yl, w1p, b1.to(dev), "gelu")
    ops.gemm_tokens(hid, w2p, b2.to(dev), "res", res=xu, out=xu)
    close(got, xu, 2e-2, 1e-2, "fused mlp vs unfused kernels (bf16 hidden tile there, fp16 here)")


@pytest.mark.parametrize("B,H,W,r,out_hw", [(2, 40, 72, 2, (60, 108)), (1, 37, 53, 2, (74, 106)), (1, 37, 53, 2, (55, 80)),
                                           (1, 24, 40, 3, (54, 90)), (1, 16, 20, 6, (70, 100)), (1, 90, 150, 2, (135, 225))])
def test_tail_fused_vs_unfused(dev, B, H, W, r, out_hw):
    """tup_tail_fused_fwd (last final_upscale stage + final_upscale_conv + "+ upscaled_input" + antialiased Resize +
    clamp, model.py:316-327) against the one-kernel-per-op HIP path: all fp32, so they agree to rounding."""
    from transformerupscaler_amd import ops, packing
    x = rnd((B, 3, H, W), 90, 0.4, 0.3).to(dev)
    wfu = packing.pack_planar(rnd((3 * r * r, 3, 3, 3), 91, 0.2)).to(dev)
    bfu = rnd((3 * r * r,), 92, 0.1).to(dev)
    wfc = packing.pack_planar(rnd((3, 3, 3, 3), 93, 0.2)).to(dev)
    bfc = rnd((3,), 94, 0.1).to(dev)
    ui = rnd((B, 3, H * r, W * r), 95, 0.3, 0.4).to(dev)
    got = ops.tail_fused(x, wfu, bfu, wfc, bfc, ui, r, out_hw, clamp=True)
    t1 = ops.conv_planar(x, wfu, bfu, r)
    same = tuple(out_hw) == (H * r, W * r)
    ref = ops.conv_planar(t1, wfc, bfc, 1, add=ui, clamp=same)
    if not same:
        ref = ops.resize_aa(ref, out_hw, clamp=True)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 2e-5, err
    assert 0.05 < ref.mean().item() < 0.95          # the clamp is not saturating the comparison away


@pytest.mark.parametrize("B,H,W,clamp", [(2, 40, 72, True), (1, 37, 53, False), (1, 1, 1, True), (1, 2, 3, False), (1, 45, 60, True),
                                         (1, 46, 61, True), (2, 91, 121, False), (1, 135, 58, True), (1, 5, 200, False)])
def test_tail_stream_r2_vs_torch_and_unfused(dev, B, H, W, clamp):
    """tup_tail_stream_r2_fwd (register-streaming output tail for a last stage of r = 2: Conv2d(3,12,3) + PixelShuffle(2),
    Conv2d(3,3,3), "+ upscaled_input" [+ clamp]; model.py:316-320,327, utils.py:62-63) against torch fp32 on the CPU and against
    the one-kernel-per-op HIP path.  Sizes cover one wave strip (60 LR columns) and one band (45 LR rows) exactly, one past
    them, several strips / bands with ragged ends, and the 1 x 1 image (every tap in the zero padding)."""
    from transformerupscaler_amd import ops, packing
    x = rnd((B, 3, H, W), 190, 0.4, 0.3)
    w_fu, b_fu = rnd((12, 3, 3, 3), 191, 0.2), rnd((12,), 192, 0.1)
    w_fc, b_fc = rnd((3, 3, 3, 3), 193, 0.2), rnd((3,), 194, 0.1)
    ui = rnd((B, 3, 2 * H, 2 * W), 195, 0.3, 0.4)
    ref = F.conv2d(F.pixel_shuffle(F.conv2d(x, w_fu, b_fu, padding=1), 2), w_fc, b_fc, padding=1) + ui
    if clamp:
        ref = ref.clamp(0, 1)
    got = ops.tail_stream_r2(x.to(dev), packing.pack_planar_t(w_fu).to(dev), b_fu.to(dev), packing.pack_planar_t(w_fc).to(dev),
                             b_fc.to(dev), ui.to(dev), clamp=clamp)
    assert got.shape == ref.shape
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 2e-5, err
    t1 = ops.conv_planar(x.to(dev), packing.pack_planar(w_fu).to(dev), b_fu.to(dev), 2)
    unf = ops.conv_planar(t1, packing.pack_planar(w_fc).to(dev), b_fc.to(dev), 1, add=ui.to(dev), clamp=clamp)
    assert (got - unf).abs().max().item() <= 2e-5
    if H * W > 100:
        assert 0.05 < ref.mean().item() < 0.95


@pytest.mark.parametrize("B,H,W,out_hw", [(2, 40, 72, (60, 108)), (1, 37, 53, (55, 80)), (1, 90, 150, (135, 225)), (1, 45, 60, (68, 90)),
                                          (2, 96, 130, (128, 174)), (1, 30, 200, (45, 300)), (1, 61, 59, (100, 100)),
                                          (1, 24, 40, (19, 32))])
def test_tail_stream_r2_fused_resize(dev, B, H, W, out_hw):
    """tup_tail_stream_r2_resize_fwd: the streaming tail with the antialiased Resize (model.py:323-325) and the clamp inside, against
    torch fp32 (F.interpolate antialias=True is what torchvision's tensor Resize calls) and against the streaming kernel followed
    by the separable Resize kernel.  Ratios 4/3 (the benchmark's), 1.36, 1.5 and a non-uniform one; several strips and bands;
    a down-scale by 2.5 (6 taps) must be refused by the planner (None) so that the caller falls back."""
    from transformerupscaler_amd import ops, packing
    x = rnd((B, 3, H, W), 290, 0.4, 0.3)
    w_fu, b_fu = rnd((12, 3, 3, 3), 291, 0.2), rnd((12,), 292, 0.1)
    w_fc, b_fc = rnd((3, 3, 3, 3), 293, 0.2), rnd((3,), 294, 0.1)
    ui = rnd((B, 3, 2 * H, 2 * W), 295, 0.3, 0.4)
    args = (x.to(dev), packing.pack_planar_t(w_fu).to(dev), b_fu.to(dev), packing.pack_planar_t(w_fc).to(dev), b_fc.to(dev), ui.to(dev))
    got = ops.tail_stream_r2(*args, clamp=True, out_hw=out_hw)
    if out_hw == (19, 32):                 # 48 -> 19 rows: scale 2.5, six taps
        assert got is None
        return
    assert got is not None and tuple(got.shape) == (B, 3) + tuple(out_hw)
    s = F.conv2d(F.pixel_shuffle(F.conv2d(x, w_fu, b_fu, padding=1), 2), w_fc, b_fc, padding=1) + ui
    ref = F.interpolate(s, size=out_hw, mode="bilinear", align_corners=False, antialias=True).clamp(0, 1)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 2e-5, err
    two = ops.resize_aa(ops.tail_stream_r2(*args, clamp=False), out_hw, clamp=True)
    assert (got - two).abs().max().item() <= 2e-5
    assert 0.05 < ref.mean().item() < 0.95


def test_ln_gemm_fused(dev):
    from transformerupscaler_amd import ops, packing
    M, N = 448, 576
    x = rnd((M, 192), 90, 2.0, 0.3)
    gm, bt = rnd((192,), 91, 0.1, 1.0), rnd((192,), 92, 0.1)
    w, b = rnd((N, 192), 93, 0.08), rnd((N,), 94, 0.2)
    ref = F.linear(bf(F.layer_norm(x, (192,), gm, bt, 1e-5)), bf(w), b)
    got = ops.ln_gemm(x.to(dev), gm.to(dev), bt.to(dev), packing.pack_linear(w).to(dev), b.to(dev))
    close(got, ref, 2e-2, 1e-2, "LN + GEMM panel kernel")


@pytest.mark.parametrize("nwin", [2, 5, 64])
def test_fused_qkv_attention(dev, nwin):
    """norm1 + qkv + window attention core in one kernel vs torch (fp32 on bf16-rounded operands)."""
    from transformerupscaler_amd import ops, packing
    g = torch.Generator().manual_seed(21)
    M = nwin * 64
    x = torch.randn((M, 192), generator=g)
    gm = 1 + 0.1 * torch.randn(192, generator=g); bt = 0.1 * torch.randn(192, generator=g)
    w = torch.randn((576, 192), generator=g) / 192 ** 0.5; b = 0.1 * torch.randn(576, generator=g)
    table = 0.5 * torch.randn((225, 12), generator=g)
    y = F.layer_norm(x, (192,), gm, bt, 1e-5).to(torch.bfloat16).float()
    qkv = (y @ w.to(torch.bfloat16).float().t() + b).view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    idx = (ys[:, None] - ys[None, :] + 7) * 15 + (xs[:, None] - xs[None, :] + 7)
    bias = table[idx.view(-1)].view(64, 64, 12).permute(2, 0, 1)
    attn = torch.softmax((qkv[0] * 0.25) @ qkv[1].transpose(-2, -1) + bias, dim=-1)
    ref = (attn @ qkv[2]).transpose(1, 2).reshape(M, 192)
    wh, bh = packing.pack_qkv_heads(w, b)
    frag = ops.relpos_bias_expand(table.to(dev))
    got = ops.fused_qkv_attn(x.to(dev), gm.to(dev), bt.to(dev), wh.to(dev), bh.to(dev), frag).float().cpu()
    err = (got - ref).abs().max().item()
    assert err <= 3e-2 + 2e-2 * ref.abs().max().item(), err


@pytest.mark.parametrize("nwin", [2, 5, 64])
def test_fused_attention_block(dev, nwin):
    """x += proj(attention(qkv(LN(x)))) + b in one kernel vs torch."""
    from transformerupscaler_amd import ops, packing
    g = torch.Generator().manual_seed(22)
    M = nwin * 64
    x = torch.randn((M, 192), generator=g)
    gm = 1 + 0.1 * torch.randn(192, generator=g); bt = 0.1 * torch.randn(192, generator=g)
    w = torch.randn((576, 192), generator=g) / 192 ** 0.5; b = 0.1 * torch.randn(576, generator=g)
    wp = torch.randn((192, 192), generator=g) / 192 ** 0.5; bp = 0.1 * torch.randn(192, generator=g)
    table = 0.5 * torch.randn((225, 12), generator=g)
    y = F.layer_norm(x, (192,), gm, bt, 1e-5).to(torch.bfloat16).float()
    qkv = (y @ w.to(torch.bfloat16).float().t() + b).view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    idx = (ys[:, None] - ys[None, :] + 7) * 15 + (xs[:, None] - xs[None, :] + 7)
    bias = table[idx.view(-1)].view(64, 64, 12).permute(2, 0, 1)
    attn = torch.softmax((qkv[0] * 0.25) @ qkv[1].transpose(-2, -1) + bias, dim=-1)
    att = (attn @ qkv[2]).transpose(1, 2).reshape(M, 192).to(torch.bfloat16).float()
    ref = x + att @ wp.to(torch.bfloat16).float().t() + bp
    wh, bh = packing.pack_qkv_heads(w, b)
    frag = ops.relpos_bias_expand(table.to(dev))
    got = ops.fused_attn_block(x.to(dev).clone(), gm.to(dev), bt.to(dev), wh.to(dev), bh.to(dev), frag,
                               packing.pack_proj_pairs(wp).to(dev), bp.to(dev)).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 3e-2 + 2e-2 * ref.abs().max().item(), err


@pytest.mark.parametrize("nwin", [1, 2, 5, 64])
def test_fused_block_equals_two_halves(dev, nwin):
    """tup_fused_block_fwd (whole WindowTransformerBlock, model.py:153-172, one kernel) against the attention-half kernel
    followed by the MLP-half kernel (which apply gamma / beta themselves): the same block to bf16 noise; and both against torch
    through the checks of the two halves above and test_block_vs_torch."""
    from transformerupscaler_amd import ops, packing
    g = torch.Generator().manual_seed(23)
    M = nwin * 64
    x = torch.randn((M, 192), generator=g).to(dev)
    gm1 = (1 + 0.1 * torch.randn(192, generator=g)).to(dev); bt1 = (0.1 * torch.randn(192, generator=g)).to(dev)
    gm2 = (1 + 0.1 * torch.randn(192, generator=g)).to(dev); bt2 = (0.1 * torch.randn(192, generator=g)).to(dev)
    w = torch.randn((576, 192), generator=g) / 192 ** 0.5; b = 0.1 * torch.randn(576, generator=g)
    wp = torch.randn((192, 192), generator=g) / 192 ** 0.5; bp = (0.1 * torch.randn(192, generator=g)).to(dev)
    w1 = torch.randn((768, 192), generator=g) * 0.08; b1 = (0.2 * torch.randn(768, generator=g)).to(dev)
    w2 = torch.randn((192, 768), generator=g) * 0.05; b2 = (0.2 * torch.randn(192, generator=g)).to(dev)
    table = 0.5 * torch.randn((225, 12), generator=g)
    wh, bh = packing.pack_qkv_heads(w, b)
    wh, bh = wh.to(dev), bh.to(dev)
    frag = ops.relpos_bias_expand(table.to(dev))
    wpp = packing.pack_proj_pairs(wp).to(dev)
    b1_raw = b1.cpu()
    w1f, b1 = packing.pack_fc1_fused_q(w1, b1_raw)
    w1f, b1, w2p = w1f.to(dev), b1.to(dev), packing.pack_fc2_h4(w2).to(dev)
    two = ops.fused_attn_block(x.clone(), gm1, bt1, wh, bh, frag, wpp, bp)
    two = ops.fused_mlp(two, gm2, bt2, w1f, b1, w2p, b2)
    # the whole-block kernel takes the LayerNorms folded into the Linears (same function of x; the bf16 roundings fall on
    # xhat and W gamma instead of on LN(x) and W): agreement to the bf16 noise of one block, not bit for bit
    whn, bhn = packing.pack_qkv_heads(*packing.fold_layernorm(w, b, gm1.cpu(), bt1.cpu()))
    w1n, b1n = packing.pack_fc1_fused_q(*packing.fold_layernorm(w1, b1_raw, gm2.cpu(), bt2.cpu()))
    one = ops.fused_block(x.clone(), whn.to(dev), bhn.to(dev), frag, wpp, bp, w1n.to(dev), b1n.to(dev), w2p, b2)
    assert torch.isfinite(one).all()
    d = (one - two).abs()
    print(f"whole block (folded LayerNorms) vs two halves: max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.max().item() <= 4e-2 and d.mean().item() <= 6e-3, (d.max().item(), d.mean().item())      # seen: 2.5e-2, 3.5e-3
    assert (one - x).abs().max().item() > 0.1            # the block did something


def _block_operands(dev, nwin, seed=23):
    from transformerupscaler_amd import ops, packing
    g = torch.Generator().manual_seed(seed)
    M = nwin * 64
    raw = dict(x=torch.randn((M, 192), generator=g),
               gm1=1 + 0.1 * torch.randn(192, generator=g), bt1=0.1 * torch.randn(192, generator=g),
               gm2=1 + 0.1 * torch.randn(192, generator=g), bt2=0.1 * torch.randn(192, generator=g),
               w=torch.randn((576, 192), generator=g) / 192 ** 0.5, b=0.1 * torch.randn(576, generator=g),
               wp=torch.randn((192, 192), generator=g) / 192 ** 0.5, bp=0.1 * torch.randn(192, generator=g),
               w1=torch.randn((768, 192), generator=g) * 0.08, b1=0.2 * torch.randn(768, generator=g),
               w2=torch.randn((192, 768), generator=g) * 0.05, b2=0.2 * torch.randn(192, generator=g),
               table=0.5 * torch.randn((225, 12), generator=g))
    # the whole-block kernels take norm1 / norm2 folded into attn.qkv / mlp.0 (packing.fold_layernorm)
    wh, bh = packing.pack_qkv_heads(*packing.fold_layernorm(raw["w"], raw["b"], raw["gm1"], raw["bt1"]))
    args = [wh.to(dev), bh.to(dev), ops.relpos_bias_expand(raw["table"].to(dev)),
            packing.pack_proj_pairs(raw["wp"]).to(dev), raw["bp"].to(dev),
            *[t.to(dev) for t in packing.pack_fc1_fused_q(*packing.fold_layernorm(raw["w1"], raw["b1"], raw["gm2"], raw["bt2"]))],
            packing.pack_fc2_h4(raw["w2"]).to(dev), raw["b2"].to(dev)]
    return raw, args


def _block_torch(raw, nwin):
    """WindowTransformerBlock.forward (model.py:153-172) in torch fp32 on bf16-rounded GEMM operands."""
    x = raw["x"]
    M = nwin * 64
    y = bf(F.layer_norm(x, (192,), raw["gm1"], raw["bt1"], 1e-5))
    qkv = (y @ bf(raw["w"]).t() + raw["b"]).view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    idx = (ys[:, None] - ys[None, :] + 7) * 15 + (xs[:, None] - xs[None, :] + 7)
    bias = raw["table"][idx.view(-1)].view(64, 64, 12).permute(2, 0, 1)
    attn = torch.softmax(bf(qkv[0] * 0.25) @ bf(qkv[1]).transpose(-2, -1) + bias, dim=-1)
    att = bf((bf(attn) @ bf(qkv[2])).transpose(1, 2).reshape(M, 192))
    x1 = x + att @ bf(raw["wp"]).t() + raw["bp"]
    y2 = bf(F.layer_norm(x1, (192,), raw["gm2"], raw["bt2"], 1e-5))
    return x1 + F.linear(bf(F.gelu(F.linear(y2, bf(raw["w1"]), raw["b1"]))), bf(raw["w2"]), raw["b2"])


@pytest.mark.parametrize("nwin", [1, 3, 4, 5, 64, 1920])
def test_block_vs_torch(dev, nwin):
    """tup_fused_block_fwd (whole WindowTransformerBlock, model.py:153-172) against torch fp32 on bf16-rounded GEMM operands.
    1920 windows = what one launch of BASELINE configs[1] processes (8 x 240); 1, 3, 5 exercise the inactive-wave paths of the
    2-window workgroup."""
    from transformerupscaler_amd import ops
    raw, args = _block_operands(dev, nwin)
    x = raw["x"].to(dev)
    b32 = ops.fused_block(x.clone(), *args)
    assert torch.isfinite(b32).all()
    ref = _block_torch(raw, nwin)
    e32 = (b32.cpu() - ref).abs().max().item()
    print(f"nwin {nwin}: vs torch {e32:.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert e32 <= 3e-2 + 1e-2 * ref.abs().max().item(), e32
    assert (b32 - x).abs().max().item() > 0.1            # the block did something


def test_blocks_six_in_one_launch_equals_six_launches(dev):
    """tup_fused_blocks32_fwd with nblk = 6 (the model's loop, model.py:288-289, in one launch; the residual stream carried in
    registers between blocks) against six single-block launches: same kernel body, identical results."""
    from transformerupscaler_amd import ops
    nwin = 37
    ops_ = [_block_operands(dev, nwin, seed=40 + i)[1] for i in range(3)]
    x = _block_operands(dev, nwin, seed=50)[0]["x"].to(dev)
    seq = [ops_[i % 3] for i in range(6)]
    one32 = ops.fused_blocks32(x.clone(), ops.block_table([tuple(a) for a in seq]))
    ref32 = x.clone()
    for a in seq:
        ops.fused_block(ref32, *a)
    assert torch.isfinite(one32).all()
    assert torch.equal(one32, ref32)


def test_blocks_one_window_per_workgroup_equals_two(dev):
    """Launches that would not fill the chip's workgroup slots run the whole-block kernel with ONE window per workgroup (16 token rows
    per wave, csrc/fused_attn.hip TGN = 1; 37 windows here), large ones with two (TGN = 2; 800 windows here).  A window never meets
    another window, so the first 37 windows of the large launch must come out bit-identical to the small launch."""
    from transformerupscaler_amd import ops
    small, large = 37, 800
    seq = [_block_operands(dev, 1, seed=60 + i)[1] for i in range(2)] * 3
    x = _block_operands(dev, large, seed=70)[0]["x"].to(dev)
    table = ops.block_table([tuple(a) for a in seq])
    big = ops.fused_blocks32(x.clone(), table)
    sm = ops.fused_blocks32(x[:small * 64].clone(), table)
    assert torch.isfinite(big).all()
    assert torch.equal(sm, big[:small * 64])


@pytest.mark.parametrize("B,H,W", [(2, 8, 32), (1, 13, 37), (2, 24, 70), (1, 6, 6), (1, 9, 113), (3, 19, 28)])
def test_branch_a_composed_training(dev, B, H, W):
    """csrc/branch_a_train.hip: forward of the composed branch A and its whole backward (input gradient incl. the ring, the
    gradient w.r.t. the composed weights, the chain rule to W_up / b_up / W_3) against torch autograd through the explicit
    chain conv 64->256 -> PixelShuffle(2) -> conv 64->3 (no bias) -> ReLU (model.py:264-265)."""
    from transformerupscaler_amd import ops
    g_ = torch.Generator().manual_seed(31 + H)
    feat = bf(torch.randn((B, 64, H, W), generator=g_) * 0.7).requires_grad_(True)
    wu = (torch.randn((256, 64, 3, 3), generator=g_) * 0.04).requires_grad_(True)
    bu = (torch.randn((256,), generator=g_) * 0.1).requires_grad_(True)
    w3 = (torch.randn((3, 64, 3, 3), generator=g_) * 0.04).requires_grad_(True)
    pre = F.conv2d(F.pixel_shuffle(F.conv2d(feat, wu, bu, padding=1), 2), w3, None, padding=1)
    ui = F.relu(pre)
    gout = torch.randn(ui.shape, generator=g_)
    ui.backward(gout)
    comp = ops.bra_compose(wu.detach().to(dev), bu.detach().to(dev), w3.detach().to(dev))
    fnhwc = feat.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    got_ui = ops.branch_a_composed(fnhwc, comp["wp"], comp["bias"], comp["wv"], comp["bv"], 2)
    close(got_ui, ui.detach(), 3e-2, 2e-2, "composed forward (training weights)")
    # the backward is evaluated at the REFERENCE's ReLU gates so that only the kernels are compared
    dfeat, dwu, dbu, dw3, G, Gb = ops.bra_backward(gout.to(dev), ui.detach().to(dev), fnhwc, comp, wu.detach().to(dev), bu.detach().to(dev), w3.detach().to(dev))
    def rel(a, b):
        return ((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-12)).item()
    r_f = rel(dfeat.permute(0, 3, 1, 2), feat.grad)
    r_u, r_b, r_3 = rel(dwu, wu.grad), rel(dbu, bu.grad), rel(dw3, w3.grad)
    print(f"B{B} {H}x{W}: rel L2 dfeat {r_f:.4f} dW_up {r_u:.4f} db_up {r_b:.4f} dW_3 {r_3:.4f}")
    # ring rows / columns of d feat separately (they come from the variant kernels)
    frame = torch.ones((H, W), dtype=torch.bool); frame[3:-3, 3:-3] = False
    fr = rel(dfeat.permute(0, 3, 1, 2)[..., frame], feat.grad[..., frame])
    assert max(r_f, fr) <= 2e-2, (r_f, fr)
    assert r_u <= 2e-2 and r_b <= 2e-2 and r_3 <= 2e-2, (r_u, r_b, r_3)


def _stream_operands(dev, raw):
    from transformerupscaler_amd import packing
    return tuple(t.to(dev) for t in packing.pack_stream_block(raw["w"], raw["b"], raw["gm1"], raw["bt1"], raw["table"], raw["wp"], raw["bp"],
                                                              raw["w1"], raw["b1"], raw["gm2"], raw["bt2"], raw["w2"], raw["b2"]))


@pytest.mark.parametrize("nwin", [1, 5, 64, 540])
def test_stream_block_bf16_tokens(dev, nwin):
    """tup_blocks_stream_fwd with out_bf16: the result as bf16 tokens = round-to-nearest-even of the in-place fp32 result, bit for
    bit (1 and 5 windows: inactive waves of the four-window workgroup must not write); and tup_patch_unembed_fwd (model.py:292-309)
    reading those bf16 tokens = the same output as reading the fp32 ones (its GEMM rounds them on load)."""
    from transformerupscaler_amd import ops, packing
    raw, _ = _block_operands(dev, nwin)
    x = raw["x"].to(dev)
    tab = ops.stream_table([_stream_operands(dev, raw)])
    want = ops.blocks_stream(x.clone(), tab)
    guard = torch.full((64, 192), 7.0, dtype=torch.bfloat16, device=dev)
    got = ops.blocks_stream(x.clone(), tab, out_bf16=True)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (nwin * 64, 192)
    assert torch.equal(got, want.to(torch.bfloat16))
    assert (guard == 7.0).all()
    if nwin == 540:                                           # 4 images of 540 x 960 / 8 = 68 x 120 tokens -> 9 x 15 windows
        B, H, W = 4, 540, 960
        g = torch.Generator().manual_seed(5)
        wu = torch.randn((192, 64, 8, 8), generator=g) * 0.05
        bu = torch.randn(64, generator=g) * 0.1
        skip = torch.randn((B, H, W, 64), generator=g).to(torch.bfloat16).to(dev)
        wt = packing.pack_patch_unembed(wu).to(dev)
        a = ops.patch_unembed(want, wt, bu.to(dev), skip)
        b = ops.patch_unembed(got, wt, bu.to(dev), skip)
        assert torch.equal(a, b)


@pytest.mark.parametrize("nwin", [1, 3, 4, 5, 64, 1920])
def test_stream_block_vs_torch(dev, nwin):
    """tup_blocks_stream_fwd (the streamed 32x32x16 whole-block kernel, model.py:153-172) with one block against torch fp32 on
    bf16-rounded GEMM operands, and against the 16x16x32 whole-block kernel; 1, 3, 5 exercise the inactive waves of the four-window
    workgroup, 1920 = one launch of BASELINE configs[1]."""
    from transformerupscaler_amd import ops
    raw, args = _block_operands(dev, nwin)
    x = raw["x"].to(dev)
    got = ops.blocks_stream(x.clone(), ops.stream_table([_stream_operands(dev, raw)]))
    assert torch.isfinite(got).all()
    ref = _block_torch(raw, nwin)
    e = (got.cpu() - ref).abs().max().item()
    b32 = ops.fused_block(x.clone(), *args)
    d = (got - b32).abs()
    print(f"nwin {nwin}: stream vs torch {e:.3e} (|ref| max {ref.abs().max().item():.2f}); vs the 16x16x32 kernel max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert e <= 3e-2 + 1e-2 * ref.abs().max().item(), e
    assert d.max().item() <= 4e-2 and d.mean().item() <= 6e-3
    assert (got - x).abs().max().item() > 0.1


def test_stream_blocks_six_in_one_launch(dev):
    """Six blocks (three weight sets, cycled) in one launch of the streamed kernel against six one-block launches of it (the residual
    stream carried in registers across blocks, the LDS regions handed from block to block) and against the 16x16x32 kernel."""
    from transformerupscaler_amd import ops
    nwin = 37
    sets = [_block_operands(dev, nwin, seed=40 + i) for i in range(3)]
    st = [_stream_operands(dev, s[0]) for s in sets]
    x = _block_operands(dev, nwin, seed=50)[0]["x"].to(dev)
    one = ops.blocks_stream(x.clone(), ops.stream_table([st[i % 3] for i in range(6)]))
    seq = x.clone()
    for i in range(6):
        ops.blocks_stream(seq, ops.stream_table([st[i % 3]]))
    assert torch.isfinite(one).all()
    assert torch.equal(one, seq)
    ref = ops.fused_blocks32(x.clone(), ops.block_table([tuple(sets[i % 3][1]) for i in range(6)]))
    d = (one - ref).abs()
    print(f"six blocks: stream vs 16x16x32 kernel max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.max().item() <= 0.15 and d.mean().item() <= 1.5e-2
