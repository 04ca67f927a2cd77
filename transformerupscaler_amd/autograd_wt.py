"""Training path of the WindowTransformer plugin: forward that keeps what the hand-written backward needs, and the
backward itself -- every gradient the reference gets from ``loss.backward()`` through models/WindowTransformer/model.py:
225-305, computed by the HIP kernels of include/tupscale_hip.h.  Structure = autograd_rt.py's shell (bicubic, stride-2 conv,
decoder) around autograd.py's window blocks at width 128 / 8 heads; torch.autograd sees one node per model call.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops, packing
from .autograd import site_seed
from .window_transformer import pad_to_even


def _token_rowmask(B, ht, wt, device):
    """uint8 [M]: 1 for rows of the window-layout token matrix that are real tokens (not the zero pad, model.py:256-263)."""
    nwy, nwx = (ht + 7) // 8, (wt + 7) // 8
    ty = (torch.arange(nwy).view(-1, 1, 1, 1) * 8 + torch.arange(8).view(1, 1, -1, 1))
    tx = (torch.arange(nwx).view(1, -1, 1, 1) * 8 + torch.arange(8).view(1, 1, 1, -1))
    m = ((ty < ht) & (tx < wt)).expand(nwy, nwx, 8, 8).reshape(1, -1).expand(B, -1).reshape(-1)
    return m.to(torch.uint8).contiguous().to(device)


def forward_train(pk, frags_t, heads, x, res_out, drop_p: float, seed: int):
    B, _, H, W = x.shape
    x = x.contiguous().float()
    hd, wd = (H + 1) // 2, (W + 1) // 2
    hs, ws = (hd // 8) * 8, (wd // 8) * 8
    sv = {"x": x, "drop_p": drop_p, "seed": seed, "heads": heads}
    feat1 = ops.conv1(x, pk["conv1.w"], pk["conv1.b"], relu=True)
    feat = pad_to_even(ops.conv_c64(feat1, pk["conv2.w"], pk["conv2.b"], 1, relu=True))       # odd sizes: + one zero row / column
    feat_down = ops.conv_c64(feat, pk["ds.w"], pk["ds.b"], 1, relu=False, in_r=2)
    skip = feat_down if (hs, ws) == (hd, wd) else feat_down[:, :hs, :ws, :].contiguous()
    sv["feat1"], sv["feat"], sv["feat_down"], sv["skip"] = feat1, feat, feat_down, skip
    xw = ops.wt_patch_embed(feat_down, pk["pe.w"], pk["pe.b"])
    blocks = []
    for i in range(pk["nblocks"]):
        s = {"x_in": xw}
        y1, s["mean1"], s["rstd1"] = ops.layernorm128(xw, pk[f"b{i}.norm1.w"], pk[f"b{i}.norm1.b"], save_stats=True)
        qkv = ops.gemm_tokens(y1, pk[f"b{i}.qkv.w"], pk[f"b{i}.qkv.b"], "bf16")
        att, s["lse"] = ops.window_attn_h(qkv, frags_t[i], heads, drop_p, site_seed(seed, i, 0), save_lse=True)
        x_mid = ops.gemm_tokens(att, pk[f"b{i}.proj.w"], pk[f"b{i}.proj.b"], "res", res=xw,
                                drop_p=drop_p, drop_seed=site_seed(seed, i, 1))
        y2, s["mean2"], s["rstd2"] = ops.layernorm128(x_mid, pk[f"b{i}.norm2.w"], pk[f"b{i}.norm2.b"], save_stats=True)
        hpre = torch.empty((y2.shape[0], 512), dtype=torch.bfloat16, device=x.device)
        hid = ops.gemm_tokens(y2, pk[f"b{i}.fc1.w"], pk[f"b{i}.fc1.b"], "gelu", aux=hpre)
        xw = ops.gemm_tokens(hid, pk[f"b{i}.fc2.w"], pk[f"b{i}.fc2.b"], "res", res=x_mid,
                             drop_p=drop_p, drop_seed=site_seed(seed, i, 2))
        s.update(y1=y1, qkv=qkv, att=att, x_mid=x_mid, y2=y2, hpre=hpre, hid=hid)
        blocks.append(s)
    sv["blocks"], sv["xw_out"] = blocks, xw
    comb = ops.wt_patch_unembed(xw, pk["pu.w"], pk["pu.b"], skip)
    dec = ops.conv_c64(comb, pk["dec1.w"], pk["dec1.b"], 1, relu=True)
    residual = ops.conv_c64_thin(dec, pk["dec2.w"], pk["dec2.b"], 3, relu=False)
    out = ops.rt_bicubic_sum(x, residual, tuple(int(v) for v in res_out), clamp=True)
    sv["comb"], sv["dec"], sv["out"] = comb, dec, out
    return out, sv


def backward_train(pk, frags_t, frags_n, sv, gout, reducer=None) -> Dict[str, torch.Tensor]:
    g: Dict[str, torch.Tensor] = {}

    def ready(*names):
        if reducer is not None:
            reducer.on_ready(list(names), g)

    x, heads = sv["x"], sv["heads"]
    B, _, H, W = x.shape
    hd, wd = (H + 1) // 2, (W + 1) // 2
    hs, ws = sv["skip"].shape[1], sv["skip"].shape[2]
    gout = gout.contiguous().float()
    g_res = ops.rt_bicubic_bwd(gout, sv["out"], (hs, ws))
    dwp, db = ops.conv_thin_wgrad(sv["dec"], g_res, True)
    g["decoder_conv2.weight"], g["decoder_conv2.bias"] = dwp.permute(0, 2, 1).reshape(3, 64, 3, 3), db
    g_dec = ops.conv1(g_res, pk["dec2.wd"], None, relu=False, out_mask=sv["dec"])
    ready("decoder_conv2.weight", "decoder_conv2.bias")
    dwp, db = ops.conv_c64_wgrad(sv["comb"], g_dec, 1)
    g["decoder_conv1.weight"], g["decoder_conv1.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_comb = ops.conv_c64(g_dec, pk["dec1.wd"], None, 1)
    del g_dec
    ready("decoder_conv1.weight", "decoder_conv1.bias")
    # ---- patch_unembed (+ cropped skip) ----
    g["patch_unembed.bias"] = ops.colsum(g_comb.view(-1, 64))
    g["patch_unembed.weight"] = ops.wt_patch_wgrad(sv["xw_out"], g_comb).view(128, 8, 8, 64).permute(0, 3, 1, 2)
    g_x = ops.wt_patch_unembed_bwd(g_comb, pk["pu.wd"])
    ready("patch_unembed.weight", "patch_unembed.bias")
    # ---- window blocks (reverse) ----
    drop_p, seed = sv["drop_p"], sv["seed"]
    g_xd = None
    for i in reversed(range(pk["nblocks"])):
        s, p = sv["blocks"][i], f"window_blocks.{i}"
        # gradient entering mlp.2's output: through the MLP dropout mask (the residual path keeps g_x itself); from the second
        # block of the loop on the previous LayerNorm1 backward has written it already (fused dropout_bwd)
        if g_xd is not None:
            g_o, g_xd = g_xd, None
        else:
            g_o = ops.dropout_bwd(g_x, drop_p, site_seed(seed, i, 2)) if drop_p > 0 else g_x
        g[p + ".mlp.2.weight"], g[p + ".mlp.2.bias"] = ops.gemm_wgrad_bias(g_o, s["hid"])
        g_h = ops.gemm_tokens(g_o, pk[f"b{i}.fc2.wd"], None, "gelu_bwd", aux=s["hpre"])
        del g_o
        g[p + ".mlp.0.weight"], g[p + ".mlp.0.bias"] = ops.gemm_wgrad_bias(g_h, s["y2"])
        g_y2 = ops.gemm_tokens(g_h, pk[f"b{i}.fc1.wd"], None, "bf16")
        del g_h
        if drop_p > 0:          # + proj_drop's backward of the result (bf16), in the same pass
            g_xm, g[p + ".norm2.weight"], g[p + ".norm2.bias"], g_o = ops.layernorm128_bwd(
                g_y2, s["x_mid"], s["mean2"], s["rstd2"], pk[f"b{i}.norm2.w"], gres=g_x, drop=(drop_p, site_seed(seed, i, 1)))
        else:
            g_xm, g[p + ".norm2.weight"], g[p + ".norm2.bias"] = ops.layernorm128_bwd(
                g_y2, s["x_mid"], s["mean2"], s["rstd2"], pk[f"b{i}.norm2.w"], gres=g_x)
            g_o = g_xm
        g[p + ".attn.proj.weight"], g[p + ".attn.proj.bias"] = ops.gemm_wgrad_bias(g_o, s["att"])
        g_att = ops.gemm_tokens(g_o, pk[f"b{i}.proj.wd"], None, "bf16")
        del g_o
        g_qkv, g[p + ".attn.relative_position_bias_table"] = ops.window_attn_bwd_h(
            s["qkv"], g_att, s["att"], s["lse"], frags_n[i], heads, drop_p, site_seed(seed, i, 0))
        g[p + ".attn.qkv.weight"], g[p + ".attn.qkv.bias"] = ops.gemm_wgrad_bias(g_qkv, s["y1"])
        g_y1 = ops.gemm_tokens(g_qkv, pk[f"b{i}.qkv.wd"], None, "bf16")
        del g_qkv, g_att
        if drop_p > 0 and i > 0:          # + the MLP dropout's backward for the block below
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"], g_xd = ops.layernorm128_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm, drop=(drop_p, site_seed(seed, i - 1, 2)))
        else:
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"] = ops.layernorm128_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm)
        ready(*[p + sfx for sfx in (".mlp.2.bias", ".mlp.2.weight", ".mlp.0.bias", ".mlp.0.weight", ".norm2.weight",
                                    ".norm2.bias", ".attn.proj.bias", ".attn.proj.weight",
                                    ".attn.relative_position_bias_table", ".attn.qkv.bias", ".attn.qkv.weight",
                                    ".norm1.weight", ".norm1.bias")])
    # ---- patch_embed (real tokens only: the zero pad carries no bias, model.py:256-263) ----
    g["patch_embed.bias"] = ops.colsum(g_x, rowmask=_token_rowmask(B, hd // 8, wd // 8, x.device))
    g["patch_embed.weight"] = ops.wt_patch_wgrad(g_x, sv["feat_down"]).view(128, 8, 8, 64).permute(0, 3, 1, 2)
    g_fd = ops.wt_patch_embed_bwd(g_x, pk["pe.wd"], add=g_comb)                  # + skip gradient, on the cropped map
    del g_x, g_comb
    if (hs, ws) != (hd, wd):          # rows / columns the stride-8 conv and the crop never read get no gradient
        full = torch.zeros((B, hd, wd, 64), dtype=g_fd.dtype, device=g_fd.device)
        full[:, :hs, :ws, :] = g_fd
        g_fd = full
    ready("patch_embed.weight", "patch_embed.bias")
    # ---- downsample (stride-2 conv), conv2, conv1 ----
    dwp, db = ops.conv_c64_wgrad_s2d(sv["feat"], g_fd, 2)
    g["downsample.weight"], g["downsample.bias"] = packing.unpack_conv_c64_stride2_wgrad(dwp), db
    g_feat = ops.conv_c64(g_fd, pk["ds.wd"], None, 2, mask=sv["feat"])
    del g_fd
    if g_feat.shape[1:3] != sv["feat1"].shape[1:3]:          # odd input size: drop the zero row / column pad_to_even appended
        g_feat = g_feat[:, :sv["feat1"].shape[1], :sv["feat1"].shape[2], :].contiguous()
    ready("downsample.weight", "downsample.bias")
    dwp, db = ops.conv_c64_wgrad(sv["feat1"], g_feat, 1)
    g["conv2.weight"], g["conv2.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_f1 = ops.conv_c64(g_feat, pk["conv2.wd"], None, 1, mask=sv["feat1"])
    g["conv1.weight"], g["conv1.bias"] = ops.conv1_wgrad(x, g_f1)
    ready("conv2.weight", "conv2.bias", "conv1.weight", "conv1.bias")
    return g


class _WindowTransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, res_out, names, *params):
        pk, frags_t, frags_n = module.packed(backward=True)
        drop_p, seed = module._next_dropout()
        out, sv = forward_train(pk, frags_t, module.num_heads, x, res_out, drop_p, seed)
        ctx.module, ctx.names, ctx.sv, ctx.pk, ctx.frags = module, names, sv, pk, (frags_t, frags_n)
        return out

    @staticmethod
    def backward(ctx, gout):
        reducer = getattr(ctx.module, "_grad_reducer", None)
        if reducer is not None:
            reducer.begin(ctx.names)          # raises if this step's parameters are not in the reducer's layout
        ops.zero_pool_begin(gout.device)
        try:
            grads = backward_train(ctx.pk, ctx.frags[0], ctx.frags[1], ctx.sv, gout, reducer)
        except BaseException:
            if reducer is not None:
                reducer._abort()
            raise
        finally:
            ops.zero_pool_end()
        if reducer is not None:
            grads = reducer.finish()
        ctx.sv = None
        outs = []
        for n in ctx.names:
            gr = grads.get(n)
            outs.append(None if gr is None else gr.contiguous())      # reducer: views of this episode's own flat buffer (dp.py)
        return (None, None, None, None) + tuple(outs)


def window_transformer_function(module, x, res_out):
    named = dict(module.named_parameters())
    names = [n for n, p in named.items() if p.requires_grad]
    return _WindowTransformerFn.apply(module, x, tuple(res_out), names, *[named[n] for n in names])
