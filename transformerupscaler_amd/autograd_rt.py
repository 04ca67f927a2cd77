"""Training path of the ResidualTransformer plugin (BASELINE.json config 5): forward that keeps what the hand-written
backward needs, and the backward itself -- every gradient the reference gets from ``loss.backward()`` through
models/ResidualTransformer/model.py:121-165 (train.py:138), computed by the HIP kernels of include/tupscale_hip.h.

Same structure as autograd.py (FastTransformer): torch.autograd sees one node per model call.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops, packing
from .autograd import site_seed

# Optional timing hook (bench.py --mode rt installs one): callable(name) -> context manager around the attention launches
# ("rt_attn_fwd" = rt_attention_kernel, "rt_attn_bwd" = the dq + dkv pair).
stage_timer = None


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _stage(name):
    return stage_timer(name) if stage_timer is not None else _Null()


def forward_train(pk, x, res_out, drop_p: float, seed: int):
    B, _, H, W = x.shape
    x = x.contiguous().float()
    sv = {"x": x, "drop_p": drop_p, "seed": seed}
    feat1 = ops.conv1(x, pk["conv1.w"], pk["conv1.b"], relu=True)
    feat = ops.conv_c64(feat1, pk["conv2.w"], pk["conv2.b"], 1, relu=True)
    feat_down = ops.conv_c64(feat, pk["ds.w"], pk["ds.b"], 1, relu=False, in_r=2)
    sv["feat1"], sv["feat"], sv["feat_down"] = feat1, feat, feat_down
    xw = ops.rt_patch_embed(feat_down, pk["pe.w"], pk["pe.b"], pk["pos"])
    N = xw.shape[0] // B
    blocks = []
    for i in range(pk["nblocks"]):
        s = {"x_in": xw}
        y1, s["mean1"], s["rstd1"] = ops.layernorm128(xw, pk[f"b{i}.norm1.w"], pk[f"b{i}.norm1.b"], save_stats=True)
        qkv = ops.gemm_tokens(y1, pk[f"b{i}.in.w"], pk[f"b{i}.in.b"], "bf16")
        with _stage("rt_attn_fwd"):
            att, lse = ops.rt_attention(qkv, B, N, save_lse=True, drop_p=drop_p, drop_seed=site_seed(seed, i, 0))
        x_mid = ops.gemm_tokens(att, pk[f"b{i}.out.w"], pk[f"b{i}.out.b"], "res", res=xw)
        y2, s["mean2"], s["rstd2"] = ops.layernorm128(x_mid, pk[f"b{i}.norm2.w"], pk[f"b{i}.norm2.b"], save_stats=True)
        hpre = torch.empty((y2.shape[0], 512), dtype=torch.bfloat16, device=x.device)
        hid = ops.gemm_tokens(y2, pk[f"b{i}.fc1.w"], pk[f"b{i}.fc1.b"], "gelu", aux=hpre)
        xw = ops.gemm_tokens(hid, pk[f"b{i}.fc2.w"], pk[f"b{i}.fc2.b"], "res", res=x_mid,
                             drop_p=drop_p, drop_seed=site_seed(seed, i, 2))
        s.update(y1=y1, qkv=qkv, att=att, lse=lse, x_mid=x_mid, y2=y2, hpre=hpre, hid=hid)
        blocks.append(s)
    sv["blocks"], sv["xw_out"] = blocks, xw
    comb = ops.rt_patch_unembed(xw, pk["pu.w"], pk["pu.b"], feat_down)
    dec = ops.conv_c64(comb, pk["dec1.w"], pk["dec1.b"], 1, relu=True)
    residual = ops.conv_c64_thin(dec, pk["dec2.w"], pk["dec2.b"], 3, relu=False)
    out = ops.rt_bicubic_sum(x, residual, tuple(int(v) for v in res_out), clamp=True)
    sv["comb"], sv["dec"], sv["out"] = comb, dec, out          # the clamp gate is read back from the output itself
    return out, sv


def backward_train(pk, sv, gout, reducer=None, l1_scale=None) -> Dict[str, torch.Tensor]:
    g: Dict[str, torch.Tensor] = {}

    def ready(*names):
        if reducer is not None:
            reducer.on_ready(list(names), g)

    x = sv["x"]
    B, _, H, W = x.shape
    hd, wd = H // 2, W // 2
    N = (hd // 8) * (wd // 8)
    # ---- clamp + bicubic (only the residual branch carries parameters); l1_scale: gout is the target of an L1 loss on the
    #      output and the loss gradient is formed inside the kernel (autograd.l1_loss(..., fuse_into_model_backward=True)) ----
    gout = gout.contiguous().float()
    g_res = ops.rt_bicubic_bwd(gout, sv["out"], (hd, wd), l1_scale=None if l1_scale is None else l1_scale[:1])
    # ---- decoder_conv2 (64->3), decoder_conv1's ReLU, decoder_conv1 ----
    dwp, db = ops.conv_thin_wgrad(sv["dec"], g_res, True)
    g["decoder_conv2.weight"], g["decoder_conv2.bias"] = dwp.permute(0, 2, 1).reshape(3, 64, 3, 3), db
    g_dec = ops.conv1(g_res, pk["dec2.wd"], None, relu=False, out_mask=sv["dec"])
    ready("decoder_conv2.weight", "decoder_conv2.bias")
    dwp, db = ops.conv_c64_wgrad(sv["comb"], g_dec, 1)
    g["decoder_conv1.weight"], g["decoder_conv1.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_comb = ops.conv_c64(g_dec, pk["dec1.wd"], None, 1)
    del g_dec
    ready("decoder_conv1.weight", "decoder_conv1.bias")
    # ---- patch_unembed (+ skip) ----
    g["patch_unembed.bias"] = ops.colsum(g_comb.view(-1, 64))
    g["patch_unembed.weight"] = ops.rt_patch_wgrad(sv["xw_out"], g_comb).view(128, 8, 8, 64).permute(0, 3, 1, 2)
    g_x = ops.rt_patch_unembed_bwd(g_comb, pk["pu.wd"])
    ready("patch_unembed.weight", "patch_unembed.bias")
    # ---- transformer blocks (reverse) ----
    drop_p, seed = sv["drop_p"], sv["seed"]
    g_xd = None
    for i in reversed(range(pk["nblocks"])):
        s, p = sv["blocks"][i], f"transformer_blocks.{i}"
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
        g_xm, g[p + ".norm2.weight"], g[p + ".norm2.bias"] = ops.layernorm128_bwd(
            g_y2, s["x_mid"], s["mean2"], s["rstd2"], pk[f"b{i}.norm2.w"], gres=g_x)
        g[p + ".attn.out_proj.weight"], g[p + ".attn.out_proj.bias"] = ops.gemm_wgrad_bias(g_xm, s["att"])
        g_att = ops.gemm_tokens(g_xm, pk[f"b{i}.out.wd"], None, "bf16")
        with _stage("rt_attn_bwd"):
            g_qkv = ops.rt_attention_bwd(s["qkv"], s["att"], g_att, s["lse"], B, N, drop_p=drop_p, drop_seed=site_seed(seed, i, 0))
        g[p + ".attn.in_proj_weight"], g[p + ".attn.in_proj_bias"] = ops.gemm_wgrad_bias(g_qkv, s["y1"])
        g_y1 = ops.gemm_tokens(g_qkv, pk[f"b{i}.in.wd"], None, "bf16")
        del g_qkv, g_att
        if drop_p > 0 and i > 0:          # + the MLP dropout's backward for the block below
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"], g_xd = ops.layernorm128_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm, drop=(drop_p, site_seed(seed, i - 1, 2)))
        else:
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"] = ops.layernorm128_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm)
        ready(*[p + sfx for sfx in (".mlp.2.bias", ".mlp.2.weight", ".mlp.0.bias", ".mlp.0.weight", ".norm2.weight",
                                    ".norm2.bias", ".attn.out_proj.bias", ".attn.out_proj.weight", ".attn.in_proj_bias",
                                    ".attn.in_proj_weight", ".norm1.weight", ".norm1.bias")])
    # ---- pos_embed, patch_embed ----
    g["pos_embed"] = g_x.view(B, N, 128).sum(0, keepdim=True)
    g["patch_embed.bias"] = ops.colsum(g_x)
    g["patch_embed.weight"] = ops.rt_patch_wgrad(g_x, sv["feat_down"]).view(128, 8, 8, 64).permute(0, 3, 1, 2)
    g_fd = ops.rt_patch_embed_bwd(g_x, pk["pe.wd"], add=g_comb)            # + skip gradient
    del g_x, g_comb
    ready("pos_embed", "patch_embed.weight", "patch_embed.bias")
    # ---- downsample (stride-2 conv), conv2, conv1 ----
    dwp, db = ops.conv_c64_wgrad_s2d(sv["feat"], g_fd, 2)
    g["downsample.weight"], g["downsample.bias"] = packing.unpack_conv_c64_stride2_wgrad(dwp), db
    g_feat = ops.conv_c64(g_fd, pk["ds.wd"], None, 2, mask=sv["feat"])      # 4 sub-pixel tiles -> HR grid, conv2's ReLU
    del g_fd
    ready("downsample.weight", "downsample.bias")
    dwp, db = ops.conv_c64_wgrad(sv["feat1"], g_feat, 1)
    g["conv2.weight"], g["conv2.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_f1 = ops.conv_c64(g_feat, pk["conv2.wd"], None, 1, mask=sv["feat1"])
    g["conv1.weight"], g["conv1.bias"] = ops.conv1_wgrad(x, g_f1)
    ready("conv2.weight", "conv2.bias", "conv1.weight", "conv1.bias")
    return g


class _ResidualTransformerFn(torch.autograd.Function):
    accepts_fused_l1 = True          # autograd.l1_loss may hand (target, scale) to this node instead of a gradient tensor

    @staticmethod
    def forward(ctx, module, x, res_out, names, *params):
        pk = module.packed(backward=True)
        drop_p, seed = module._next_dropout()
        out, sv = forward_train(pk, x, res_out, drop_p, seed)
        ctx.module, ctx.names, ctx.sv, ctx.pk = module, names, sv, pk
        return out

    @staticmethod
    def backward(ctx, gout):
        # the fused-L1 hand-off is validated BEFORE the reducer opens its episode: a refusal here must not leave it open
        fused = getattr(ctx, "_fused_l1", None)
        l1_scale = None
        if fused is not None:
            target, l1_scale, stand_in = fused
            ctx._fused_l1 = None
            if gout.data_ptr() != stand_in.data_ptr() or any(st != 0 for st in gout.stride()):
                raise RuntimeError("l1_loss(..., fuse_into_model_backward=True): the model output has a consumer besides the loss "
                                   "(its gradient is not the loss's stand-in); call l1_loss without the fusion")
            gout = target
        reducer = getattr(ctx.module, "_grad_reducer", None)
        if reducer is not None:
            reducer.begin(ctx.names)          # raises if this step's parameters are not in the reducer's layout
        ops.zero_pool_begin(gout.device)
        try:
            grads = backward_train(ctx.pk, ctx.sv, gout, reducer, l1_scale=l1_scale)
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


def residual_transformer_function(module, x, res_out):
    named = dict(module.named_parameters())
    names = [n for n, p in named.items() if p.requires_grad]
    return _ResidualTransformerFn.apply(module, x, tuple(res_out), names, *[named[n] for n in names])
