"""Training path: forward that saves activations + the hand-written backward, as ONE autograd node.

Replaces what torch autograd does for reference models/FastTransformer/model.py:231-327 under
train.py:113-140: every gradient below is produced by a HIP kernel through the C ABI (ops.py); torch
only carries the node in its graph, owns the tensors and accumulates ``.grad``.

Dropout (model.py:80-82,127,132,150; p = 0.1) is applied in ``.train()`` mode with stateless hash masks that
the backward re-derives (nothing is stored); ``.eval()`` with grads enabled gives the dropout-free graph the
parity fixtures use.  The masks are not torch's Philox stream -- same distribution, different bits.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import os

import torch

from . import ops, packing
from .weights import BLOCKS, active_param_names, upsampler_layout

BF16, F32 = torch.bfloat16, torch.float32

# Data-parallel training (dp.py): a reducer attached to the module is told, in reverse execution order,
# as soon as a group of parameter gradients is final, so its all-reduce overlaps the rest of the backward.

_ROWMASK_CACHE = {}


def _valid_token_rowmask(B, H, W, device):
    key = (B, H, W, str(device))
    if key not in _ROWMASK_CACHE:
        ht, wt, nwy, nwx = ops.window_geometry(H, W)
        ty = (torch.arange(nwy).view(-1, 1, 1, 1) * 8 + torch.arange(8).view(1, 1, -1, 1))
        tx = (torch.arange(nwx).view(1, -1, 1, 1) * 8 + torch.arange(8).view(1, 1, 1, -1))
        m = ((ty < ht) & (tx < wt)).expand(nwy, nwx, 8, 8).reshape(1, -1).expand(B, -1).reshape(-1)
        _ROWMASK_CACHE[key] = m.to(torch.uint8).contiguous().to(device)
    return _ROWMASK_CACHE[key]


def site_seed(seed: int, block: int, site: int) -> int:
    """Per-dropout-site seed (site 0 = attn_drop, 1 = proj_drop, 2 = MLP dropout of block `block`)."""
    return (seed * 0x9E3779B9 + (3 * block + site + 1) * 0x85EBCA6B) & 0xFFFFFFFF


pe_merge = True          # A/B attribute (tests flip it): the gradient merge at `feat` inside patch_embed's input gradient


def forward_train(pk, frags_t, x, scale, res_out, require_ratio, drop_p=0.0, seed=0):
    """Same stages as engine.forward, keeping what the backward needs in `sv`.  drop_p > 0 applies the three
    nn.Dropout sites of every block (model.py:80-82,127,132,150) with stateless masks keyed by `seed`."""
    sv = {"drop_p": float(drop_p), "seed": int(seed)}
    x = x.contiguous().float()
    B, _, H, W = x.shape
    sv["x"] = x
    sv["feat1"] = ops.conv1(x, pk["conv1.w"], pk["conv1.b"], relu=True)
    feat = sv["feat"] = ops.conv_c64(sv["feat1"], pk["conv2.w"], pk["conv2.b"], 1, relu=True)
    stages = upsampler_layout(scale)
    ups = [feat]
    # the last Upsampler stage + up1_conv through their exact composition (csrc/branch_a_train.hip) when it is a x2 stage:
    # one 5x5 conv with 12 outputs instead of the 64 -> 256 conv, the 64-channel HR tensor and the 64 -> 3 conv
    composed = "bra.comp" in pk and stages[-1][1] == 2 and min(H, W) * (scale // 2) >= 6
    for si, (_, r) in enumerate(stages[:-1] if composed else stages):
        ups.append(ops.conv_c64(ups[-1], pk[f"up1.{si}.w"], pk[f"up1.{si}.b"], r, relu=False))
    sv["ups"], sv["bra"] = ups, composed
    if composed:
        c = pk["bra.comp"]
        ui = sv["ui"] = ops.branch_a_composed(ups[-1], c["wp"], c["bias"], c["wv"], c["bv"], 2, relu=True)
    else:
        ui = sv["ui"] = ops.conv_c64_thin(ups[-1], pk["up1_conv.w"], None, 3, relu=True)
    xw = ops.patch_embed(feat, pk["pe.w"], pk["pe.b"])
    blocks = []
    for i in range(BLOCKS):
        s = {"x_in": xw}
        s["y1"], s["mean1"], s["rstd1"] = ops.layernorm(xw, pk[f"b{i}.norm1.w"], pk[f"b{i}.norm1.b"], save_stats=True)
        s["qkv"] = ops.gemm_tokens(s["y1"], pk[f"b{i}.qkv.w"], pk[f"b{i}.qkv.b"], "bf16")
        s["att"], s["lse"] = ops.window_attn(s["qkv"], frags_t[i], drop_p, site_seed(seed, i, 0), save_lse=True)
        xm = s["x_mid"] = ops.gemm_tokens(s["att"], pk[f"b{i}.proj.w"], pk[f"b{i}.proj.b"], "res", res=xw,
                                          drop_p=drop_p, drop_seed=site_seed(seed, i, 1))
        s["y2"], s["mean2"], s["rstd2"] = ops.layernorm(xm, pk[f"b{i}.norm2.w"], pk[f"b{i}.norm2.b"], save_stats=True)
        s["hpre"] = torch.empty((xm.shape[0], 768), dtype=BF16, device=x.device)
        s["hid"] = ops.gemm_tokens(s["y2"], pk[f"b{i}.fc1.w"], pk[f"b{i}.fc1.b"], "gelu", aux=s["hpre"])
        xw = ops.gemm_tokens(s["hid"], pk[f"b{i}.fc2.w"], pk[f"b{i}.fc2.b"], "res", res=xm,
                             drop_p=drop_p, drop_seed=site_seed(seed, i, 2))
        blocks.append(s)
    sv["blocks"] = blocks
    sv["xw_out"] = xw
    comb = sv["comb"] = ops.patch_unembed(xw, pk["pu.w"], pk["pu.b"], feat)
    dec = sv["dec"] = ops.conv_c64(comb, pk["dec1.w"], pk["dec1.b"], 1, relu=True)
    res = ops.conv_c64_thin(dec, pk["dec2.w"], pk["dec2.b"], 3, relu=False)
    ts = [res]
    for si, (_, r) in enumerate(upsampler_layout(scale)):
        ts.append(ops.conv_planar(ts[-1], pk[f"fu.{si}.w"], pk[f"fu.{si}.b"], r))
    sv["ts"] = ts
    hs, ws = H * scale, W * scale
    needs_resize = bool(require_ratio) and tuple(res_out) != (hs, hs) and tuple(res_out) != (hs, ws)
    if needs_resize:
        total = ops.conv_planar(ts[-1], pk["fuc.w"], pk["fuc.b"], 1, add=ui, clamp=False)     # pre-clamp sum
        pre, out = ops.resize_aa(total, res_out, clamp="both")      # the clamp's backward gate and the model output in one pass
        sv["resized_from"] = (hs, ws)
    else:
        pre, out = ops.conv_planar(ts[-1], pk["fuc.w"], pk["fuc.b"], 1, add=ui, clamp="both")   # likewise, straight from the conv
        sv["resized_from"] = None
    sv["pre"] = pre
    return out, sv


def backward_train(pk, frags_t, frags_n, sv, scale, gout, reducer=None, want_input_grad=False, l1_scale=None) -> Dict[str, torch.Tensor]:
    """Returns {reference parameter name: gradient} for the parameters active at `scale` (+ "__input__" = d loss / d x when
    `want_input_grad`: conv1's input gradient as a 64 -> 3 conv of the flipped weights)."""
    g: Dict[str, torch.Tensor] = {}

    def ready(*names):
        if reducer is not None:
            reducer.on_ready(list(names), g)

    x, feat, ui = sv["x"], sv["feat"], sv["ui"]
    B, _, H, W = x.shape
    stages = upsampler_layout(scale)
    gout = gout.contiguous().float()          # (l1_scale given: the L1 target -- the loss gradient is formed inside the first kernel)
    # ---- clamp (+ resize) ----
    if sv["resized_from"] is not None:
        g_sum = ops.resize_aa_bwd(gout, sv["resized_from"], pre=sv["pre"], l1_scale=l1_scale)
    else:
        g_sum = ops.mask_bwd(gout, pre=sv["pre"], l1_scale=l1_scale)
    # ---- final_upscale_conv (3->3) + "+ upscaled_input" ----
    ts = sv["ts"]
    g["final_upscale_conv.weight"], g["final_upscale_conv.bias"] = ops.conv_planar_wgrad(ts[-1], g_sum, 1)
    g_t = ops.conv_planar(g_sum, pk["fuc.wd"], None, 1)
    ready("final_upscale_conv.weight", "final_upscale_conv.bias")
    # ---- final_upscale stages (reverse) ----
    for si in reversed(range(len(stages))):
        idx, r = stages[si]
        k = f"final_upscale.upsamplers.{scale}.{idx}"
        g[k + ".weight"], g[k + ".bias"] = ops.conv_planar_wgrad(ts[si], g_t, r)
        g_t = ops.conv_planar_dgrad(g_t, pk[f"fu.{si}.raw"], r)
        ready(k + ".weight", k + ".bias")
    g_res = g_t                                                   # d residual, planar [B][3][H][W]
    # ---- decoder_conv2 (64->3) and decoder_conv1's ReLU ----
    dwp, db = ops.conv_thin_wgrad(sv["dec"], g_res, True)
    g["decoder_conv2.weight"], g["decoder_conv2.bias"] = dwp.permute(0, 2, 1).reshape(3, 64, 3, 3), db
    g_dec = ops.conv1(g_res, pk["dec2.wd"], None, relu=False, out_mask=sv["dec"])
    ready("decoder_conv2.weight", "decoder_conv2.bias")
    # ---- decoder_conv1 (64->64) ----
    dwp, db = ops.conv_c64_wgrad(sv["comb"], g_dec, 1)
    g["decoder_conv1.weight"], g["decoder_conv1.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_comb = ops.conv_c64(g_dec, pk["dec1.wd"], None, 1)
    del g_dec
    ready("decoder_conv1.weight", "decoder_conv1.bias")
    # ---- patch_unembed (+ skip) ----
    # patch_unembed's bias gradient = the column sums of g_comb.  Without a gradient reducer they ride along in the kernel that reads
    # g_comb last (the merged patch_embed input gradient at the end of this function); with one, the bias has to be ready NOW so that
    # its bucket's all-reduce starts under the rest of the backward
    late_pu_bias = reducer is None and pe_merge and H % 8 == 0 and W % 8 == 0
    if not late_pu_bias:
        g["patch_unembed.bias"] = ops.colsum(g_comb.view(-1, 64))
    g["patch_unembed.weight"] = ops.patch_wgrad(sv["xw_out"], g_comb, reflect=False).view(192, 8, 8, 64).permute(0, 3, 1, 2)
    g_x = ops.patch_unembed_bwd(g_comb, pk["pu.wd"])
    ready("patch_unembed.weight", "patch_unembed.bias")
    # ---- transformer blocks (reverse) ----
    drop_p, seed = sv["drop_p"], sv["seed"]
    g_xd = None
    for i in reversed(range(BLOCKS)):
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
            g_xm, g[p + ".norm2.weight"], g[p + ".norm2.bias"], g_o = ops.layernorm_bwd(
                g_y2, s["x_mid"], s["mean2"], s["rstd2"], pk[f"b{i}.norm2.w"], gres=g_x, drop=(drop_p, site_seed(seed, i, 1)))
        else:
            g_xm, g[p + ".norm2.weight"], g[p + ".norm2.bias"] = ops.layernorm_bwd(
                g_y2, s["x_mid"], s["mean2"], s["rstd2"], pk[f"b{i}.norm2.w"], gres=g_x)
            g_o = g_xm      # proj_drop
        g[p + ".attn.proj.weight"], g[p + ".attn.proj.bias"] = ops.gemm_wgrad_bias(g_o, s["att"])
        g_att = ops.gemm_tokens(g_o, pk[f"b{i}.proj.wd"], None, "bf16")
        del g_o
        g_qkv, g[p + ".attn.relative_position_bias_table"] = ops.window_attn_bwd(
            s["qkv"], g_att, s["att"], s["lse"], frags_n[i], drop_p, site_seed(seed, i, 0))
        g[p + ".attn.qkv.weight"], g[p + ".attn.qkv.bias"] = ops.gemm_wgrad_bias(g_qkv, s["y1"])
        g_y1 = ops.gemm_tokens(g_qkv, pk[f"b{i}.qkv.wd"], None, "bf16")
        del g_qkv, g_att
        if drop_p > 0 and i > 0:          # + the MLP dropout's backward for the block below
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"], g_xd = ops.layernorm_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm, drop=(drop_p, site_seed(seed, i - 1, 2)))
        else:
            g_x, g[p + ".norm1.weight"], g[p + ".norm1.bias"] = ops.layernorm_bwd(
                g_y1, s["x_in"], s["mean1"], s["rstd1"], pk[f"b{i}.norm1.w"], gres=g_xm)
        ready(*[p + sfx for sfx in (".mlp.2.bias", ".mlp.2.weight", ".mlp.0.bias", ".mlp.0.weight", ".norm2.weight",
                                    ".norm2.bias", ".attn.proj.bias", ".attn.proj.weight",
                                    ".attn.relative_position_bias_table", ".attn.qkv.bias", ".attn.qkv.weight",
                                    ".norm1.weight", ".norm1.bias")])
    # ---- patch_embed ----
    g["patch_embed.bias"] = ops.colsum(g_x, rowmask=_valid_token_rowmask(B, H, W, x.device))
    g["patch_embed.weight"] = ops.patch_wgrad(g_x, feat, reflect=True).view(192, 8, 8, 64).permute(0, 3, 1, 2)
    # the input gradient of patch_embed: when the map needs no reflect padding (H, W multiples of 8) it is computed LAST, with the
    # merge of the three gradient paths into `feat` and conv2's ReLU gate in its epilogue (no padded map, no feat_grad_combine pass)
    merge_in_pe = pe_merge and H % 8 == 0 and W % 8 == 0
    g_pe = None if merge_in_pe else ops.patch_embed_bwd(g_x, pk["pe.wd"], B, H, W)
    if not merge_in_pe:
        del g_x
    ready("patch_embed.weight", "patch_embed.bias")
    # ---- branch A: up1_conv (ReLU, no bias) and the Upsampler stages ----
    ups = sv["ups"]
    nexp = len(stages)                 # stages whose backward runs through the explicit conv kernels
    if sv["bra"]:
        idx, _ = stages[-1]
        k = f"up1.upsamplers.{scale}.{idx}"
        g_up, g[k + ".weight"], g[k + ".bias"], g["up1_conv.conv.weight"] = ops.bra_backward(
            g_sum, ui, ups[-1], pk["bra.comp"], pk["bra.wu"], pk["bra.bu"], pk["bra.w3"])[:4]
        ready("up1_conv.conv.weight", k + ".weight", k + ".bias")
        nexp -= 1
    else:
        g_ui = ops.mask_bwd(g_sum, relu_src=ui)
        dwp, _ = ops.conv_thin_wgrad(ups[-1], g_ui, False)
        g["up1_conv.conv.weight"] = dwp.permute(0, 2, 1).reshape(3, 64, 3, 3)
        g_up = ops.conv1(g_ui, pk["up1_conv.wd"], None, relu=False)
        ready("up1_conv.conv.weight")
    for si in reversed(range(nexp)):
        idx, r = stages[si]
        k = f"up1.upsamplers.{scale}.{idx}"
        dwp, db = ops.conv_c64_wgrad(ups[si], g_up, r)
        g[k + ".weight"], g[k + ".bias"] = packing.unpack_conv_c64_wgrad(dwp, db, r)
        g_up = ops.conv_c64(g_up, pk[f"up1.{si}.wd"], None, 1, in_r=r)
        ready(k + ".weight", k + ".bias")
    # ---- merge at feat + conv2's ReLU, conv2, conv1 ----
    if merge_in_pe and late_pu_bias:
        g_feat, g["patch_unembed.bias"] = ops.patch_embed_bwd_merge(g_x, pk["pe.wd"], g_comb, g_up, feat, want_add1_colsum=True)
        del g_x
    elif merge_in_pe:
        g_feat = ops.patch_embed_bwd_merge(g_x, pk["pe.wd"], g_comb, g_up, feat)
        del g_x
    else:
        g_feat = ops.feat_grad_combine(g_comb, g_up, g_pe, feat)
    del g_comb, g_up, g_pe
    dwp, db = ops.conv_c64_wgrad(sv["feat1"], g_feat, 1)
    g["conv2.weight"], g["conv2.bias"] = packing.unpack_conv_c64_wgrad(dwp, db, 1)
    g_f1 = ops.conv_c64(g_feat, pk["conv2.wd"], None, 1, mask=sv["feat1"])
    g["conv1.weight"], g["conv1.bias"] = ops.conv1_wgrad(x, g_f1)
    ready("conv2.weight", "conv2.bias", "conv1.weight", "conv1.bias")
    if want_input_grad:
        g["__input__"] = ops.conv_c64_thin(g_f1, pk["conv1.wd"], None, 3, relu=False)
    return g


class _FastTransformerFn(torch.autograd.Function):
    accepts_fused_l1 = True          # l1_loss may hand (target, scale) to this node instead of a gradient tensor (see _L1LossFn)

    @staticmethod
    def forward(ctx, module, x, scale, res_out, require_ratio, names, *params):
        pk, frags_t, frags_n = module.packed(scale, backward=True)
        drop_p, seed = module._next_dropout()
        out, sv = forward_train(pk, frags_t, x, scale, res_out, require_ratio, drop_p, seed)
        ctx.module, ctx.scale, ctx.names, ctx.sv = module, scale, names, sv
        ctx.pk, ctx.frags = pk, (frags_t, frags_n)
        ctx.x_dtype = x.dtype
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
            grads = backward_train(ctx.pk, ctx.frags[0], ctx.frags[1], ctx.sv, ctx.scale, gout, reducer,
                                   want_input_grad=bool(ctx.needs_input_grad[1]), l1_scale=l1_scale)
        except BaseException:
            if reducer is not None:
                reducer._abort()
            raise
        finally:
            ops.zero_pool_end()
        gx = grads.pop("__input__", None)       # the reference supplies d/dx too (an input that requires grad)
        if reducer is not None:
            grads = reducer.finish()          # averaged over ranks (views of the flat bucket buffer)
        ctx.sv = None
        outs = []
        for n in ctx.names:
            gr = grads.get(n)
            outs.append(None if gr is None else gr.contiguous())      # reducer: views of this episode's own flat buffer (dp.py)
        return (None, None if gx is None else gx.to(ctx.x_dtype), None, None, None, None) + tuple(outs)


def fast_transformer_function(module, x, scale, res_out, require_ratio):
    """Only the parameters active at `scale` enter the node, so the other scales' upsamplers keep
    ``grad is None`` and Adam skips them exactly as in the reference."""
    named = dict(module.named_parameters())
    names = [n for n in active_param_names(scale) if named[n].requires_grad]
    return _FastTransformerFn.apply(module, x, scale, tuple(res_out), bool(require_ratio), names, *[named[n] for n in names])


class _ResizeAAFn(torch.autograd.Function):
    """transforms.Resize on a tensor (train.py:127-130) with a HIP forward and backward."""
    accepts_fused_l1 = True          # the loss of train.py:132 sits right behind this Resize: its gradient is formed inside the backward kernel

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = tuple(x.shape[2:])
        out = ops.resize_aa(x.contiguous().float(), tuple(size), clamp=False)
        ctx.save_for_backward(out)          # only read when the L1 loss hands its target over (sign(out - target))
        return out

    @staticmethod
    def backward(ctx, gout):
        fused = getattr(ctx, "_fused_l1", None)
        if fused is not None:
            out, = ctx.saved_tensors
            target, scale, stand_in = fused
            ctx._fused_l1 = None
            if gout.data_ptr() != stand_in.data_ptr() or any(st != 0 for st in gout.stride()):
                raise RuntimeError("l1_loss(..., fuse_into_model_backward=True): the resized output has a consumer besides the loss "
                                   "(its gradient is not the loss's stand-in); call l1_loss without the fusion")
            scale[1] = 1.0          # plain: `pre` below is the loss input itself, no clamp between this Resize and the loss
            return ops.resize_aa_bwd(target, ctx.in_hw, pre=out, l1_scale=scale), None
        return ops.resize_aa_bwd(gout.contiguous().float(), ctx.in_hw), None


def resize_aa(x: torch.Tensor, size) -> torch.Tensor:
    if tuple(x.shape[2:]) == tuple(size):
        return x
    return _ResizeAAFn.apply(x, tuple(size))


class _L1LossFn(torch.autograd.Function):
    """nn.L1Loss() (train.py:103,132) with a HIP forward and backward: two streaming passes instead of aten's six."""

    @staticmethod
    def forward(ctx, out, target, model_node):
        out = out.contiguous().float()
        target = target.contiguous().float()
        ctx.save_for_backward(out, target)
        ctx.model_node = model_node
        part = ops.l1_loss_partial(out, target)
        return (part.double().sum() / out.numel()).float()

    @staticmethod
    def backward(ctx, g):
        out, target = ctx.saved_tensors
        if ctx.model_node is not None:
            # hand (target, d loss / numel) to the model's backward node, which forms sign(out - target) * scale inside its first
            # kernel; what is returned here is a stride-0 stand-in of the right shape that node recognises and never reads
            # scale = {d loss / numel, 0}: the second float is the receiving node's "plain" flag (csrc/conv_bwd.hip l1_grad_plain)
            scale = torch.zeros(2, dtype=torch.float32, device=out.device)
            scale[:1] = g.detach().float().reshape(1) / out.numel()
            stand_in = scale[:1].expand(out.shape)
            ctx.model_node._fused_l1 = (target, scale, stand_in)
            return stand_in, None, None
        return ops.l1_loss_bwd(out, target, g.contiguous().float().reshape(1)), None, None


def l1_loss(out: torch.Tensor, target: torch.Tensor, fuse_into_model_backward: bool = False) -> torch.Tensor:
    """nn.L1Loss()(out, target).  fuse_into_model_backward=True (the training harness, where the model output feeds the loss and
    nothing else, train.py:124-136): when `out` comes straight from a model whose backward supports it (FastTransformer, ResidualTransformer), the
    loss gradient is not materialised -- the model's first backward kernel computes sign(out - target) / numel itself.  The model's
    backward raises if it then receives anything but that stand-in (an output with a second consumer).  Under the fusion the
    gradient that flows into `out` is that stand-in, not sign(out - target) / numel: ``out.grad`` / tensor hooks on `out` would
    see it, so an `out` that retains its gradient or carries hooks is given the materialised gradient instead (no fusion)."""
    if out.shape != target.shape:
        raise ValueError(f"l1_loss: output {tuple(out.shape)} and target {tuple(target.shape)} differ (train.py:132 compares equal shapes)")
    node = None
    if fuse_into_model_backward and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() \
            and not out.retains_grad and not getattr(out, "_backward_hooks", None) \
            and getattr(getattr(out.grad_fn, "_forward_cls", None), "accepts_fused_l1", False):
        node = out.grad_fn
    return _L1LossFn.apply(out, target, node)
