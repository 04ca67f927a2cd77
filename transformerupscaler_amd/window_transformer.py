"""``TransformerModel`` of the WindowTransformer plugin (reference models/WindowTransformer/model.py:172-305; SURVEY
§8(f) rank 2) on the MI355X HIP kernels.

Architecture = ResidualTransformer's shell (bicubic global residual, conv1/conv2, stride-2 downsample, patch_embed
k8 s8 -> 128, patch_unembed, decoder convs) around FastTransformer's window blocks (8x8 windows, relative position
bias) at width 128 / 8 heads, 8 blocks.  With gradients enabled the call goes through autograd_wt (hand-written
backward; dropout p = 0.01 in ``.train()`` as stateless hash masks).  Same constructor keywords, forward signature and state_dict keys as the
reference (incl. the int64 ``relative_position_index`` buffers).  Arbitrary input sizes: the stride-8 patch conv drops
the remainder rows / columns and the decoder runs on the cropped (H_t*8 x W_t*8) map, as the reference does.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch
import torch.nn as nn

from . import ops, packing
from .fast_transformer import WindowTransformerBlock, _ConvParams

use_pack_plan = True          # A/B attribute: training re-pack as two gather launches (pack_plan.py)


def pad_to_even(feat: torch.Tensor) -> torch.Tensor:
    """NHWC map with an odd height / width -> one zero row / column appended.  The stride-2, pad-1 conv of the reference
    (models/WindowTransformer/model.py:205) reads input row H only through its zero padding, so on the padded map the even-size
    kernel computes exactly the reference's ceil(H / 2) x ceil(W / 2) outputs."""
    B, H, W, C = feat.shape
    if H % 2 == 0 and W % 2 == 0:
        return feat
    return torch.nn.functional.pad(feat, (0, 0, 0, W % 2, 0, H % 2))


class TransformerModel(nn.Module):
    def __init__(self, in_channels: int = 3, base_channels: int = 64, transformer_dim: int = 128, num_window_blocks: int = 8,
                 num_heads: int = 8, mlp_ratio: float = 4.0, dropout: float = 0.01, window_size: int = 8):
        super().__init__()
        if (in_channels, base_channels, transformer_dim, num_heads, window_size) != (3, 64, 128, 8, 8) or float(mlp_ratio) != 4.0:
            raise NotImplementedError("the HIP kernels are specialised for the reference's defaults (3, 64, 128, 8 heads, window 8, mlp 4.0)")
        self.dropout_p = float(dropout)
        self.num_heads = num_heads
        self.conv1 = _ConvParams(in_channels, base_channels, 3)
        self.conv2 = _ConvParams(base_channels, base_channels, 3)
        self.downsample = _ConvParams(base_channels, base_channels, 3)
        self.patch_embed = _ConvParams(base_channels, transformer_dim, 8)
        self.window_size = window_size
        self.window_blocks = nn.ModuleList([WindowTransformerBlock(transformer_dim, window_size, num_heads, mlp_ratio)
                                            for _ in range(num_window_blocks)])
        self.patch_unembed = _ConvParams(transformer_dim, base_channels, 8, transposed=True)
        self.decoder_conv1 = _ConvParams(base_channels, base_channels, 3)
        self.decoder_conv2 = _ConvParams(base_channels, in_channels, 3)
        self._pack_cache = {}
        self._dropout_calls = 0

    def _next_dropout(self):
        """(p, seed) of the next training forward (see fast_transformer.TransformerModel._next_dropout)."""
        if not self.training or self.dropout_p <= 0.0:
            return 0.0, 0
        import os
        self._dropout_calls += 1
        base = (torch.initial_seed() + 7919 * int(os.environ.get("RANK", "0"))) & 0x7FFFFFFF
        return self.dropout_p, (base * 2654435761 + self._dropout_calls) & 0xFFFFFFFF

    def invalidate_packed(self) -> None:
        """Drop the packed-weight cache (after ``p.data`` writes, which do not bump the version counter the cache is keyed on)."""
        self._pack_cache = {}

    def packed(self, backward: bool = False):
        """(packed weights, T-layout bias fragments[, N-layout fragments for the backward])."""
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters())
        hit = self._pack_cache.get(bool(backward))
        if hit is None or hit[0] != ver:
            sd, pk = dict(self.named_parameters()), None
            if backward and use_pack_plan:       # training: re-pack = two gather launches (pack_plan.py)
                from .pack_plan import packed_with_plan
                pk = packed_with_plan(self, "wt", sd, lambda d: packing.pack_wt_state_dict(d, backward=True))
            if pk is None:
                pk = packing.pack_wt_state_dict(sd, backward=backward)
            frags = [ops.relpos_bias_expand_h(pk[f"b{i}.table"], self.num_heads) for i in range(pk["nblocks"])]
            entry = (ver, pk, frags)
            if backward:
                entry += ([ops.relpos_bias_expand_n_h(pk[f"b{i}.table"], self.num_heads) for i in range(pk["nblocks"])],)
            self._pack_cache[bool(backward)] = hit = entry
        return hit[1:]

    def forward(self, x: torch.Tensor, res_out: Tuple[int, int] = (1080, 1920), upscale_factor: Optional[int] = None,
                require_ratio: bool = True) -> torch.Tensor:
        if upscale_factor is not None:
            res_out = (x.shape[2] * upscale_factor, x.shape[3] * upscale_factor)          # model.py:236-237
        if not x.is_cuda:
            raise RuntimeError("TransformerModel (MI355X build) runs on the GPU only; there is no CPU fallback.")
        B, _, H, W = x.shape
        hd, wd = (H + 1) // 2, (W + 1) // 2             # the stride-2, pad-1 conv: ceil(H / 2) x ceil(W / 2), model.py:205,244
        if hd < 8 or wd < 8:
            raise RuntimeError("input too small for one 8x8 patch after the stride-2 conv")    # the reference's conv fails too
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd_wt import window_transformer_function
            out = window_transformer_function(self, x, tuple(int(v) for v in res_out))
            return out.to(torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else out
        if self.training and self.dropout_p > 0.0:       # dropout is active in .train() with or without gradients, as in the reference
            from .autograd_wt import forward_train
            pk, frags_t, _ = self.packed(backward=True)
            drop_p, seed = self._next_dropout()
            out, _ = forward_train(pk, frags_t, self.num_heads, x, tuple(int(v) for v in res_out), drop_p, seed)
            return out.to(torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else out
        pk, frags = self.packed()
        x = x.contiguous().float()
        feat = ops.conv_c64(ops.conv1(x, pk["conv1.w"], pk["conv1.b"], relu=True), pk["conv2.w"], pk["conv2.b"], 1, relu=True)
        feat_down = ops.conv_c64(pad_to_even(feat), pk["ds.w"], pk["ds.b"], 1, relu=False, in_r=2)        # model.py:244
        del feat
        xw = ops.wt_patch_embed(feat_down, pk["pe.w"], pk["pe.b"])                          # model.py:247-268
        heads = self.num_heads
        for i in range(pk["nblocks"]):
            y = ops.layernorm128(xw, pk[f"b{i}.norm1.w"], pk[f"b{i}.norm1.b"])
            qkv = ops.gemm_tokens(y, pk[f"b{i}.qkv.w"], pk[f"b{i}.qkv.b"], "bf16")
            att = ops.window_attn_h(qkv, frags[i], heads)
            ops.gemm_tokens(att, pk[f"b{i}.proj.w"], pk[f"b{i}.proj.b"], "res", res=xw, out=xw)
            y = ops.layernorm128(xw, pk[f"b{i}.norm2.w"], pk[f"b{i}.norm2.b"])
            hid = ops.gemm_tokens(y, pk[f"b{i}.fc1.w"], pk[f"b{i}.fc1.b"], "gelu")
            ops.gemm_tokens(hid, pk[f"b{i}.fc2.w"], pk[f"b{i}.fc2.b"], "res", res=xw, out=xw)
        hs, ws = (hd // 8) * 8, (wd // 8) * 8                                               # crop for the skip, model.py:284-288
        skip = feat_down if (hs, ws) == (hd, wd) else feat_down[:, :hs, :ws, :].contiguous()
        comb = ops.wt_patch_unembed(xw, pk["pu.w"], pk["pu.b"], skip)
        dec = ops.conv_c64(comb, pk["dec1.w"], pk["dec1.b"], 1, relu=True)
        residual = ops.conv_c64_thin(dec, pk["dec2.w"], pk["dec2.b"], 3, relu=False)
        out = ops.rt_bicubic_sum(x, residual, tuple(int(v) for v in res_out), clamp=True)   # model.py:240,294-298
        if torch.is_autocast_enabled():
            out = out.to(torch.get_autocast_gpu_dtype())
        return out
