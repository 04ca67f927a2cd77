"""``TransformerModel`` of the ResidualTransformer plugin (reference models/ResidualTransformer/model.py:53-165,
BASELINE.json config 5) on the MI355X HIP kernels -- forward / inference path.

Same constructor keywords, forward signature and state_dict keys as the reference (incl. nn.MultiheadAttention's
packed ``attn.in_proj_weight`` / ``attn.out_proj``).  Like the reference it only accepts inputs whose token grid has
3600 tokens (720x1280): ``tokens + pos_embed`` (model.py:140) fixes the sequence length.  With gradients enabled
the call goes through autograd_rt (hand-written backward; dropout in ``.train()`` mode as stateless hash masks on the
attention probabilities and the MLP output, reference model.py:30,37).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops, packing
from .fast_transformer import _ConvParams, _LayerNormParams, _LinearParams

use_pack_plan = True          # A/B attribute: training re-pack as two gather launches (pack_plan.py)


class _MHAParams(nn.Module):
    """Parameter layout of nn.MultiheadAttention(embed, heads): in_proj_weight/bias + out_proj.{weight,bias}."""

    def __init__(self, dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = _LinearParams(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class TransformerBlock(nn.Module):
    def __init__(self, dim: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = _LayerNormParams(dim)
        self.attn = _MHAParams(dim)
        self.norm2 = _LayerNormParams(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(_LinearParams(dim, hidden), nn.Identity(), _LinearParams(hidden, dim), nn.Identity())


class TransformerModel(nn.Module):
    def __init__(self, in_channels=3, base_channels=64, embed_dim=64, transformer_dim=128, num_transformer_blocks=8,
                 num_heads=8, mlp_ratio=4.0, dropout=0.1):
        super().__init__()
        if (in_channels, base_channels, transformer_dim, num_heads) != (3, 64, 128, 8) or float(mlp_ratio) != 4.0:
            raise NotImplementedError("the HIP kernels are specialised for the reference's defaults (3, 64, 128, 8 heads, mlp 4.0)")
        self.dropout_p = float(dropout)
        self.token_H, self.token_W = 360 // 8, 640 // 8
        self.num_tokens = self.token_H * self.token_W
        self.pos_embed = nn.Parameter(torch.randn(1, self.num_tokens, transformer_dim))
        self.conv1 = _ConvParams(in_channels, base_channels, 3)
        self.conv2 = _ConvParams(base_channels, base_channels, 3)
        self.downsample = _ConvParams(base_channels, base_channels, 3)
        self.patch_embed = _ConvParams(base_channels, transformer_dim, 8)
        self.transformer_blocks = nn.ModuleList([TransformerBlock(transformer_dim, mlp_ratio) for _ in range(num_transformer_blocks)])
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
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters())
        hit = self._pack_cache.get(bool(backward))
        if hit is None or hit[0] != ver:
            sd, pk = dict(self.named_parameters()), None
            if backward and use_pack_plan:       # training: re-pack = two gather launches (pack_plan.py)
                from .pack_plan import packed_with_plan
                pk = packed_with_plan(self, "rt", sd, lambda d: packing.pack_rt_state_dict(d, backward=True))
            hit = (ver, pk if pk is not None else packing.pack_rt_state_dict(sd, backward=backward))
            self._pack_cache[bool(backward)] = hit
        return hit[1]

    def forward(self, x: torch.Tensor, res_out: Tuple[int, int] = (1080, 1920), upscale_factor: Optional[int] = None,
                require_ratio: bool = True) -> torch.Tensor:
        if upscale_factor is not None:
            res_out = (x.shape[2] * upscale_factor, x.shape[3] * upscale_factor)          # model.py:121-122
        if not x.is_cuda:
            raise RuntimeError("TransformerModel (MI355X build) runs on the GPU only; there is no CPU fallback.")
        B, _, H, W = x.shape
        hd, wd = (H + 1) // 2, (W + 1) // 2                                               # stride-2, pad-1, k3 conv
        if H % 2 or W % 2 or hd % 8 or wd % 8 or (hd // 8) * (wd // 8) != self.num_tokens:
            raise RuntimeError(f"input {H}x{W} gives {(hd // 8) * (wd // 8)} tokens; pos_embed has {self.num_tokens} "
                               "(the reference fails the same way at model.py:140)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd_rt import residual_transformer_function
            out = residual_transformer_function(self, x, tuple(int(v) for v in res_out))
            return out.to(torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else out
        if self.training and self.dropout_p > 0.0:       # dropout is active in .train() with or without gradients, as in the reference
            from .autograd_rt import forward_train
            drop_p, seed = self._next_dropout()
            out, _ = forward_train(self.packed(backward=True), x, tuple(int(v) for v in res_out), drop_p, seed)
            return out.to(torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else out
        pk = self.packed()
        x = x.contiguous().float()
        feat = ops.conv_c64(ops.conv1(x, pk["conv1.w"], pk["conv1.b"], relu=True), pk["conv2.w"], pk["conv2.b"], 1, relu=True)
        feat_down = ops.conv_c64(feat, pk["ds.w"], pk["ds.b"], 1, relu=False, in_r=2)        # stride-2 conv, model.py:132
        del feat
        xw = ops.rt_patch_embed(feat_down, pk["pe.w"], pk["pe.b"], pk["pos"])
        N = self.num_tokens
        for i in range(pk["nblocks"]):
            y = ops.layernorm128(xw, pk[f"b{i}.norm1.w"], pk[f"b{i}.norm1.b"])
            qkv = ops.gemm_tokens(y, pk[f"b{i}.in.w"], pk[f"b{i}.in.b"], "bf16")
            att = ops.rt_attention(qkv, B, N)
            ops.gemm_tokens(att, pk[f"b{i}.out.w"], pk[f"b{i}.out.b"], "res", res=xw, out=xw)
            y = ops.layernorm128(xw, pk[f"b{i}.norm2.w"], pk[f"b{i}.norm2.b"])
            hid = ops.gemm_tokens(y, pk[f"b{i}.fc1.w"], pk[f"b{i}.fc1.b"], "gelu")
            ops.gemm_tokens(hid, pk[f"b{i}.fc2.w"], pk[f"b{i}.fc2.b"], "res", res=xw, out=xw)
        comb = ops.rt_patch_unembed(xw, pk["pu.w"], pk["pu.b"], feat_down)
        dec = ops.conv_c64(comb, pk["dec1.w"], pk["dec1.b"], 1, relu=True)
        residual = ops.conv_c64_thin(dec, pk["dec2.w"], pk["dec2.b"], 3, relu=False)
        out = ops.rt_bicubic_sum(x, residual, tuple(int(v) for v in res_out), clamp=True)   # model.py:125,160-164
        if torch.is_autocast_enabled():
            out = out.to(torch.get_autocast_gpu_dtype())
        return out
