"""``TransformerModel`` -- the reference's plugin surface on top of the MI355X HIP path.

Same constructor keywords, ``forward`` signature, parameter names/shapes and ``state_dict`` keys as
reference models/FastTransformer/model.py:174-327 (+ utils.py:43-98), so ``train.py`` /
``inference.py`` / ``speed_test.py``-style callers (``importlib.import_module(
"models.FastTransformer.model").TransformerModel()``) work unchanged.  The sub-modules below only
*own parameters*; all arithmetic runs in libtupscale_hip.so (there is no eager/CPU fallback).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import engine, ops, packing
from .weights import VALID_SCALES, upsampler_layout


# training: run the last x2 Upsampler stage + up1_conv (and their backward) through the exact composition; False = explicit kernels
compose_branch_a_in_training = True          # A/B attribute
use_pack_plan = True          # A/B attribute: training re-pack as two gather launches (pack_plan.py)


class _ConvParams(nn.Module):
    """Parameter holder with nn.Conv2d's names, shapes and default init."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool = True, transposed: bool = False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            fan_in = shape[1] * k * k
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)


class _PixelShuffleMarker(nn.Module):
    """Keeps the reference's Sequential numbering (conv at 0, 2, ...); the shuffle itself is fused
    into the conv kernels' stores."""

    def __init__(self, r: int):
        super().__init__()
        self.r = r


class Upsampler(nn.Module):
    """utils.py:43-98: ``upsamplers[str(scale)]`` = conv(+PixelShuffle) stacks for 2, 3, 4, 6."""

    def __init__(self, n_feats: int, valid_scales=VALID_SCALES):
        super().__init__()
        self.upsamplers = nn.ModuleDict()
        for s in valid_scales:
            seq = []
            for _, r in upsampler_layout(s):
                seq += [_ConvParams(n_feats, n_feats * r * r, 3), _PixelShuffleMarker(r)]
            self.upsamplers[str(s)] = nn.Sequential(*seq)


class BasicConv(nn.Module):
    """utils.py:13-40 as used by the model: ``.conv`` without bias, ReLU fused in the kernel."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = _ConvParams(cin, cout, 3, bias=False)


class _LinearParams(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin)
        nn.init.uniform_(self.bias, -bound, bound)


class _LayerNormParams(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class WindowAttention(nn.Module):
    """model.py:65-102 parameters + the int64 ``relative_position_index`` buffer."""

    def __init__(self, dim: int, window_size: int, num_heads: int):
        super().__init__()
        assert dim % num_heads == 0, "dim must be divisible by num_heads"
        self.qkv = _LinearParams(dim, 3 * dim)
        self.proj = _LinearParams(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        ys, xs = torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing="ij")
        ys, xs = ys.flatten(), xs.flatten()
        idx = (ys[:, None] - ys[None, :] + window_size - 1) * (2 * window_size - 1) + \
              (xs[:, None] - xs[None, :] + window_size - 1)
        self.register_buffer("relative_position_index", idx.long())
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class WindowTransformerBlock(nn.Module):
    def __init__(self, dim: int, window_size: int, num_heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = _LayerNormParams(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = _LayerNormParams(dim)
        hidden = int(dim * mlp_ratio)
        # indices 0 and 2 carry parameters, as in nn.Sequential(Linear, GELU, Linear, Dropout)
        self.mlp = nn.Sequential(_LinearParams(dim, hidden), nn.Identity(), _LinearParams(hidden, dim), nn.Identity())


class TransformerModel(nn.Module):
    def __init__(self, in_channels: int = 3, base_channels: int = 64, transformer_dim: int = 192,
                 num_window_blocks: int = 6, num_heads: int = 12, mlp_ratio: float = 4.0,
                 dropout: float = 0.1, window_size: int = 8):
        super().__init__()
        if (in_channels, base_channels, transformer_dim, num_window_blocks, num_heads, window_size) != (3, 64, 192, 6, 12, 8) \
                or float(mlp_ratio) != 4.0:
            raise NotImplementedError("the HIP kernels are specialised for the reference's defaults "
                                      "(3, 64, 192, 6 blocks, 12 heads, mlp 4.0, window 8)")
        self.dropout_p = float(dropout)
        self.window_size = window_size
        self.conv1 = _ConvParams(in_channels, base_channels, 3)
        self.conv2 = _ConvParams(base_channels, base_channels, 3)
        self.up1 = Upsampler(base_channels)
        self.up1_conv = BasicConv(base_channels, 3)
        self.final_upscale = Upsampler(3)
        self.final_upscale_conv = _ConvParams(3, 3, 3)
        self.patch_embed = _ConvParams(base_channels, transformer_dim, 8)
        self.window_blocks = nn.ModuleList(
            [WindowTransformerBlock(transformer_dim, window_size, num_heads, mlp_ratio) for _ in range(num_window_blocks)])
        self.patch_unembed = _ConvParams(transformer_dim, base_channels, 8, transposed=True)
        self.decoder_conv1 = _ConvParams(base_channels, base_channels, 3)
        self.decoder_conv2 = _ConvParams(base_channels, in_channels, 3)
        self._pack_cache: Dict[int, tuple] = {}
        self._dropout_calls = 0

    def _next_dropout(self):
        """(p, seed) for the next training forward: p = 0 in eval mode; the seed advances every call and is
        offset by torch's seed and the data-parallel rank so replicas draw different masks."""
        if not self.training or self.dropout_p <= 0.0:
            return 0.0, 0
        import os
        self._dropout_calls += 1
        base = (torch.initial_seed() + 7919 * int(os.environ.get("RANK", "0"))) & 0x7FFFFFFF
        return self.dropout_p, (base * 2654435761 + self._dropout_calls) & 0xFFFFFFFF

    # ---- packed-weight cache, invalidated by in-place parameter updates (optimizer steps) ----
    def _versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate_packed(self) -> None:
        """Drop the packed-weight cache.  Needed after writes that bypass autograd's version counter
        (``p.data.copy_(...)`` -- the EMA / weight-swap idiom -- or ``dist.broadcast(p.data)``): the cache key is
        (data_ptr, p._version) per parameter and such writes change neither.  Optimizer steps, ``load_state_dict`` and
        ``.to()`` are detected automatically."""
        self._pack_cache = {}

    def _pack_with_plan(self, sd, scale: int):
        """Training re-pack (once per optimizer step) through pack_plan.PackPlan: two gather launches instead of packing.py's
        ~75 torch launches.  The plan is traced from packing.pack_state_dict itself and verified bit for bit against it on the
        current weights before its first use; None (torch path) if it cannot be built."""
        from .pack_plan import packed_with_plan
        return packed_with_plan(self, scale, sd, lambda d: packing.pack_state_dict(d, scale, backward=True))

    def packed(self, scale: int, backward: bool = False):
        """Packed weights (+ dense relative-position biases) for `scale`; re-packed when any parameter
        was updated in place (optimizer step) or moved."""
        ver = self._versions()
        key = (scale, bool(backward))
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != ver:
            sd = {k: v for k, v in self.named_parameters()}
            pk = self._pack_with_plan(sd, scale) if backward and use_pack_plan else None
            if pk is None:
                pk = packing.pack_state_dict(sd, scale, backward=backward)
            nb = len(self.window_blocks)
            frags_t = [ops.relpos_bias_expand(pk[f"b{i}.table"]) for i in range(nb)]
            frags_n = [ops.relpos_bias_expand_n(pk[f"b{i}.table"]) for i in range(nb)] if backward else None
            if backward and compose_branch_a_in_training:
                from .weights import upsampler_layout as _layout
                idx, r = _layout(scale)[-1]
                if r == 2:         # training: last Upsampler stage + up1_conv as one composed conv (csrc/branch_a_train.hip)
                    k = f"up1.upsamplers.{scale}.{idx}"
                    pk["bra.wu"], pk["bra.bu"] = sd[k + ".weight"].detach().float().contiguous(), sd[k + ".bias"].detach().float().contiguous()
                    pk["bra.w3"] = sd["up1_conv.conv.weight"].detach().float().contiguous()
                    pk["bra.comp"] = ops.bra_compose(pk["bra.wu"], pk["bra.bu"], pk["bra.w3"])
            hit = (ver, pk, frags_t, frags_n)
            # one entry per (scale, backward) of the CURRENT weights: alternating-scale inference / mixed-scale training re-packs
            # nothing until a parameter changes (entries of an older version are dropped here)
            self._pack_cache = {k: v for k, v in self._pack_cache.items() if v[0] == ver}
            self._pack_cache[key] = hit
        return (hit[1], hit[2], hit[3]) if backward else (hit[1], hit[2])

    def forward(self, x: torch.Tensor, res_out: Tuple[int, int] = (1080, 1920), upscale_factor: Optional[int] = None,
                require_ratio: bool = True) -> torch.Tensor:
        res_out, scale = engine.resolve_scale(x.shape[2], x.shape[3], res_out, upscale_factor)
        if not x.is_cuda:
            raise RuntimeError("TransformerModel (MI355X build) runs on the GPU only: move the module and the "
                               "input to 'cuda'. There is no CPU fallback.")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from .autograd import fast_transformer_function   # training path (forward + hand-written backward)
            out = fast_transformer_function(self, x, scale, res_out, require_ratio)
        elif self.training and self.dropout_p > 0.0:
            # .train() without gradients (e.g. a validation pass that forgot .eval()): the reference's nn.Dropout layers are
            # active whenever the module is in training mode, so run the training forward (dropout in its kernels) and
            # drop what it saved
            from .autograd import forward_train
            pk, frags_t, _ = self.packed(scale, backward=True)
            drop_p, seed = self._next_dropout()
            out, _ = forward_train(pk, frags_t, x, scale, res_out, require_ratio, drop_p, seed)
        else:
            pk, frags = self.packed(scale)
            out = engine.forward(pk, frags, x, scale, res_out, require_ratio)
        if torch.is_autocast_enabled():
            out = out.to(torch.get_autocast_gpu_dtype())
        return out
