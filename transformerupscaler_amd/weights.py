"""Parameter inventory and deterministic synthetic weights for FastTransformer.

Key names / shapes mirror the reference ``state_dict`` (SURVEY.md §8(b);
reference models/FastTransformer/model.py:189-229, utils.py:54-91) so that checkpoints
interchange.  There is no network here, so benchmarks, fixtures and tests use weights
from a closed-form seeded formula instead of a trained checkpoint.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

IN_CH, BASE, DIM, BLOCKS, HEADS, MLP, WINDOW = 3, 64, 192, 6, 12, 768, 8
VALID_SCALES = (2, 3, 4, 6)


def upsampler_layout(scale: int):
    """[(sequential index, r)] of the conv+PixelShuffle stages of one scale (utils.py:54-91)."""
    if scale == 2:
        return [(0, 2)]
    if scale == 4:
        return [(0, 2), (2, 2)]
    if scale in (3, 6):
        return [(0, scale)]
    raise ValueError(f"Requested scale={scale} was not built.")


def param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["conv1.weight"] = (BASE, IN_CH, 3, 3); s["conv1.bias"] = (BASE,)
    s["conv2.weight"] = (BASE, BASE, 3, 3); s["conv2.bias"] = (BASE,)
    for pre, nf in (("up1", BASE), ("final_upscale", IN_CH)):
        if pre == "final_upscale":
            s["up1_conv.conv.weight"] = (IN_CH, BASE, 3, 3)
        for sc in VALID_SCALES:
            for idx, r in upsampler_layout(sc):
                s[f"{pre}.upsamplers.{sc}.{idx}.weight"] = (nf * r * r, nf, 3, 3)
                s[f"{pre}.upsamplers.{sc}.{idx}.bias"] = (nf * r * r,)
    s["final_upscale_conv.weight"] = (IN_CH, IN_CH, 3, 3); s["final_upscale_conv.bias"] = (IN_CH,)
    s["patch_embed.weight"] = (DIM, BASE, 8, 8); s["patch_embed.bias"] = (DIM,)
    for i in range(BLOCKS):
        p = f"window_blocks.{i}"
        s[p + ".norm1.weight"] = (DIM,); s[p + ".norm1.bias"] = (DIM,)
        s[p + ".attn.relative_position_bias_table"] = ((2 * WINDOW - 1) ** 2, HEADS)
        s[p + ".attn.qkv.weight"] = (3 * DIM, DIM); s[p + ".attn.qkv.bias"] = (3 * DIM,)
        s[p + ".attn.proj.weight"] = (DIM, DIM); s[p + ".attn.proj.bias"] = (DIM,)
        s[p + ".norm2.weight"] = (DIM,); s[p + ".norm2.bias"] = (DIM,)
        s[p + ".mlp.0.weight"] = (MLP, DIM); s[p + ".mlp.0.bias"] = (MLP,)
        s[p + ".mlp.2.weight"] = (DIM, MLP); s[p + ".mlp.2.bias"] = (DIM,)
    s["patch_unembed.weight"] = (DIM, BASE, 8, 8); s["patch_unembed.bias"] = (BASE,)
    s["decoder_conv1.weight"] = (BASE, BASE, 3, 3); s["decoder_conv1.bias"] = (BASE,)
    s["decoder_conv2.weight"] = (IN_CH, BASE, 3, 3); s["decoder_conv2.bias"] = (IN_CH,)
    return s


def active_param_names(scale: int):
    """Parameters that receive a gradient when training at one fixed scale."""
    names = []
    for k in param_shapes():
        if ".upsamplers." in k:
            if int(k.split(".upsamplers.")[1].split(".")[0]) != scale:
                continue
        names.append(k)
    return names


def _fan_in(name: str, shape) -> int:
    if name == "patch_unembed.weight":      # ConvTranspose2d: weight (in, out, kh, kw)
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 4:
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return 0


def deterministic_state_dict(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded closed-form weights: uniform(+-1/sqrt(fan_in)) matrices (PyTorch's default
    Conv/Linear scale), LayerNorm gamma 1+-0.1 / beta +-0.1, rel-pos table N(0, 0.5) so the
    bias path is numerically visible, and a +0.5 offset on the last bias so outputs straddle
    the [0,1] clamp instead of saturating at 0."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    shapes = param_shapes()
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith("relative_position_bias_table"):
            t = torch.randn(shape, generator=g) * 0.5
        elif ".norm" in name and name.endswith("weight"):
            t = 1.0 + (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif ".norm" in name:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif name.endswith("bias"):
            wname = name[:-4] + "weight"
            bound = 1.0 / math.sqrt(_fan_in(wname, shapes[wname]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            if name == "final_upscale_conv.bias":
                t = t + 0.5
        else:
            bound = 1.0 / math.sqrt(_fan_in(name, shape))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            if name in ("up1_conv.conv.weight", "final_upscale_conv.weight"):
                t = t * 3.0     # widen the output spread so both clamp edges are exercised
        sd[name] = t.to(dtype)
    return sd


# ------------------------------------------------------------------------------------------------
# ResidualTransformer (reference models/ResidualTransformer/model.py:53-112; BASELINE.json config 5)
# ------------------------------------------------------------------------------------------------
RT_DIM, RT_BLOCKS, RT_HEADS, RT_MLP, RT_TOKENS_H, RT_TOKENS_W = 128, 8, 8, 512, 45, 80


def rt_param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["pos_embed"] = (1, RT_TOKENS_H * RT_TOKENS_W, RT_DIM)
    s["conv1.weight"] = (BASE, IN_CH, 3, 3); s["conv1.bias"] = (BASE,)
    s["conv2.weight"] = (BASE, BASE, 3, 3); s["conv2.bias"] = (BASE,)
    s["downsample.weight"] = (BASE, BASE, 3, 3); s["downsample.bias"] = (BASE,)
    s["patch_embed.weight"] = (RT_DIM, BASE, 8, 8); s["patch_embed.bias"] = (RT_DIM,)
    for i in range(RT_BLOCKS):
        p = f"transformer_blocks.{i}"
        s[p + ".norm1.weight"] = (RT_DIM,); s[p + ".norm1.bias"] = (RT_DIM,)
        s[p + ".attn.in_proj_weight"] = (3 * RT_DIM, RT_DIM); s[p + ".attn.in_proj_bias"] = (3 * RT_DIM,)
        s[p + ".attn.out_proj.weight"] = (RT_DIM, RT_DIM); s[p + ".attn.out_proj.bias"] = (RT_DIM,)
        s[p + ".norm2.weight"] = (RT_DIM,); s[p + ".norm2.bias"] = (RT_DIM,)
        s[p + ".mlp.0.weight"] = (RT_MLP, RT_DIM); s[p + ".mlp.0.bias"] = (RT_MLP,)
        s[p + ".mlp.2.weight"] = (RT_DIM, RT_MLP); s[p + ".mlp.2.bias"] = (RT_DIM,)
    s["patch_unembed.weight"] = (RT_DIM, BASE, 8, 8); s["patch_unembed.bias"] = (BASE,)
    s["decoder_conv1.weight"] = (BASE, BASE, 3, 3); s["decoder_conv1.bias"] = (BASE,)
    s["decoder_conv2.weight"] = (IN_CH, BASE, 3, 3); s["decoder_conv2.bias"] = (IN_CH,)
    return s


def rt_deterministic_state_dict(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Same recipe as deterministic_state_dict for the ResidualTransformer keys (pos_embed ~ N(0, 0.5))."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    shapes = rt_param_shapes()
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(("rt." + name).encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name == "pos_embed":
            t = torch.randn(shape, generator=g) * 0.5
        elif ".norm" in name and name.endswith("weight"):
            t = 1.0 + (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif ".norm" in name:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif name.endswith("bias"):
            wname = name[:-4] + "weight"
            fan = _fan_in(wname, shapes[wname]) if wname in shapes else RT_DIM
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan)
            if name == "decoder_conv2.bias":
                t = t * 0.1
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(_fan_in(name, shape))
        sd[name] = t.to(dtype)
    return sd


# ------------------------------------------------------------------------------------------------
# WindowTransformer plugin (reference models/WindowTransformer/model.py:172-225): state_dict layout
# ------------------------------------------------------------------------------------------------
WT_DIM, WT_HEADS, WT_BLOCKS, WT_MLP = 128, 8, 8, 512


def wt_param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["conv1.weight"] = (BASE, IN_CH, 3, 3); s["conv1.bias"] = (BASE,)
    s["conv2.weight"] = (BASE, BASE, 3, 3); s["conv2.bias"] = (BASE,)
    s["downsample.weight"] = (BASE, BASE, 3, 3); s["downsample.bias"] = (BASE,)
    s["patch_embed.weight"] = (WT_DIM, BASE, 8, 8); s["patch_embed.bias"] = (WT_DIM,)
    for i in range(WT_BLOCKS):
        p = f"window_blocks.{i}"
        s[p + ".norm1.weight"] = (WT_DIM,); s[p + ".norm1.bias"] = (WT_DIM,)
        s[p + ".attn.relative_position_bias_table"] = (225, WT_HEADS)
        s[p + ".attn.qkv.weight"] = (3 * WT_DIM, WT_DIM); s[p + ".attn.qkv.bias"] = (3 * WT_DIM,)
        s[p + ".attn.proj.weight"] = (WT_DIM, WT_DIM); s[p + ".attn.proj.bias"] = (WT_DIM,)
        s[p + ".norm2.weight"] = (WT_DIM,); s[p + ".norm2.bias"] = (WT_DIM,)
        s[p + ".mlp.0.weight"] = (WT_MLP, WT_DIM); s[p + ".mlp.0.bias"] = (WT_MLP,)
        s[p + ".mlp.2.weight"] = (WT_DIM, WT_MLP); s[p + ".mlp.2.bias"] = (WT_DIM,)
    s["patch_unembed.weight"] = (WT_DIM, BASE, 8, 8); s["patch_unembed.bias"] = (BASE,)
    s["decoder_conv1.weight"] = (BASE, BASE, 3, 3); s["decoder_conv1.bias"] = (BASE,)
    s["decoder_conv2.weight"] = (IN_CH, BASE, 3, 3); s["decoder_conv2.bias"] = (IN_CH,)
    return s


def wt_deterministic_state_dict(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Same recipe as deterministic_state_dict for the WindowTransformer keys (parameters only; the int64
    ``relative_position_index`` buffers are created by the module)."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    shapes = wt_param_shapes()
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(("wt." + name).encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith("relative_position_bias_table"):
            t = torch.randn(shape, generator=g) * 0.5
        elif ".norm" in name and name.endswith("weight"):
            t = 1.0 + (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif ".norm" in name:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif name.endswith("bias"):
            wname = name[:-4] + "weight"
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(_fan_in(wname, shapes[wname]))
            if name == "decoder_conv2.bias":
                t = t * 0.1
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(_fan_in(name, shape))
        sd[name] = t.to(dtype)
    return sd
