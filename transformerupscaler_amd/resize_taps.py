"""Tap tables of the antialiased bilinear resize (torchvision.transforms.Resize on tensors).

Product-side builder for tup_resize_aa_fwd.  Follows aten's float32 evaluation
(_compute_indices_weights_aa in UpSampleKernel.cpp, PyTorch being a third-party dependency of
the reference: requirements.txt:42) so results agree with the reference's CPU path to rounding.
Vectorised numpy, float32 throughout.
"""
from __future__ import annotations

import functools
import math

import numpy as np


@functools.lru_cache(maxsize=64)
def aa_taps(in_size: int, out_size: int):
    f = np.float32
    scale = f(in_size) / f(out_size)
    support = scale if scale >= 1.0 else f(1.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    kmax = int(math.ceil(float(support))) * 2 + 1
    i = np.arange(out_size, dtype=np.float32)
    center = (scale * (i + f(0.5))).astype(np.float32)
    lo = np.maximum(0, (center - support + f(0.5)).astype(np.float32).astype(np.int64))
    hi = np.minimum(in_size, (center + support + f(0.5)).astype(np.float32).astype(np.int64))
    n = (hi - lo).astype(np.int32)
    j = np.arange(kmax, dtype=np.int64)[None, :]
    arg = ((j + lo[:, None]).astype(np.float32) - center[:, None] + f(0.5)).astype(np.float32) * invscale
    w = np.maximum(f(0.0), f(1.0) - np.abs(arg)).astype(np.float32)
    w = np.where(j < n[:, None], w, f(0.0)).astype(np.float32)
    tot = np.zeros(out_size, np.float32)
    for k in range(kmax):           # sequential float32 accumulation, as aten does
        tot = (tot + w[:, k]).astype(np.float32)
    w = np.where(tot[:, None] != 0, w / np.where(tot == 0, f(1.0), tot)[:, None], w).astype(np.float32)
    return lo.astype(np.int32), n, np.ascontiguousarray(w), kmax


@functools.lru_cache(maxsize=64)
def aa_inverse_ranges(in_size: int, out_size: int):
    """For every input index the contiguous range [o0, o0+on) of output indices whose taps cover it
    (the transpose of the forward table; used by the resize backward)."""
    lo, n, _, _ = aa_taps(in_size, out_size)
    o0 = np.full(in_size, out_size, np.int32)
    o1 = np.zeros(in_size, np.int32)
    for o in range(out_size):
        a, b = int(lo[o]), int(lo[o] + n[o])
        o0[a:b] = np.minimum(o0[a:b], o)
        o1[a:b] = np.maximum(o1[a:b], o + 1)
    on = np.maximum(o1 - o0, 0).astype(np.int32)
    o0 = np.where(on > 0, o0, 0).astype(np.int32)
    # contiguity check: every output in [o0, o0+on) must really cover the input index
    for i in range(in_size):
        for o in range(int(o0[i]), int(o0[i] + on[i])):
            assert lo[o] <= i < lo[o] + n[o]
    return o0, on


@functools.lru_cache(maxsize=64)
def taps_or_identity(in_size: int, out_size: int):
    """aa_taps, or exact identity tables when the sizes match (transforms.Resize returns its input then)."""
    if in_size == out_size:
        return (np.arange(out_size, dtype=np.int32), np.ones(out_size, np.int32), np.ones((out_size, 1), np.float32), 1)
    return aa_taps(in_size, out_size)


@functools.lru_cache(maxsize=64)
def tile_extent(in_size: int, out_size: int, tile: int) -> int:
    """Largest input window any `tile`-wide run of outputs touches (sizes the fused tail kernel's LDS tiles)."""
    lo, n, _, _ = taps_or_identity(in_size, out_size)
    best = 1
    for o0 in range(0, out_size, tile):
        o1 = min(o0 + tile, out_size) - 1
        best = max(best, int(lo[o1] + n[o1] - lo[o0]))
    return best


@functools.lru_cache(maxsize=64)
def bicubic_taps(in_size: int, out_size: int):
    """F.interpolate(mode='bicubic', align_corners=False) (aten upsample_bicubic2d, A = -0.75): per output index
    four clamped source indices and float32 weights.  Third-party PyTorch semantics (reference
    models/ResidualTransformer/model.py:125,160 call it); vectorised numpy."""
    f = np.float32
    scale = f(in_size) / f(out_size)
    o = np.arange(out_size, dtype=np.float32)
    src = (scale * (o + f(0.5)) - f(0.5)).astype(np.float32)
    fl = np.floor(src)
    t = (src - fl).astype(np.float32)
    A = f(-0.75)
    c1 = lambda x: (((A + f(2)) * x - (A + f(3))) * x * x + f(1)).astype(np.float32)
    c2 = lambda x: (((A * x - f(5) * A) * x + f(8) * A) * x - f(4) * A).astype(np.float32)
    w = np.stack([c2(t + f(1)), c1(t), c1(f(1) - t), c2(f(2) - t)], axis=1).astype(np.float32)
    idx = np.clip(fl.astype(np.int64)[:, None] - 1 + np.arange(4)[None, :], 0, in_size - 1).astype(np.int32)
    return np.ascontiguousarray(idx), np.ascontiguousarray(w)


def transpose_taps(idx: np.ndarray, w: np.ndarray, in_size: int):
    """Per-output tap lists (idx, w: [out][K]) -> per-source CSR lists for the gather-form backward:
    start int32 [in+1], out_index int32 [nnz], weight float32 [nnz] (taps clamped onto the same source stay separate
    entries, so the backward adds exactly the terms the forward used)."""
    out_size, K = idx.shape
    flat_src = idx.reshape(-1).astype(np.int64)
    flat_out = np.repeat(np.arange(out_size, dtype=np.int64), K)
    order = np.argsort(flat_src, kind="stable")
    counts = np.bincount(flat_src, minlength=in_size)
    start = np.zeros(in_size + 1, dtype=np.int32)
    start[1:] = np.cumsum(counts)
    return start, np.ascontiguousarray(flat_out[order].astype(np.int32)), np.ascontiguousarray(w.reshape(-1)[order].astype(np.float32))


PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow's coefficient tables for an 8-bit BILINEAR resample of one axis (libImaging/Resample.c precompute_coeffs +
    normalize_coeffs_8bpc; the arithmetic behind transforms.Resize on a PIL image, reference data_handling/data_class.py:61-71):
    (min int32 [out], size int32 [out], k int32 [out][ksize], ksize).  Weights are computed and normalised in double and rounded
    half away from zero to 22 fractional bits, exactly as Pillow does."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    lo = np.zeros(out_size, np.int32)
    n = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    inv = 1.0 / fscale
    for i in range(out_size):
        center = (i + 0.5) * scale
        a = max(int(center - support + 0.5), 0)
        b = min(int(center + support + 0.5), in_size)
        ww = 0.0
        for x in range(b - a):
            t = abs((x + a - center + 0.5) * inv)
            w = 1.0 - t if t < 1.0 else 0.0
            kk[i, x] = w
            ww += w
        if ww != 0.0:
            kk[i, :b - a] /= ww
        lo[i], n[i] = a, b - a
    ki = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PIL_PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PIL_PRECISION_BITS)))
    return lo, n, np.ascontiguousarray(ki.astype(np.int32)), ksize
