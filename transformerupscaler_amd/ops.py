"""Tensor-level wrappers over the C ABI (include/tupscale_hip.h).

Each wrapper validates device / dtype / shape / contiguity on the host (a kernel that faults
can take the whole GPU node down), allocates the output with torch (plumbing only) and launches
on the caller's current HIP stream.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from .resize_taps import aa_taps

BF16, F32 = torch.bfloat16, torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def _stream():
    """The current stream of the current device as the integer hipStream_t the C ABI takes.  (The raw accessor skips building a
    torch.cuda.Stream object per launch: the training step issues ~275 launches and its host time is within 10 % of its GPU time.)"""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, shape=None, name="tensor"):
    # the common case falls through five cheap tests; the messages are built only on failure
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if shape is not None and t.shape != shape and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a GPU tensor (no CPU fallback)")
    if t.device.index != _cur_dev():
        # every launch goes to the CURRENT device's stream: a tensor of another device would be a wild pointer there
        raise RuntimeError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(...) (one process per GPU) before building / calling the model")
    return t.data_ptr()


def _opt(t, dtype, shape, name):
    return None if t is None else _chk(t, dtype, shape, name)


# ---- zero pool: the backward pass needs ~130 small zero-initialised fp32 buffers per step (atomically accumulated weight /
# bias gradients).  One torch.zeros of the whole lot + views replaces 130 fill launches.  The pool is a fresh allocation per
# backward pass and stays alive as long as any gradient view does, so there is no reuse hazard. ----
_zero_pool = None          # [tensor, next free offset (floats)]
_ZERO_POOL_FLOATS = 9 * 1024 * 1024


def zero_pool_begin(device):
    global _zero_pool
    if _ZERO_POOL_FLOATS > 0:
        _zero_pool = [torch.zeros((_ZERO_POOL_FLOATS,), dtype=F32, device=device), 0]


def zero_pool_end():
    global _zero_pool
    _zero_pool = None


def _zeros(shape, device):
    n = 1
    for d in shape:
        n *= int(d)
    zp = _zero_pool
    if zp is not None and zp[0].device == torch.device(device) and zp[1] + n <= zp[0].numel():
        off = zp[1]
        zp[1] = off + ((n + 63) // 64) * 64          # 256-byte aligned slices
        return zp[0][off:off + n].view(shape)
    return torch.zeros(shape, dtype=F32, device=device)


def conv1(x, wp, bias, relu=True, in_mask=None, out_mask=None):
    """planar fp32 [B][3][H][W] -> NHWC bf16 64 ch (conv1; also the input-gradient conv of the 64->3 convs)."""
    B, C, H, W = x.shape
    assert C == 3
    out = torch.empty((B, H, W, 64), dtype=BF16, device=x.device)
    _lib.call("tup_conv3x3_c3_fwd", _chk(x, F32, None, "x"), _chk(wp, BF16, (64, 32), "wp"),
              _opt(bias, F32, (64,), "bias"), _opt(in_mask, F32, x.shape, "in_mask"),
              _opt(out_mask, BF16, out.shape, "out_mask"), out.data_ptr(), B, H, W, int(relu), _stream())
    return out


def conv_c64(x, wp, bias, r=1, relu=False, add=None, mask=None, in_r=1, out=None):
    """NHWC bf16 conv 64*in_r^2 -> 64*r*r with fused PixelShuffle(r); returns [B][H*r][W*r][64] bf16.
    x is [B][H*in_r][W*in_r][64] (in_r > 1: channels read through PixelShuffle^-1)."""
    B, Hi, Wi, C = x.shape
    assert C == 64 and Hi % in_r == 0 and Wi % in_r == 0
    H, W = Hi // in_r, Wi // in_r
    nt = r * r
    if out is None:
        out = torch.empty((B, H * r, W * r, 64), dtype=BF16, device=x.device)
    else:
        _chk(out, BF16, (B, H * r, W * r, 64), "out")
    _lib.call("tup_conv3x3_c64_fwd", _chk(x, BF16, None, "x"), _chk(wp, BF16, (nt, in_r * in_r, 9, 64, 64), "wp"),
              _opt(bias, F32, (nt, 64), "bias"), _opt(add, BF16, out.shape, "add"), _opt(mask, BF16, out.shape, "mask"),
              out.data_ptr(), B, H, W, nt, r, 64, int(relu), 0, in_r, _stream())
    return out


def conv_c64_thin(x, wp, bias, cout, relu=False, out=None):
    """NHWC bf16 conv 64 -> cout (<=16); returns fp32 planar [B][cout][H][W]."""
    B, H, W, C = x.shape
    assert C == 64 and 1 <= cout <= 16
    if out is None:
        out = torch.empty((B, cout, H, W), dtype=F32, device=x.device)
    else:
        _chk(out, F32, (B, cout, H, W), "out")
    _lib.call("tup_conv3x3_c64_fwd", _chk(x, BF16, None, "x"), _chk(wp, BF16, (1, 1, 9, 16, 64), "wp"),
              _opt(bias, F32, (cout,), "bias"), None, None, out.data_ptr(), B, H, W, 1, 1, cout, int(relu), 1, 1, _stream())
    return out


def branch_a_composed(x, wp, bias, wv, bv, r, relu=True):
    """Composed (last up-conv + PixelShuffle + up1_conv [+ReLU]) 5x5 conv: NHWC bf16 [B][H][W][64] -> fp32 [B][3][H*r][W*r]."""
    B, H, W, C = x.shape
    assert C == 64 and r in (2, 3, 6)
    rows, n = {2: 16, 3: 32, 6: 112}[r], 3 * r * r
    out = torch.empty((B, 3, H * r, W * r), dtype=F32, device=x.device)
    _lib.call("tup_conv5x5_c64_planar_fwd", _chk(x, BF16, None, "x"), _chk(wp, BF16, (1, 1, 25, rows, 64), "wp"),
              _chk(bias, F32, (n,), "bias"), _chk(wv, BF16, (9, n, 25, 64), "wv"), _chk(bv, F32, (9, n), "bv"),
              out.data_ptr(), B, H, W, r, int(relu), _stream())
    return out


def conv_planar(x, w28, bias, r=1, add=None, clamp=False):
    """clamp = "both" (training): returns (unclamped, clamped), written by the same kernel pass."""
    B, C, H, W = x.shape
    assert C == 3
    cout = 3 * r * r
    both = clamp == "both"
    shape = (B, 3, H * r, W * r)
    out = torch.empty(((2,) if both else ()) + shape, dtype=F32, device=x.device)
    _lib.call("tup_conv3x3_planar_fwd", _chk(x, F32, None, "x"), _chk(w28, F32, (cout, 28), "w28"),
              _opt(bias, F32, (cout,), "bias"), _opt(add, F32, shape, "add"), out.data_ptr(),
              B, H, W, r, 2 if both else int(bool(clamp)), _stream())
    return (out[0], out[1]) if both else out


_TAP_CACHE = {}


def _taps_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _TAP_CACHE:
        lo, n, w, k = aa_taps(in_size, out_size)
        _TAP_CACHE[key] = (torch.from_numpy(lo).to(device), torch.from_numpy(n).to(device),
                           torch.from_numpy(w).to(device), k)
    return _TAP_CACHE[key]


def resize_aa(x, size, clamp=False):
    """clamp = "both" (training): returns (unclamped, clamped), written by the same kernel pass."""
    B, C, Hi, Wi = x.shape
    Ho, Wo = size
    ylo, yn, yw, ky = _taps_on(x.device, Hi, Ho)
    xlo, xn, xw, kx = _taps_on(x.device, Wi, Wo)
    both = clamp == "both"
    out = torch.empty(((2,) if both else ()) + (B, C, Ho, Wo), dtype=F32, device=x.device)
    _lib.call("tup_resize_aa_fwd", _chk(x, F32, None, "x"), out.data_ptr(), ylo.data_ptr(), yn.data_ptr(),
              yw.data_ptr(), ky, xlo.data_ptr(), xn.data_ptr(), xw.data_ptr(), kx, B * C, Hi, Wi, Ho, Wo,
              2 if both else int(bool(clamp)), _stream())
    return (out[0], out[1]) if both else out


TAIL_TILE_H = 16          # = OT_H of csrc/tail_fused.hip


def tail_fused(x, wfu, bfu, wfc, bfc, ui, r, out_hw, clamp=True):
    """Last final_upscale stage + final_upscale_conv + "+ upscaled_input" + Resize(out_hw) + clamp in one kernel."""
    from .resize_taps import taps_or_identity, tile_extent
    B, C, H, W = x.shape
    Hs, Ws = H * r, W * r
    Ho, Wo = out_hw
    key = (str(x.device), "tail", Hs, Ws, Ho, Wo)
    if key not in _TAP_CACHE:
        ylo, yn, yw, ky = taps_or_identity(Hs, Ho)
        xlo, xn, xw, kx = taps_or_identity(Ws, Wo)
        d = x.device
        _TAP_CACHE[key] = (torch.from_numpy(ylo).to(d), torch.from_numpy(yn).to(d), torch.from_numpy(np.ascontiguousarray(yw)).to(d), ky,
                           torch.from_numpy(xlo).to(d), torch.from_numpy(xn).to(d), torch.from_numpy(np.ascontiguousarray(xw)).to(d), kx,
                           tile_extent(Hs, Ho, TAIL_TILE_H), tile_extent(Ws, Wo, 64))
    ylo, yn, yw, ky, xlo, xn, xw, kx, eh, ew = _TAP_CACHE[key]
    out = torch.empty((B, 3, Ho, Wo), dtype=F32, device=x.device)
    _lib.call("tup_tail_fused_fwd", _chk(x, F32, None, "x"), _chk(wfu, F32, (3 * r * r, 28), "wfu"), _chk(bfu, F32, (3 * r * r,), "bfu"),
              _chk(wfc, F32, (3, 28), "wfc"), _chk(bfc, F32, (3,), "bfc"), _chk(ui, F32, (B, 3, Hs, Ws), "ui"), out.data_ptr(),
              ylo.data_ptr(), yn.data_ptr(), yw.data_ptr(), ky, xlo.data_ptr(), xn.data_ptr(), xw.data_ptr(), kx,
              B, H, W, r, Ho, Wo, eh, ew, int(clamp), _stream())
    return out


TAIL_STREAM_WAVES_PER_SIMD = 3      # csrc/tail_stream.hip: 150 registers, 52 KB of LDS per four-wave workgroup


def _tail_stream_plan(device, B, H, W, Ho, Wo):
    """Decomposition of the fused streaming tail + Resize (csrc/tail_stream.hip, RESIZE = true): strip stride, band height, the
    output columns / rows each strip / band owns.  None when a tap table has more than 4 taps (then the Resize runs as its own kernel)."""
    key = (str(device), "tail_stream", B, H, W, Ho, Wo)
    if key not in _TAP_CACHE:
        from .resize_taps import aa_taps
        ylo, yn, yw, ky = aa_taps(2 * H, Ho)
        xlo, xn, xw, kx = aa_taps(2 * W, Wo)
        plan = None
        if int(yn.max()) <= 4 and int(xn.max()) <= 4:
            sc = 60 - (int(xn.max()) - 1 + 1) // 2
            ext = (int(yn.max()) - 1 + 1) // 2
            nstrip = (W + sc - 1) // sc
            nb = max(1, (256 * 4 * TAIL_STREAM_WAVES_PER_SIMD) // max(1, B * nstrip))      # fill the chip once at the kernel's occupancy
            bh = max(12, ((H + nb - 1) // nb + 2) // 3 * 3)
            nband = (H + bh - 1) // bh
            oxb = np.searchsorted(xlo, 2 * sc * np.arange(nstrip + 1), side="left").astype(np.int32)
            oyb = np.searchsorted(ylo, 2 * bh * np.arange(nband + 1), side="left").astype(np.int32)
            oxb[-1], oyb[-1] = Wo, Ho
            ok = True
            for s_ in range(nstrip):           # every owned column's taps inside the strip's 120 valid HR columns, <= 128 gathered
                a, b_ = int(oxb[s_]), int(oxb[s_ + 1])
                if b_ > a:
                    ok &= bool((xlo[a:b_] + xn[a:b_]).max() <= 2 * s_ * sc + 120) and b_ - a <= 128
            for k_ in range(nband):            # every owned row's taps inside the rows the band computes
                a, b_ = int(oyb[k_]), int(oyb[k_ + 1])
                if b_ > a:
                    ok &= bool((ylo[a:b_] + yn[a:b_]).max() <= 2 * min(H, (k_ + 1) * bh + ext))
            if ok:
                t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(device)
                plan = (t(ylo), t(yn), t(yw), ky, t(xlo), t(xn), t(xw), kx, t(oxb), t(oyb), sc, bh, ext)
        _TAP_CACHE[key] = plan
    return _TAP_CACHE[key]


def tail_stream_fits(B, H, W, out_hw=None):
    """The streaming tail addresses its tensors with 32-bit byte offsets: tup_tail_stream_r2_fwd refuses B*12*H*W >= 2^29 elements
    (the HR map of the last stage), tup_tail_stream_r2_resize_fwd also B*3*Ho*Wo >= 2^29 (csrc/tail_stream.hip).  Batches beyond that
    (4x of 720p from B = 13, 540p -> 2160p from B = 22) take the tiled tail (tail_fused), which has no such limit."""
    if B * 12 * H * W >= 1 << 29:
        return False
    if out_hw is not None and tuple(out_hw) != (2 * H, 2 * W) and B * 3 * int(out_hw[0]) * int(out_hw[1]) >= 1 << 29:
        return False
    return True


def tail_stream_r2(x, wfu_t, bfu, wfc_t, bfc, ui, clamp=True, out_hw=None):
    """Last final_upscale stage (r = 2) + final_upscale_conv + "+ upscaled_input" [+ antialiased Resize to out_hw] [+ clamp],
    streaming kernel.  Returns None when out_hw needs a Resize whose tap tables the fused kernel does not take (> 4 taps)."""
    B, C, H, W = x.shape
    args = (_chk(x, F32, (B, 3, H, W), "x"), _chk(wfu_t, F32, (27, 12), "wfu_t"), _chk(bfu, F32, (12,), "bfu"),
            _chk(wfc_t, F32, (27, 4), "wfc_t"), _chk(bfc, F32, (3,), "bfc"), _chk(ui, F32, (B, 3, 2 * H, 2 * W), "ui"))
    if out_hw is None or tuple(out_hw) == (2 * H, 2 * W):
        out = torch.empty((B, 3, 2 * H, 2 * W), dtype=F32, device=x.device)
        _lib.call("tup_tail_stream_r2_fwd", *args, out.data_ptr(), B, H, W, int(clamp), _stream())
        return out
    Ho, Wo = int(out_hw[0]), int(out_hw[1])
    plan = _tail_stream_plan(x.device, B, H, W, Ho, Wo)
    if plan is None:
        return None
    ylo, yn, yw, ky, xlo, xn, xw, kx, oxb, oyb, sc, bh, ext = plan
    out = torch.empty((B, 3, Ho, Wo), dtype=F32, device=x.device)
    _lib.call("tup_tail_stream_r2_resize_fwd", *args, out.data_ptr(), ylo.data_ptr(), yn.data_ptr(), yw.data_ptr(), ky,
              xlo.data_ptr(), xn.data_ptr(), xw.data_ptr(), kx, oxb.data_ptr(), oyb.data_ptr(), B, H, W, Ho, Wo, sc, bh, ext,
              int(clamp), _stream())
    return out


def clamp01(x):
    out = torch.empty_like(x)
    _lib.call("tup_clamp01_fwd", _chk(x, F32, None, "x"), out.data_ptr(), x.numel(), _stream())
    return out


def layernorm(x, gamma, beta, save_stats=False):
    M, D = x.shape
    assert D == 192
    y = torch.empty((M, 192), dtype=BF16, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty((M,), dtype=F32, device=x.device)
        rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.call("tup_layernorm_fwd", _chk(x, F32, None, "x"), _chk(gamma, F32, (192,), "gamma"),
              _chk(beta, F32, (192,), "beta"), y.data_ptr(), None if mean is None else mean.data_ptr(),
              None if rstd is None else rstd.data_ptr(), M, _stream())
    return (y, mean, rstd) if save_stats else y


def relpos_bias_expand(table):
    frag = torch.empty((12, 4, 4, 64, 4), dtype=F32, device=table.device)
    _lib.call("tup_relpos_bias_expand", _chk(table, F32, (225, 12), "table"), frag.data_ptr(), _stream())
    return frag


def window_attn(qkv, bias_frag, drop_p=0.0, drop_seed=0, save_lse=False):
    """WindowAttention core.  save_lse (training): also returns the fp32 [M // 64][12][64] log-sum-exp of every score row, which
    window_attn_bwd rebuilds the probabilities from."""
    M, D = qkv.shape
    assert D == 576 and M % 64 == 0
    out = torch.empty((M, 192), dtype=BF16, device=qkv.device)
    lse = torch.empty((M // 64, 12, 64), dtype=F32, device=qkv.device) if save_lse else None
    _lib.call("tup_window_attn_fwd", _chk(qkv, BF16, None, "qkv"), _chk(bias_frag, F32, (12, 4, 4, 64, 4), "bias"),
              out.data_ptr(), lse.data_ptr() if save_lse else None, M // 64, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    return (out, lse) if save_lse else out


def gemm_tokens(a, wt, bias, epilogue, res=None, out=None, aux=None, drop_p=0.0, drop_seed=0):
    """epilogue 'bf16' | 'gelu' (aux, if given, receives the bf16 pre-activation) | 'res' (fp32 out = a@wt^T + bias
    + res) | 'gelu_bwd' (bf16 out = (a@wt^T) * gelu'(aux))."""
    M, K = a.shape
    N = wt.shape[0]
    assert tuple(wt.shape) == (N, K) and N % 64 == 0 and K % 64 == 0
    a_dtype = {BF16: 0, F32: 1}[a.dtype]
    epi = {"bf16": 0, "gelu": 1, "res": 2, "gelu_bwd": 3}[epilogue]
    resp = auxp = None
    if epi == 2:
        if out is None:
            out = torch.empty((M, N), dtype=F32, device=a.device)
        _chk(out, F32, (M, N), "out")
        resp = _chk(res, F32, (M, N), "res")
    else:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
        if epi == 3 or (epi == 1 and aux is not None):
            auxp = _chk(aux, BF16, (M, N), "aux")
    _lib.call("tup_gemm_tokens_fwd", _chk(a, a.dtype, None, "a"), a_dtype, K, _chk(wt, BF16, None, "wt"),
              _opt(bias, F32, (N,), "bias"), resp, auxp, out.data_ptr(), N, M, N, K, epi, float(drop_p),
              int(drop_seed) & 0xFFFFFFFF, _stream())
    return out


def ln_gemm(x, gamma, beta, wt, bias):
    """bf16 [M][N] = LayerNorm(x) @ wt^T + bias (LN fused into the GEMM's A load)."""
    M = x.shape[0]
    N = wt.shape[0]
    out = torch.empty((M, N), dtype=BF16, device=x.device)
    _lib.call("tup_ln_gemm_fwd", _chk(x, F32, (M, 192), "x"), _chk(gamma, F32, (192,), "gamma"), _chk(beta, F32, (192,), "beta"),
              _chk(wt, BF16, (N, 192), "wt"), _chk(bias, F32, (N,), "bias"), out.data_ptr(), M, N, _stream())
    return out


def fused_qkv_attn(x, gamma, beta, wh, bh, bias_frag):
    """bf16 [M][192] = attention core of qkv(LayerNorm(x)) per 8x8 window (inference fusion; the qkv tensor never exists)."""
    M = x.shape[0]
    assert M % 64 == 0
    out = torch.empty((M, 192), dtype=BF16, device=x.device)
    _lib.call("tup_fused_qkv_attn_fwd", _chk(x, F32, (M, 192), "x"), _chk(gamma, F32, (192,), "gamma"), _chk(beta, F32, (192,), "beta"),
              _chk(wh, BF16, (12, 64, 192), "wh"), _chk(bh, F32, (12, 48), "bh"), _chk(bias_frag, F32, (12, 4, 4, 64, 4), "bias"),
              out.data_ptr(), M // 64, _stream())
    return out


def fused_attn_block(x, gamma, beta, wh, bh, bias_frag, wproj, bproj):
    """In place: x += proj(attention(qkv(LayerNorm(x)))) + b_proj (the attention half of a block, inference fusion)."""
    M = x.shape[0]
    assert M % 64 == 0
    _lib.call("tup_fused_attn_block_fwd", _chk(x, F32, (M, 192), "x"), _chk(gamma, F32, (192,), "gamma"), _chk(beta, F32, (192,), "beta"),
              _chk(wh, BF16, (12, 64, 192), "wh"), _chk(bh, F32, (12, 48), "bh"), _chk(bias_frag, F32, (12, 4, 4, 64, 4), "bias"),
              _chk(wproj, BF16, (192, 192), "wproj"), _chk(bproj, F32, (192,), "bproj"), M // 64, _stream())
    return x


F16 = torch.float16


def block_table(blocks):
    """Pointer table for tup_fused_blocks32_fwd: `blocks` = per block the 9 tensors (wh, bh, bias_frag, wproj, bproj, w1, b1, w2, b2)
    with norm1 / norm2 folded into attn.qkv / mlp.0 (packing.fold_layernorm), w1 / b1 = mlp.0 / 4 (packing.pack_fc1_fused_q),
    w2 = 4 W2 in fp16 (packing.pack_fc2_h4), validated here.  Returns (ctypes array [nblk*9] of device pointers, nblk, the
    tensors -- kept alive by the caller holding the tuple)."""
    import ctypes
    shapes = [(BF16, (12, 64, 192)), (F32, (12, 48)), (F32, (12, 4, 4, 64, 4)), (BF16, (192, 192)), (F32, (192,)),
              (BF16, (768, 192)), (F32, (768,)), (F16, (192, 768)), (F32, (192,))]
    if not 1 <= len(blocks) <= 8:
        raise ValueError("1..8 blocks per launch")
    ptrs = []
    for blk in blocks:
        assert len(blk) == 9
        ptrs += [_chk(t, dt, sh, f"block operand {i}") for i, (t, (dt, sh)) in enumerate(zip(blk, shapes))]
    return ((ctypes.c_void_p * len(ptrs))(*ptrs), len(blocks), [list(b) for b in blocks])


def fused_blocks32(x, table):
    """In place: the table's consecutive WindowTransformerBlocks in one launch, two waves per window (the default kernel)."""
    M = x.shape[0]
    assert M % 64 == 0
    arr, nblk, _keep = table
    _lib.call("tup_fused_blocks32_fwd", _chk(x, F32, (M, 192), "x"), arr, nblk, M // 64, _stream())
    return x


def clock_probe():
    """Two device int64 [2]: (shader cycles, 100 MHz ticks) of the stream position of the call; see bench.py `sustained`."""
    out = torch.zeros(2, dtype=torch.int64, device=torch.device("cuda", _cur_dev()))
    _lib.call("tup_clock_probe", out.data_ptr(), _stream())
    return out


def stream_table(blocks):
    """Pointer table for tup_blocks_stream_fwd: per block the 7 tensors of packing.pack_stream_block (wqk, wv, wproj, w1, w2, tab,
    sbias), validated here.  Returns (ctypes array, nblk, the tensors -- kept alive by the caller holding the tuple)."""
    import ctypes
    shapes = [(BF16, 12 * 6144), (BF16, 6 * 6144), (BF16, 6 * 6144), (BF16, 24 * 6144), (F16, 24 * 6144), (F32, 1536), (F32, 12 * 2 * 2 * 64 * 16)]
    if not 1 <= len(blocks) <= 8:
        raise ValueError("1..8 blocks per launch")
    ptrs = []
    for blk in blocks:
        assert len(blk) == 7
        for i, (t, (dt, n)) in enumerate(zip(blk, shapes)):
            if t.numel() != n:
                raise ValueError(f"stream block operand {i}: {t.numel()} elements, expected {n}")
            ptrs.append(_chk(t, dt, None, f"stream block operand {i}"))
    return ((ctypes.c_void_p * len(ptrs))(*ptrs), len(blocks), [list(b) for b in blocks])


def blocks_stream(x, table, out_bf16=False):
    """The table's consecutive WindowTransformerBlocks in one launch of the streamed 32x32x16 kernel (csrc/block_stream.hip).
    out_bf16=False: in place, returns x.  out_bf16=True: returns the result as a NEW bf16 [M][192] tensor (round to nearest even of
    the same values) and leaves x holding the kernel's parked intermediate -- for patch_unembed, whose GEMM rounds its operand to
    bf16 anyway and then reads half the bytes."""
    M = x.shape[0]
    assert M % 64 == 0
    arr, nblk, _keep = table
    out = torch.empty((M, 192), dtype=BF16, device=x.device) if out_bf16 else None
    _lib.call("tup_blocks_stream_fwd", _chk(x, F32, (M, 192), "x"), out.data_ptr() if out_bf16 else None, arr, nblk, M // 64, _stream())
    return out if out_bf16 else x


def fused_block(x, wh, bh, bias_frag, wproj, bproj, w1, b1, w2, b2):
    """In place: one whole WindowTransformerBlock (attention half + MLP half) in one kernel (inference fusion).  Operands as the
    rows of block_table: LayerNorm scale / shift folded into wh / bh and w1 / b1 (packing.fold_layernorm)."""
    M = x.shape[0]
    assert M % 64 == 0
    _lib.call("tup_fused_block_fwd", _chk(x, F32, (M, 192), "x"),
              _chk(wh, BF16, (12, 64, 192), "wh"), _chk(bh, F32, (12, 48), "bh"), _chk(bias_frag, F32, (12, 4, 4, 64, 4), "bias"),
              _chk(wproj, BF16, (192, 192), "wproj"), _chk(bproj, F32, (192,), "bproj"),
              _chk(w1, BF16, (768, 192), "w1"), _chk(b1, F32, (768,), "b1"), _chk(w2, F16, (192, 768), "w2"), _chk(b2, F32, (192,), "b2"),
              M // 64, _stream())
    return x


def fused_mlp(x, gamma, beta, w1, b1, w2, b2):
    """In place: x += mlp.2(GELU(mlp.0(LayerNorm(x)))) (inference fusion; hidden tensor stays on chip).
    w1 / b1 = mlp.0 scaled by 1/4 (packing.pack_fc1_fused_q), w2 = 4 mlp.2.weight in fp16 (packing.pack_fc2_h4)."""
    M = x.shape[0]
    _lib.call("tup_fused_mlp_fwd", _chk(x, F32, (M, 192), "x"), _chk(gamma, F32, (192,), "gamma"), _chk(beta, F32, (192,), "beta"),
              _chk(w1, BF16, (768, 192), "w1"), _chk(b1, F32, (768,), "b1"), _chk(w2, F16, (192, 768), "w2"),
              _chk(b2, F32, (192,), "b2"), M, _stream())
    return x


def window_geometry(H, W):
    ht, wt = (H + 7) // 8, (W + 7) // 8
    nwy, nwx = (ht + 7) // 8, (wt + 7) // 8
    return ht, wt, nwy, nwx


def patch_embed(feat, wt, bias):
    B, H, W, C = feat.shape
    assert C == 64
    if (8 - H % 8) % 8 >= H or (8 - W % 8) % 8 >= W:
        raise RuntimeError("reflect padding needs pad < dim")     # same rule as F.pad(mode='reflect')
    _, _, nwy, nwx = window_geometry(H, W)
    x = torch.empty((B * nwy * nwx * 64, 192), dtype=F32, device=feat.device)
    _lib.call("tup_patch_embed_fwd", _chk(feat, BF16, None, "feat"), _chk(wt, BF16, (192, 4096), "wt"),
              _chk(bias, F32, (192,), "bias"), x.data_ptr(), B, H, W, _stream())
    return x


def patch_unembed(x, wt, bias, skip):
    B, H, W, C = skip.shape
    _, _, nwy, nwx = window_geometry(H, W)
    out = torch.empty_like(skip)
    x16 = x.dtype == BF16            # the streamed block kernel's bf16 output (ops.blocks_stream(out_bf16=True))
    _lib.call("tup_patch_unembed_fwd", _chk(x, BF16 if x16 else F32, (B * nwy * nwx * 64, 192), "x"), int(x16), _chk(wt, BF16, (4096, 192), "wt"),
              _chk(bias, F32, (64,), "bias"), _chk(skip, BF16, None, "skip"), out.data_ptr(), B, H, W, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# backward wrappers
# ------------------------------------------------------------------------------------------------
def gemm_wgrad(p, q, out=None):
    """out[NI][NJ] fp32 (+)= p^T q; p [M][NI], q [M][NJ] (bf16 or fp32)."""
    M, NI = p.shape
    NJ = q.shape[1]
    assert q.shape[0] == M and NI % 64 == 0 and NJ % 64 == 0
    if out is None:
        out = _zeros((NI, NJ), p.device)
    _lib.call("tup_gemm_wgrad", _chk(p, p.dtype, None, "p"), {BF16: 0, F32: 1}[p.dtype], NI,
              _chk(q, q.dtype, None, "q"), {BF16: 0, F32: 1}[q.dtype], NJ, _chk(out, F32, (NI, NJ), "out"), NJ,
              M, NI, NJ, _stream())
    return out


def gemm_wgrad_bias(p, q):
    """(dW [NI][NJ], db [NI]) = (p^T q, column sums of p): weight and bias gradient of a Linear layer in one pass over p."""
    M, NI = p.shape
    NJ = q.shape[1]
    assert q.shape[0] == M and NI % 64 == 0 and NJ % 64 == 0
    out = _zeros((NI, NJ), p.device)
    db = _zeros((NI,), p.device)
    _lib.call("tup_gemm_wgrad_bias", _chk(p, p.dtype, None, "p"), {BF16: 0, F32: 1}[p.dtype], NI,
              _chk(q, q.dtype, None, "q"), {BF16: 0, F32: 1}[q.dtype], NJ, out.data_ptr(), NJ, db.data_ptr(),
              M, NI, NJ, _stream())
    return out, db


PATCH_WGRAD_WIDE = True          # A/B switch: the wide-tile kernel (tup_patch_wgrad_bf16) for the two patch weights


def patch_wgrad(p, fmap, reflect):
    B, H, W, C = fmap.shape
    _, _, nwy, nwx = window_geometry(H, W)
    out = _zeros((192, 4096), p.device)
    if PATCH_WGRAD_WIDE and B * H * W * 128 < 2 ** 31:
        # bf16 token rows (the rounding the fp32 entry applies on load): the wide kernel fetches its operands by DMA
        _chk(p, F32, (B * nwy * nwx * 64, 192), "p")
        pb = p.to(BF16)
        _lib.call("tup_patch_wgrad_bf16", pb.data_ptr(), _chk(fmap, BF16, None, "map"), out.data_ptr(), B, H, W, int(reflect), _stream())
        return out
    _lib.call("tup_patch_wgrad", _chk(p, F32, (B * nwy * nwx * 64, 192), "p"), _chk(fmap, BF16, None, "map"),
              out.data_ptr(), B, H, W, int(reflect), _stream())
    return out


def colsum(g, out=None, rowmask=None):
    M, N = g.shape
    if out is None:
        out = _zeros((N,), g.device)
    _lib.call("tup_colsum", _chk(g, g.dtype, None, "g"), {BF16: 0, F32: 1}[g.dtype], N, _chk(out, F32, (N,), "out"),
              M, N, _opt(rowmask, torch.uint8, (M,), "rowmask"), _stream())
    return out


def layernorm_bwd(gy, x, mean, rstd, gamma, gres=None, drop=None):
    """drop = (p, seed): also returns dx * dropout mask / (1 - p) as bf16 (dropout_bwd of dx, fused): (dx, dgamma, dbeta, gdrop)."""
    M = x.shape[0]
    dx = torch.empty((M, 192), dtype=F32, device=x.device)
    dg = _zeros((192,), x.device)
    db = _zeros((192,), x.device)
    gd = torch.empty((M, 192), dtype=BF16, device=x.device) if drop else None
    _lib.call("tup_layernorm_bwd", _chk(gy, BF16, (M, 192), "gy"), _chk(x, F32, (M, 192), "x"), _chk(mean, F32, (M,), "mean"),
              _chk(rstd, F32, (M,), "rstd"), _chk(gamma, F32, (192,), "gamma"), _opt(gres, F32, (M, 192), "gres"),
              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, gd.data_ptr() if drop else None,
              float(drop[0]) if drop else 0.0, (int(drop[1]) & 0xFFFFFFFF) if drop else 0, _stream())
    return (dx, dg, db, gd) if drop else (dx, dg, db)


def relpos_bias_expand_n(table):
    frag = torch.empty((12, 4, 4, 64, 4), dtype=F32, device=table.device)
    _lib.call("tup_relpos_bias_expand_n", _chk(table, F32, (225, 12), "table"), frag.data_ptr(), _stream())
    return frag


def dropout_bwd(g, drop_p, drop_seed):
    """bf16 [M][192] = fp32 g * mask / (1 - p) for the site keyed by drop_seed."""
    out = torch.empty(g.shape, dtype=BF16, device=g.device)
    _lib.call("tup_dropout_bwd", _chk(g, F32, None, "g"), out.data_ptr(), g.numel(), float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    return out


def _attn_bwd_scratch(nwin, heads):
    """Floats of scratch the attention backward needs (tup_window_attn_bwd_scratch returns the count, not an error code)."""
    fn = _lib.load().tup_window_attn_bwd_scratch
    return int(fn(int(nwin), int(heads)))


def window_attn_bwd(qkv, gout, att, lse, bias_n, drop_p=0.0, drop_seed=0):
    """att, lse: what window_attn(..., save_lse=True) returned.  Returns (gqkv bf16 [M][576], dtable fp32 [225][12])."""
    M = qkv.shape[0]
    assert M % 64 == 0
    gqkv = torch.empty((M, 576), dtype=BF16, device=qkv.device)
    dbias = torch.empty((12, 4, 4, 64, 4), dtype=F32, device=qkv.device)
    scratch = torch.empty(_attn_bwd_scratch(M // 64, 12), dtype=F32, device=qkv.device)
    _lib.call("tup_window_attn_bwd", _chk(qkv, BF16, (M, 576), "qkv"), _chk(gout, BF16, (M, 192), "gout"),
              _chk(att, BF16, (M, 192), "att"), _chk(lse, F32, (M // 64, 12, 64), "lse"), _chk(bias_n, F32, (12, 4, 4, 64, 4), "bias_n"),
              gqkv.data_ptr(), dbias.data_ptr(), scratch.data_ptr(), M // 64, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    dtable = torch.empty((225, 12), dtype=F32, device=qkv.device)
    _lib.call("tup_relpos_bias_reduce", dbias.data_ptr(), dtable.data_ptr(), _stream())
    return gqkv, dtable


def patch_unembed_bwd(gmap, wt):
    B, H, W, C = gmap.shape
    _, _, nwy, nwx = window_geometry(H, W)
    gx = torch.empty((B * nwy * nwx * 64, 192), dtype=F32, device=gmap.device)
    _lib.call("tup_patch_unembed_bwd", _chk(gmap, BF16, None, "gmap"), _chk(wt, BF16, (192, 4096), "wt"),
              gx.data_ptr(), B, H, W, _stream())
    return gx


def patch_embed_bwd(gx, wt, B, H, W):
    ht, wt_, nwy, nwx = window_geometry(H, W)
    gmap = torch.empty((B, ht * 8, wt_ * 8, 64), dtype=BF16, device=gx.device)
    _lib.call("tup_patch_embed_bwd", _chk(gx, F32, (B * nwy * nwx * 64, 192), "gx"), _chk(wt, BF16, (4096, 192), "wt"),
              gmap.data_ptr(), B, H, W, _stream())
    return gmap


def patch_embed_bwd_merge(gx, wt, add1, add2, relu_src, want_add1_colsum=False):
    """patch_embed's input gradient + the gradient merge at `feat` (feat_grad_combine) in one kernel; H, W multiples of 8.
    want_add1_colsum: also returns the fp32 [64] column sums of add1 (colsum(add1.view(-1, 64)) for free)."""
    B, H, W, C = relu_src.shape
    assert C == 64 and H % 8 == 0 and W % 8 == 0
    _, _, nwy, nwx = window_geometry(H, W)
    out = torch.empty_like(relu_src)
    cs = _zeros((16, 64), relu_src.device) if want_add1_colsum else None
    _lib.call("tup_patch_embed_bwd_merge", _chk(gx, F32, (B * nwy * nwx * 64, 192), "gx"), _chk(wt, BF16, (4096, 192), "wt"),
              _chk(add1, BF16, relu_src.shape, "add1"), _opt(add2, BF16, relu_src.shape, "add2"), _chk(relu_src, BF16, None, "relu_src"),
              out.data_ptr(), cs.data_ptr() if want_add1_colsum else None, B, H, W, _stream())
    return (out, cs.sum(0)) if want_add1_colsum else out


def conv_c64_wgrad(x, gmap, gr=1):
    """-> (dwp fp32 [gr*gr][64][9][64] (sp, co, tap, ci), dbias fp32 [gr*gr][64])."""
    B, H, W, C = x.shape
    assert C == 64 and tuple(gmap.shape) == (B, H * gr, W * gr, 64)
    dwp = _zeros((gr * gr, 64, 9, 64), x.device)
    db = _zeros((gr * gr, 64), x.device)
    for sp in range(gr * gr):
        _lib.call("tup_conv3x3_c64_wgrad", _chk(x, BF16, None, "x"), _chk(gmap, BF16, None, "gmap"), dwp[sp].data_ptr(),
                  db[sp].data_ptr(), B, H, W, gr, sp, _stream())
    return dwp, db


def conv_thin_wgrad(x, gpl, want_bias):
    B, H, W, C = x.shape
    assert C == 64
    dwp = _zeros((3, 9, 64), x.device)
    db = _zeros((3,), x.device) if want_bias else None
    _lib.call("tup_conv3x3_thin_wgrad", _chk(x, BF16, None, "x"), _chk(gpl, F32, (B, 3, H, W), "gpl"), dwp.data_ptr(),
              None if db is None else db.data_ptr(), B, H, W, _stream())
    return dwp, db


def conv1_wgrad(x, gmap):
    """Weight / bias gradient of conv1 (3 -> 64): x planar fp32 [B][3][H][W], gmap NHWC bf16 [B][H][W][64].
    dw[co][ci][tap] = sum_p g[p][co] x[p + tap - 1][ci] = sum_q x[q][ci] g[q - (tap - 1)][co]: the thin (cout = 3) MFMA weight-
    gradient kernel with the roles of the two maps swapped and the taps flipped, plus a column sum for the bias."""
    B, C, H, W = x.shape
    assert C == 3
    dwp, _ = conv_thin_wgrad(gmap, x, False)                     # [ci][tap'][co], tap' = 8 - tap
    dw = dwp.flip(1).permute(2, 0, 1).reshape(64, 3, 3, 3)
    db = colsum(gmap.view(-1, 64))
    return dw, db


def conv1_wgrad_direct(x, gmap):
    """The original VALU kernel (tup_conv3x3_c3_wgrad), kept for the kernel-level test."""
    B, C, H, W = x.shape
    assert C == 3
    dw = _zeros((64, 3, 3, 3), x.device)
    db = _zeros((64,), x.device)
    _lib.call("tup_conv3x3_c3_wgrad", _chk(x, F32, None, "x"), _chk(gmap, BF16, (B, H, W, 64), "gmap"), dw.data_ptr(),
              db.data_ptr(), B, H, W, _stream())
    return dw, db


def conv_planar_wgrad(x, gpl, r):
    B, C, H, W = x.shape
    cout = 3 * r * r
    dw = _zeros((cout, 3, 3, 3), x.device)
    db = _zeros((cout,), x.device)
    _lib.call("tup_conv3x3_planar_wgrad", _chk(x, F32, None, "x"), _chk(gpl, F32, (B, 3, H * r, W * r), "gpl"),
              dw.data_ptr(), db.data_ptr(), B, H, W, r, _stream())
    return dw, db


def conv_planar_dgrad(gpl, w, r):
    B, C, Hr, Wr = gpl.shape
    H, W = Hr // r, Wr // r
    gx = torch.empty((B, 3, H, W), dtype=F32, device=gpl.device)
    _lib.call("tup_conv3x3_planar_dgrad", _chk(gpl, F32, None, "gpl"), _chk(w, F32, (3 * r * r, 3, 3, 3), "w"),
              gx.data_ptr(), B, H, W, r, _stream())
    return gx


_INV_CACHE = {}


def _inv_taps_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _INV_CACHE:
        from .resize_taps import aa_inverse_ranges
        o0, on = aa_inverse_ranges(in_size, out_size)
        _INV_CACHE[key] = (torch.from_numpy(o0).to(device), torch.from_numpy(on).to(device))
    return _INV_CACHE[key]


def resize_aa_bwd(gout, in_hw, pre=None, l1_scale=None):
    """Backward of resize_aa (+clamp when `pre`, the pre-clamp output, is given).  l1_scale (device float[2] = {d loss / numel,
    plain}): `gout` is the L1 target and the loss gradient is formed inside the kernel (autograd.l1_loss(...,
    fuse_into_model_backward=True)); plain != 0: `pre` is the loss input itself (this Resize's output), no clamp in between."""
    B, C, Ho, Wo = gout.shape
    Hi, Wi = in_hw
    ylo, _, yw, ky = _taps_on(gout.device, Hi, Ho)
    xlo, _, xw, kx = _taps_on(gout.device, Wi, Wo)
    oy0, oyn = _inv_taps_on(gout.device, Hi, Ho)
    ox0, oxn = _inv_taps_on(gout.device, Wi, Wo)
    gin = torch.empty((B, C, Hi, Wi), dtype=F32, device=gout.device)
    _lib.call("tup_resize_aa_bwd", _chk(gout, F32, None, "gout"), _opt(pre, F32, gout.shape, "pre"), gin.data_ptr(),
              ylo.data_ptr(), yw.data_ptr(), ky, xlo.data_ptr(), xw.data_ptr(), kx, oy0.data_ptr(), oyn.data_ptr(),
              ox0.data_ptr(), oxn.data_ptr(), B * C, Hi, Wi, Ho, Wo, _opt(l1_scale, F32, (2,), "l1_scale"), _stream())
    return gin


def mask_bwd(gout, pre=None, relu_src=None, l1_scale=None):
    gin = torch.empty_like(gout)
    _lib.call("tup_mask_bwd", _chk(gout, F32, None, "gout"), _opt(pre, F32, gout.shape, "pre"),
              _opt(relu_src, F32, gout.shape, "relu_src"), gin.data_ptr(), gout.numel(), _opt(l1_scale, F32, (2,), "l1_scale"), _stream())
    return gin


def feat_grad_combine(a, b, gpe, feat):
    B, H, W, C = feat.shape
    hp, wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
    out = torch.empty_like(feat)
    _lib.call("tup_feat_grad_combine", _chk(a, BF16, feat.shape, "a"), _opt(b, BF16, feat.shape, "b"),
              _chk(gpe, BF16, (B, hp, wp, 64), "gpe"), _chk(feat, BF16, None, "feat"), out.data_ptr(), B, H, W, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# ResidualTransformer wrappers
# ------------------------------------------------------------------------------------------------
def rt_patch_embed(feat, wt, bias, pos):
    B, H, W, C = feat.shape
    T = (H // 8) * (W // 8)
    x = torch.empty((B * T, 128), dtype=F32, device=feat.device)
    _lib.call("tup_rt_patch_embed_fwd", _chk(feat, BF16, None, "feat"), _chk(wt, BF16, (128, 4096), "wt"), _chk(bias, F32, (128,), "bias"),
              _chk(pos, F32, (T, 128), "pos"), x.data_ptr(), B, H, W, _stream())
    return x


def rt_patch_unembed(x, wt, bias, skip):
    B, H, W, C = skip.shape
    out = torch.empty_like(skip)
    _lib.call("tup_rt_patch_unembed_fwd", _chk(x, F32, (B * (H // 8) * (W // 8), 128), "x"), _chk(wt, BF16, (4096, 128), "wt"),
              _chk(bias, F32, (64,), "bias"), _chk(skip, BF16, None, "skip"), out.data_ptr(), B, H, W, _stream())
    return out


def _rt_dropout_check(N, drop_p):
    # the attention-probability mask decides two neighbouring keys per hash and a lane holds four consecutive keys (csrc/common.h
    # drop_pair4): the kernels take dropout only on token counts that are multiples of 4 (the reference's 720 x 1280 input = 3600
    # tokens, models/ResidualTransformer/model.py:135-140); p is quantised to multiples of 1 / 65536
    if drop_p > 0.0 and N % 4 != 0:
        raise ValueError(f"ResidualTransformer attention dropout needs a token count that is a multiple of 4 (got N = {N}); "
                         "call .eval() or use an input whose token grid has a multiple of 4 tokens")


def rt_attention(qkv, B, N, save_lse=False, drop_p=0.0, drop_seed=0):
    _rt_dropout_check(N, drop_p)
    out = torch.empty((B * N, 128), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, 8, N), dtype=F32, device=qkv.device) if save_lse else None
    _lib.call("tup_rt_attention_fwd", _chk(qkv, BF16, (B * N, 384), "qkv"), out.data_ptr(),
              None if lse is None else lse.data_ptr(), B, N, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    return (out, lse) if save_lse else out


def rt_attention_bwd(qkv, out, gout, lse, B, N, drop_p=0.0, drop_seed=0):
    _rt_dropout_check(N, drop_p)
    gqkv = torch.empty((B * N, 384), dtype=BF16, device=qkv.device)
    work = torch.empty((B, 8, N), dtype=F32, device=qkv.device)
    _lib.call("tup_rt_attention_bwd", _chk(qkv, BF16, (B * N, 384), "qkv"), _chk(out, BF16, (B * N, 128), "out"),
              _chk(gout, BF16, (B * N, 128), "gout"), _chk(lse, F32, (B, 8, N), "lse"), work.data_ptr(), gqkv.data_ptr(), B, N,
              float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    return gqkv


def layernorm128(x, gamma, beta, save_stats=False):
    M = x.shape[0]
    y = torch.empty((M, 128), dtype=BF16, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty((M,), dtype=F32, device=x.device)
        rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.call("tup_layernorm128_fwd", _chk(x, F32, (M, 128), "x"), _chk(gamma, F32, (128,), "gamma"), _chk(beta, F32, (128,), "beta"),
              y.data_ptr(), None if mean is None else mean.data_ptr(), None if rstd is None else rstd.data_ptr(), M, _stream())
    return (y, mean, rstd) if save_stats else y


def layernorm128_bwd(gy, x, mean, rstd, gamma, gres=None, drop=None):
    """As layernorm_bwd for 128-wide rows."""
    M = x.shape[0]
    dx = torch.empty((M, 128), dtype=F32, device=x.device)
    dg = _zeros((128,), x.device)
    db = _zeros((128,), x.device)
    gd = torch.empty((M, 128), dtype=BF16, device=x.device) if drop else None
    _lib.call("tup_layernorm128_bwd", _chk(gy, BF16, (M, 128), "gy"), _chk(x, F32, (M, 128), "x"), _chk(mean, F32, (M,), "mean"),
              _chk(rstd, F32, (M,), "rstd"), _chk(gamma, F32, (128,), "gamma"), _opt(gres, F32, (M, 128), "gres"),
              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, gd.data_ptr() if drop else None,
              float(drop[0]) if drop else 0.0, (int(drop[1]) & 0xFFFFFFFF) if drop else 0, _stream())
    return (dx, dg, db, gd) if drop else (dx, dg, db)


def rt_patch_wgrad(p, fmap):
    """fp32 [128][4096] = p^T patches(fmap); p fp32 [B*T][128] (plain token grid), fmap NHWC bf16."""
    B, H, W, C = fmap.shape
    out = _zeros((128, 4096), p.device)
    _lib.call("tup_rt_patch_wgrad", _chk(p, F32, (B * (H // 8) * (W // 8), 128), "p"), _chk(fmap, BF16, None, "map"),
              out.data_ptr(), B, H, W, _stream())
    return out


def rt_patch_unembed_bwd(gmap, wd):
    """d tokens fp32 [B*T][128] of patch_unembed: the patch_embed GEMM with the transposed weight (no bias / pos)."""
    B, H, W, C = gmap.shape
    x = torch.empty((B * (H // 8) * (W // 8), 128), dtype=F32, device=gmap.device)
    _lib.call("tup_rt_patch_embed_fwd", _chk(gmap, BF16, None, "gmap"), _chk(wd, BF16, (128, 4096), "wd"), None, None,
              x.data_ptr(), B, H, W, _stream())
    return x


def rt_patch_embed_bwd(gx, wd, add):
    """d feat_down NHWC bf16 = add + scatter(gx wd^T): the patch_unembed GEMM with the transposed weight; `add` carries
    the gradient of the skip connection (model.py:153)."""
    B, H, W, C = add.shape
    out = torch.empty_like(add)
    _lib.call("tup_rt_patch_unembed_fwd", _chk(gx, F32, (B * (H // 8) * (W // 8), 128), "gx"), _chk(wd, BF16, (4096, 128), "wd"),
              None, _chk(add, BF16, None, "add"), out.data_ptr(), B, H, W, _stream())
    return out


def conv_c64_wgrad_s2d(x, gmap, xr):
    """Stride-xr conv weight gradient: fp32 [xr*xr][64 co][9][64 ci] block-tap gradients + bias gradient [64]."""
    B, H, W, C = gmap.shape
    assert tuple(x.shape) == (B, H * xr, W * xr, 64)
    dwp = _zeros((xr * xr, 64, 9, 64), x.device)
    db = _zeros((64,), x.device)
    for sp in range(xr * xr):
        _lib.call("tup_conv3x3_c64_wgrad_s2d", _chk(x, BF16, None, "x"), _chk(gmap, BF16, None, "gmap"), dwp[sp].data_ptr(),
                  db.data_ptr() if sp == 0 else None, B, H, W, xr, sp, _stream())
    return dwp, db


_BIC_CACHE = {}


def _bicubic_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _BIC_CACHE:
        from .resize_taps import bicubic_taps
        idx, w = bicubic_taps(in_size, out_size)
        _BIC_CACHE[key] = (torch.from_numpy(idx).to(device), torch.from_numpy(w).to(device))
    return _BIC_CACHE[key]


_BICT_CACHE = {}


def _bicubic_t_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _BICT_CACHE:
        from .resize_taps import bicubic_taps, transpose_taps
        idx, w = bicubic_taps(in_size, out_size)
        _BICT_CACHE[key] = tuple(torch.from_numpy(t).to(device) for t in transpose_taps(idx, w, in_size))
    return _BICT_CACHE[key]


_BICB_CACHE = {}
_BIC_YB = 16
bicubic_bwd_banded = True        # A/B attribute: per-source-row gather of the row pass


def _bicubic_bands_on(device, in_size, out_size):
    """Band tables of the row pass of rt_bicubic_bwd: for source rows 16b .. 16b+15 the contiguous range of output rows that touch
    them and the dense weights [rows][16] (transpose of the forward tap matrix)."""
    key = (str(device), in_size, out_size)
    if key not in _BICB_CACHE:
        import numpy as np
        from .resize_taps import bicubic_taps
        idx, w = bicubic_taps(in_size, out_size)             # [out][4]
        nb = (in_size + _BIC_YB - 1) // _BIC_YB
        lo = np.full(in_size, out_size, dtype=np.int64); hi = np.full(in_size, -1, dtype=np.int64)
        for k in range(4):
            np.minimum.at(lo, idx[:, k], np.arange(out_size)); np.maximum.at(hi, idx[:, k], np.arange(out_size))
        r0 = np.array([lo[b * _BIC_YB:(b + 1) * _BIC_YB].min() for b in range(nb)], dtype=np.int64)
        r1 = np.array([hi[b * _BIC_YB:(b + 1) * _BIC_YB].max() for b in range(nb)], dtype=np.int64)
        n = (r1 - r0 + 1).clip(min=1)
        r0 = np.minimum(r0, out_size - 1)
        nr_max = int(n.max())
        bw = np.zeros((nb, nr_max, _BIC_YB), dtype=np.float32)
        rows = np.arange(out_size)
        for k in range(4):
            y = idx[:, k]; b = y // _BIC_YB
            np.add.at(bw, (b, rows - r0[b], y - b * _BIC_YB), w[:, k].astype(np.float32))
        _BICB_CACHE[key] = (torch.from_numpy(r0.astype(np.int32)).to(device), torch.from_numpy(n.astype(np.int32)).to(device),
                            torch.from_numpy(bw).to(device), nr_max)
    return _BICB_CACHE[key]


_BICC_CACHE = {}


def _bicubic_cols_on(device, in_size, out_size):
    """Dense column tables of rt_bicubic_bwd: (xoT int32 [kmax][in], xwT fp32 [kmax][in], kmax, blk_c0, blk_n) from the transposed
    tap lists; None when a 256-column block's stretch exceeds the kernel's LDS tile (very large ratios)."""
    key = (str(device), in_size, out_size)
    if key not in _BICC_CACHE:
        import numpy as np
        from .resize_taps import bicubic_taps, transpose_taps
        idx, w = bicubic_taps(in_size, out_size)
        xs, xo, xw = transpose_taps(idx, w, in_size)
        cnt = np.diff(xs)
        kmax = int(cnt.max())
        xoT = np.zeros((kmax, in_size), dtype=np.int32); xwT = np.zeros((kmax, in_size), dtype=np.float32)
        for x in range(in_size):
            n = cnt[x]
            xoT[:n, x] = xo[xs[x]:xs[x] + n]; xwT[:n, x] = xw[xs[x]:xs[x] + n]
            xoT[n:, x] = xo[xs[x]]                                  # padding: in-range index, weight 0
        nblk = (in_size + 255) // 256
        c0 = np.array([xoT[:, j * 256:(j + 1) * 256].min() for j in range(nblk)], dtype=np.int32)
        c1 = np.array([xoT[:, j * 256:(j + 1) * 256].max() for j in range(nblk)], dtype=np.int32)
        nn = c1 - c0 + 1
        _BICC_CACHE[key] = None if int(nn.max()) > 4096 else (
            torch.from_numpy(xoT).to(device), torch.from_numpy(xwT).to(device), kmax, torch.from_numpy(c0).to(device),
            torch.from_numpy(nn.astype(np.int32)).to(device))
    return _BICC_CACHE[key]


def rt_bicubic_bwd(gout, out, in_hw, l1_scale=None):
    """Gradient of clamp(bicubic(src -> size) + ...) w.r.t. a planar fp32 source of size in_hw; `out` = the saved
    forward output (the clamp gate) or None.  l1_scale (fp32 device scalar): `gout` is then the TARGET of an L1 loss on `out`
    and the upstream gradient sign(out - target) * l1_scale is formed inside the kernel."""
    B, C, Ho, Wo = gout.shape
    Ha, Wa = in_hw
    ga = torch.empty((B, C, Ha, Wa), dtype=F32, device=gout.device)
    tmp = torch.empty((B, C, Ha, Wo), dtype=F32, device=gout.device)
    cols = _bicubic_cols_on(gout.device, Wa, Wo) if bicubic_bwd_banded else None
    if cols is not None:
        r0, bn, bw, nr_max = _bicubic_bands_on(gout.device, Ha, Ho)
        xoT, xwT, kmax, c0, cn = cols
        _lib.call("tup_rt_bicubic_bwd_banded", _chk(gout, F32, None, "gout"), _opt(out, F32, (B, C, Ho, Wo), "out"), ga.data_ptr(),
                  tmp.data_ptr(), r0.data_ptr(), bn.data_ptr(), bw.data_ptr(), nr_max, xoT.data_ptr(), xwT.data_ptr(), kmax,
                  c0.data_ptr(), cn.data_ptr(), B * C, Ha, Wa, Ho, Wo,
                  None if l1_scale is None else _chk(l1_scale, F32, None, "l1_scale"), _stream())
        return ga
    if l1_scale is not None:
        raise RuntimeError("the fused L1 form of rt_bicubic_bwd needs the banded kernels (ratio too large or TUP_BICUBIC_BWD_GATHER set)")
    xs, xo, xw = _bicubic_t_on(gout.device, Wa, Wo)
    ys, yo, yw = _bicubic_t_on(gout.device, Ha, Ho)
    _lib.call("tup_rt_bicubic_bwd", _chk(gout, F32, None, "gout"), _opt(out, F32, (B, C, Ho, Wo), "out"), ga.data_ptr(), tmp.data_ptr(),
              ys.data_ptr(), yo.data_ptr(), yw.data_ptr(), xs.data_ptr(), xo.data_ptr(), xw.data_ptr(), B * C, Ha, Wa, Ho, Wo, _stream())
    return ga


def rt_bicubic_sum(a, b, size, clamp=True):
    """clamp(bicubic(a -> size) + bicubic(b -> size)) for planar fp32 [B][3][H][W] tensors."""
    B, C, Ha, Wa = a.shape
    _, _, Hb, Wb = b.shape
    Ho, Wo = size
    ayi, ayw = _bicubic_on(a.device, Ha, Ho); axi, axw = _bicubic_on(a.device, Wa, Wo)
    byi, byw = _bicubic_on(a.device, Hb, Ho); bxi, bxw = _bicubic_on(a.device, Wb, Wo)
    out = torch.empty((B, C, Ho, Wo), dtype=F32, device=a.device)
    _lib.call("tup_rt_bicubic_sum_fwd", _chk(a, F32, None, "a"), _chk(b, F32, (B, C, Hb, Wb), "b"), out.data_ptr(),
              ayi.data_ptr(), ayw.data_ptr(), axi.data_ptr(), axw.data_ptr(), byi.data_ptr(), byw.data_ptr(),
              bxi.data_ptr(), bxw.data_ptr(), B * C, Ha, Wa, Hb, Wb, Ho, Wo, int(clamp), _stream())
    return out


# ---- frame pre/post-processing ----
def frames_to_tensor(frames_u8, bgr=False):
    """uint8 [B][H][W][3] (or [H][W][3]) on the GPU -> fp32 [B][3][H][W] in [0, 1] (ToTensor); bgr=True for BGR frames."""
    if frames_u8.dim() == 3:
        frames_u8 = frames_u8.unsqueeze(0)
    B, H, W, C = frames_u8.shape
    assert C == 3
    out = torch.empty((B, 3, H, W), dtype=F32, device=frames_u8.device)
    _lib.call("tup_u8hwc_to_f32chw", _chk(frames_u8, torch.uint8, None, "frames"), out.data_ptr(), B, H, W, int(bgr), _stream())
    return out


def tensor_to_frames(x, bgr=False):
    """fp32 [B][3][H][W] -> uint8 [B][H][W][3] = trunc(clamp(x * 255, 0, 255)); bgr=True writes BGR (app_overlay.py:381-388)."""
    B, C, H, W = x.shape
    assert C == 3
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=x.device)
    _lib.call("tup_f32chw_to_u8hwc", _chk(x, F32, None, "x"), out.data_ptr(), B, H, W, int(bgr), _stream())
    return out


# ---- branch A in training through its composition (csrc/branch_a_train.hip), r = 2 ----
def bra_compose(wu, bu, w3):
    """Composed weights of (Conv2d(64, 256, 3) + PixelShuffle(2) + Conv2d(64, 3, 3, bias=False)) in the kernels' layouts:
    dict(wp, bias, wv, bv, wd)."""
    dev = wu.device
    wv = torch.empty((9, 12, 25, 64), dtype=BF16, device=dev)
    bv = torch.empty((9, 12), dtype=F32, device=dev)
    wp = torch.zeros((1, 1, 25, 16, 64), dtype=BF16, device=dev)
    wd = torch.zeros((13, 64, 32), dtype=BF16, device=dev)
    _lib.call("tup_bra_compose", _chk(wu, F32, (256, 64, 3, 3), "wu"), _chk(bu, F32, (256,), "bu"), _chk(w3, F32, (3, 64, 3, 3), "w3"),
              wv.data_ptr(), bv.data_ptr(), wp.data_ptr(), wd.data_ptr(), _stream())
    return {"wp": wp, "bias": bv[0].contiguous(), "wv": wv, "bv": bv, "wd": wd}


def bra_backward(g, ui, feat, comp, wu, bu, w3):
    """g: fp32 [B][3][2H][2W] gradient w.r.t. upscaled_input (pre-mask), ui: upscaled_input, feat: bf16 [B][H][W][64].
    Returns (dfeat bf16 [B][H][W][64], dwu, dbu, dw3)."""
    B, H, W, C = feat.shape
    assert C == 64 and tuple(g.shape) == (B, 3, 2 * H, 2 * W)
    dev = feat.device
    g12 = torch.empty((B, H, W, 16), dtype=BF16, device=dev)
    dfeat = torch.empty((B, H, W, 64), dtype=BF16, device=dev)
    G = _zeros((16, 9, 12, 25, 64), dev)          # 16 replicas against atomic contention, summed by tup_bra_chain
    Gb = _zeros((16, 9, 12), dev)
    _lib.call("tup_bra_backward", _chk(g, F32, None, "g"), _chk(ui, F32, g.shape, "ui"), _chk(feat, BF16, None, "feat"),
              _chk(comp["wd"], BF16, (13, 64, 32), "wd"), _chk(comp["wv"], BF16, (9, 12, 25, 64), "wv"),
              g12.data_ptr(), dfeat.data_ptr(), G.data_ptr(), Gb.data_ptr(), B, H, W, _stream())
    dM = torch.empty((3 * 9 * 4 * 9 * 64,), dtype=F32, device=dev)
    dMb = torch.empty((108,), dtype=F32, device=dev)
    dwu = torch.empty((256, 64, 3, 3), dtype=F32, device=dev)
    dbu = torch.empty((256,), dtype=F32, device=dev)
    dw3 = torch.empty((3, 64, 3, 3), dtype=F32, device=dev)
    _lib.call("tup_bra_chain", G.data_ptr(), Gb.data_ptr(), _chk(wu, F32, (256, 64, 3, 3), "wu"), _chk(bu, F32, (256,), "bu"),
              _chk(w3, F32, (3, 64, 3, 3), "w3"), dM.data_ptr(), dMb.data_ptr(), dwu.data_ptr(), dbu.data_ptr(), dw3.data_ptr(), _stream())
    return dfeat, dwu, dbu, dw3, G, Gb


_PIL_TAPS = {}


def _pil_taps_on(device, in_size, out_size):
    from .resize_taps import pil_bilinear_coeffs
    key = (str(device), in_size, out_size)
    if key not in _PIL_TAPS:
        lo, n, k, ks = pil_bilinear_coeffs(in_size, out_size)
        _PIL_TAPS[key] = (torch.from_numpy(lo).to(device), torch.from_numpy(n).to(device), torch.from_numpy(k).to(device), ks)
    return _PIL_TAPS[key]


def resize_frames(frames_u8, size, to_tensor=False, bgr=False):
    """transforms.Resize(size) on uint8 frames [B][H][W][3] (or [H][W][3]) on the GPU, bit-exact with Pillow's BILINEAR resample
    of an RGB image (data_handling/data_class.py:61-71, inference.py:65-75).  to_tensor=False -> uint8 [B][h][w][3];
    to_tensor=True -> fp32 [B][3][h][w] in [0, 1] (Resize + ToTensor in the same launches)."""
    if frames_u8.dim() == 3:
        frames_u8 = frames_u8.unsqueeze(0)
    B, H, W, C = frames_u8.shape
    assert C == 3
    h, w = int(size[0]), int(size[1])
    cur = frames_u8
    _chk(cur, torch.uint8, None, "frames")
    if w != W:                                            # Pillow: horizontal pass first, into a uint8 image
        lo, n, k, ks = _pil_taps_on(cur.device, W, w)
        tmp = torch.empty((B, H, w, 3), dtype=torch.uint8, device=cur.device)
        _lib.call("tup_resize_u8_rows", cur.data_ptr(), tmp.data_ptr(), lo.data_ptr(), n.data_ptr(), k.data_ptr(), ks, B, H, W, w, _stream())
        cur = tmp
    if h != H:
        lo, n, k, ks = _pil_taps_on(cur.device, H, h)
        out_u8 = None if to_tensor else torch.empty((B, h, w, 3), dtype=torch.uint8, device=cur.device)
        out_f = torch.empty((B, 3, h, w), dtype=F32, device=cur.device) if to_tensor else None
        _lib.call("tup_resize_u8_cols", cur.data_ptr(), None if out_u8 is None else out_u8.data_ptr(),
                  None if out_f is None else out_f.data_ptr(), lo.data_ptr(), n.data_ptr(), k.data_ptr(), ks, B, H, w, h, int(bgr), _stream())
        return out_f if to_tensor else out_u8
    return frames_to_tensor(cur, bgr=bgr) if to_tensor else (cur if cur is not frames_u8 else cur.clone())


# ---- WindowTransformer: window block at width 128 / 8 heads ----
def relpos_bias_expand_h(table, heads):
    frag = torch.empty((heads, 4, 4, 64, 4), dtype=F32, device=table.device)
    _lib.call("tup_relpos_bias_expand_h", _chk(table, F32, (225, heads), "table"), frag.data_ptr(), heads, _stream())
    return frag


def window_attn_h(qkv, bias_frag, heads, drop_p=0.0, drop_seed=0, save_lse=False):
    M, D = qkv.shape
    assert D == 48 * heads and M % 64 == 0
    out = torch.empty((M, 16 * heads), dtype=BF16, device=qkv.device)
    lse = torch.empty((M // 64, heads, 64), dtype=F32, device=qkv.device) if save_lse else None
    _lib.call("tup_window_attn_fwd_h", _chk(qkv, BF16, None, "qkv"), _chk(bias_frag, F32, (heads, 4, 4, 64, 4), "bias"),
              out.data_ptr(), lse.data_ptr() if save_lse else None, M // 64, heads, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    return (out, lse) if save_lse else out


def wt_patch_embed(feat, wt, bias):
    """Stride-8 patch conv without padding (floor(H/8) x floor(W/8) tokens) -> fp32 window-layout tokens [M][N]."""
    B, H, W, C = feat.shape
    N = wt.shape[0]
    nwy, nwx = (H // 8 + 7) // 8, (W // 8 + 7) // 8
    x = torch.empty((B * nwy * nwx * 64, N), dtype=F32, device=feat.device)
    _lib.call("tup_wt_patch_embed_fwd", _chk(feat, BF16, None, "feat"), _chk(wt, BF16, (N, 4096), "wt"), _chk(bias, F32, (N,), "bias"),
              x.data_ptr(), B, H, W, N, _stream())
    return x


def wt_patch_unembed(x, wt, bias, skip):
    """skip + ConvTranspose(k8, s8)(window_reverse(x)): skip / result NHWC bf16 [B][Ht*8][Wt*8][64]."""
    B, Hs, Ws, C = skip.shape
    K = wt.shape[1]
    nwy, nwx = (Hs // 8 + 7) // 8, (Ws // 8 + 7) // 8
    out = torch.empty_like(skip)
    _lib.call("tup_wt_patch_unembed_fwd", _chk(x, F32, (B * nwy * nwx * 64, K), "x"), _chk(wt, BF16, (4096, K), "wt"),
              _chk(bias, F32, (64,), "bias"), _chk(skip, BF16, None, "skip"), out.data_ptr(), B, Hs, Ws, K, _stream())
    return out


# ---- training-step loss ----
def l1_loss_partial(a, b, nblocks=2048):
    n = a.numel()
    part = torch.empty((nblocks,), dtype=F32, device=a.device)
    _lib.call("tup_l1_loss_partial", _chk(a, F32, None, "a"), _chk(b, F32, tuple(a.shape), "b"), part.data_ptr(), n, nblocks, _stream())
    return part


def l1_loss_bwd(a, b, gout):
    ga = torch.empty_like(a)
    _lib.call("tup_l1_loss_bwd", _chk(a, F32, None, "a"), _chk(b, F32, tuple(a.shape), "b"), _chk(gout, F32, None, "gout"),
              ga.data_ptr(), a.numel(), _stream())
    return ga


def relpos_bias_expand_n_h(table, heads):
    frag = torch.empty((heads, 4, 4, 64, 4), dtype=F32, device=table.device)
    _lib.call("tup_relpos_bias_expand_n_h", _chk(table, F32, (225, heads), "table"), frag.data_ptr(), heads, _stream())
    return frag


def window_attn_bwd_h(qkv, gout, att, lse, bias_n, heads, drop_p=0.0, drop_seed=0):
    """returns (gqkv bf16 [M][48*heads], dtable fp32 [225][heads])."""
    M = qkv.shape[0]
    assert M % 64 == 0
    gqkv = torch.empty((M, 48 * heads), dtype=BF16, device=qkv.device)
    dbias = torch.empty((heads, 4, 4, 64, 4), dtype=F32, device=qkv.device)
    scratch = torch.empty(_attn_bwd_scratch(M // 64, heads), dtype=F32, device=qkv.device)
    _lib.call("tup_window_attn_bwd_h", _chk(qkv, BF16, (M, 48 * heads), "qkv"), _chk(gout, BF16, (M, 16 * heads), "gout"),
              _chk(att, BF16, (M, 16 * heads), "att"), _chk(lse, F32, (M // 64, heads, 64), "lse"),
              _chk(bias_n, F32, (heads, 4, 4, 64, 4), "bias_n"),
              gqkv.data_ptr(), dbias.data_ptr(), scratch.data_ptr(), M // 64, heads, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _stream())
    dtable = torch.empty((225, heads), dtype=F32, device=qkv.device)
    _lib.call("tup_relpos_bias_reduce_h", dbias.data_ptr(), dtable.data_ptr(), heads, _stream())
    return gqkv, dtable


def wt_patch_wgrad(p, fmap):
    """fp32 [NI][4096] = p^T patches(fmap); p fp32 window-layout tokens [M][NI] over the floor(H/8) x floor(W/8) grid."""
    B, H, W, C = fmap.shape
    NI = p.shape[1]
    out = _zeros((NI, 4096), p.device)
    _lib.call("tup_wt_patch_wgrad", _chk(p, F32, None, "p"), _chk(fmap, BF16, None, "map"), out.data_ptr(), B, H, W, NI, _stream())
    return out


def wt_patch_unembed_bwd(gmap, wd):
    """d tokens (window layout, fp32 [M][N]) of the WindowTransformer patch_unembed: the patch_embed GEMM with W^T, no bias."""
    B, H, W, C = gmap.shape
    N = wd.shape[0]
    nwy, nwx = (H // 8 + 7) // 8, (W // 8 + 7) // 8
    x = torch.empty((B * nwy * nwx * 64, N), dtype=F32, device=gmap.device)
    _lib.call("tup_wt_patch_embed_fwd", _chk(gmap, BF16, None, "gmap"), _chk(wd, BF16, (N, 4096), "wd"), None,
              x.data_ptr(), B, H, W, N, _stream())
    return x


def wt_patch_embed_bwd(gx, wd, add):
    """d feat_down on the token-covered map = add + scatter(gx wd^T) (the patch_unembed GEMM with W^T, no bias)."""
    B, Hs, Ws, C = add.shape
    K = wd.shape[1]
    out = torch.empty_like(add)
    _lib.call("tup_wt_patch_unembed_fwd", _chk(gx, F32, None, "gx"), _chk(wd, BF16, (4096, K), "wd"), None,
              _chk(add, BF16, None, "add"), out.data_ptr(), B, Hs, Ws, K, _stream())
    return out
