"""CPU oracle for the FastTransformer SISR forward/backward path.

TEST INFRASTRUCTURE ONLY.  This file is a pure-PyTorch (CPU, fp32) restatement of the
reference algorithm; it is imported only by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.  The product path (``transformerupscaler_amd``)
never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference
module (``/root/reference/models/FastTransformer/model.py``, with a 6-line stub for
the absent ``torchvision.transforms.Resize``) in the build container, runs it on the
deterministic weights of ``transformerupscaler_amd.weights`` and stores inputs,
outputs and gradients under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
this restatement against those fixtures.  The reference itself has no tests or golden
vectors of its own.

The arithmetic itself lives in third-party PyTorch aten ops (not vendored by the
reference); every function below cites the reference call site it restates.
All functions take a plain ``state_dict`` (name -> tensor) with the reference's keys
(SURVEY.md §8(b)).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

WINDOW = 8          # models/FastTransformer/model.py:198
PATCH = 8           # models/FastTransformer/model.py:215 (kernel_size=8, stride=8)
NUM_BLOCKS = 6      # models/FastTransformer/model.py:194
NUM_HEADS = 12      # models/FastTransformer/model.py:195
VALID_SCALES = (2, 3, 4, 6)   # models/FastTransformer/utils.py:49


# --------------------------------------------------------------------------------------
# scale logic -- models/FastTransformer/model.py:245-248
# --------------------------------------------------------------------------------------
def resolve_scale(h: int, w: int, res_out: Tuple[int, int], upscale_factor: Optional[int]):
    if upscale_factor is not None:
        res_out = (h * upscale_factor, w * upscale_factor)
    else:
        upscale_factor = math.ceil(max(res_out[0] / h, res_out[1] / w))
    if upscale_factor not in VALID_SCALES:
        # models/FastTransformer/utils.py:96-97
        raise ValueError(f"Requested scale={upscale_factor} was not built.")
    return tuple(res_out), upscale_factor


# --------------------------------------------------------------------------------------
# Upsampler -- models/FastTransformer/utils.py:54-98
# --------------------------------------------------------------------------------------
def upsampler(sd: Dict[str, Tensor], prefix: str, x: Tensor, scale: int) -> Tensor:
    """conv(+bias) -> PixelShuffle stacks; keys ``{prefix}.upsamplers.{scale}.{idx}``."""
    if scale == 2 or scale == 4:
        steps = int(math.log2(scale))
        for s in range(steps):
            k = f"{prefix}.upsamplers.{scale}.{2 * s}"
            x = F.conv2d(x, sd[k + ".weight"], sd[k + ".bias"], padding=1)
            x = F.pixel_shuffle(x, 2)
        return x
    if scale in (3, 6):
        k = f"{prefix}.upsamplers.{scale}.0"
        x = F.conv2d(x, sd[k + ".weight"], sd[k + ".bias"], padding=1)
        return F.pixel_shuffle(x, scale)
    raise ValueError(f"Requested scale={scale} was not built.")


# --------------------------------------------------------------------------------------
# windows -- models/FastTransformer/model.py:31-63
# --------------------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int) -> Tensor:
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, -1, ws * ws, c)


def window_reverse(windows: Tensor, ws: int, h: int, w: int) -> Tensor:
    b = windows.shape[0]
    x = windows.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def relative_position_index(ws: int = WINDOW) -> Tensor:
    """models/FastTransformer/model.py:89-100; idx[i,j]=(yi-yj+ws-1)*(2ws-1)+(xi-xj+ws-1)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


# --------------------------------------------------------------------------------------
# WindowAttention.forward -- models/FastTransformer/model.py:104-133
# --------------------------------------------------------------------------------------
def window_attention(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int = NUM_HEADS,
                     ws: int = WINDOW, capture: Optional[dict] = None) -> Tensor:
    b, n, c = x.shape
    hd = c // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    if capture is not None:
        capture["qkv"] = qkv
    qkv = qkv.view(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (hd ** -0.5)
    attn = torch.matmul(q, k.transpose(-2, -1))
    idx = sd.get(p + ".relative_position_index")
    if idx is None:
        idx = relative_position_index(ws)
    bias = sd[p + ".relative_position_bias_table"][idx.view(-1)]
    bias = bias.view(ws * ws, ws * ws, -1).permute(2, 0, 1).unsqueeze(0)
    attn = (attn + bias).softmax(dim=-1)
    # attn_drop / proj_drop (model.py:127,132): identity in eval mode; the oracle is eval-mode.
    out = torch.matmul(attn, v).transpose(1, 2).reshape(b, n, c)
    if capture is not None:
        capture["attn_out"] = out
    return F.linear(out, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


# --------------------------------------------------------------------------------------
# WindowTransformerBlock.forward -- models/FastTransformer/model.py:153-172
# --------------------------------------------------------------------------------------
def window_block(sd: Dict[str, Tensor], p: str, x: Tensor, capture: Optional[dict] = None) -> Tensor:
    c = x.shape[-1]
    y = F.layer_norm(x, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    x = x + window_attention(sd, p + ".attn", y, capture=capture)
    y = F.layer_norm(x, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    y = F.linear(y, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"])
    y = F.gelu(y)  # nn.GELU() default = exact erf form (model.py:148)
    y = F.linear(y, sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"])
    return x + y


# --------------------------------------------------------------------------------------
# antialiased bilinear resize (what torchvision.transforms.Resize does on tensors;
# models/FastTransformer/model.py:323-325 and train.py:127-130)
# --------------------------------------------------------------------------------------
def aa_resize(x: Tensor, size: Tuple[int, int]) -> Tensor:
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=True)


def aa_bilinear_taps(in_size: int, out_size: int):
    """Explicit tap table of aten's antialiased bilinear (triangle) filter, one axis.

    Restates aten/src/ATen/native/cpu/UpSampleKernel.cpp ``_compute_indices_weights_aa``
    (PyTorch is a third-party dependency of the reference, requirements.txt:42; pinned
    torch~=2.6, run here on torch 2.10).  Returns (xmin[int32 out], xsize[int32 out],
    weights[float32 out x kmax]).  The GPU resize kernel consumes the same table
    (built by ``transformerupscaler_amd.resize_taps`` with its own code); this copy
    is the checker.
    """
    f = np.float32   # aten evaluates the table in the tensor dtype (float32): mimic it exactly
    scale = f(in_size) / f(out_size)
    support = f(scale) if scale >= 1.0 else f(1.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    kmax = int(math.ceil(float(support))) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    wts = np.zeros((out_size, kmax), np.float32)
    for i in range(out_size):
        center = scale * f(i + 0.5)
        lo = max(0, int(f(center - support + f(0.5))))
        hi = min(in_size, int(f(center + support + f(0.5))))
        n = hi - lo
        w = np.zeros(kmax, np.float32)
        tot = f(0.0)
        for j in range(n):
            t = abs(f(f(j + lo) - center + f(0.5)) * invscale)
            w[j] = max(f(0.0), f(1.0) - t)
            tot = f(tot + w[j])
        if tot != 0:
            w = (w / tot).astype(np.float32)
        xmin[i], xsize[i] = lo, n
        wts[i] = w
    return xmin, xsize, wts


def aa_resize_explicit(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """Separable resize from the explicit tap tables (H pass then W pass), fp32."""
    b, c, h, w = x.shape
    oh, ow = size
    ymin, ysz, yw = aa_bilinear_taps(h, oh)
    xmin, xsz, xw = aa_bilinear_taps(w, ow)
    mh = torch.zeros(oh, h, dtype=torch.float64)
    for i in range(oh):
        mh[i, ymin[i]:ymin[i] + ysz[i]] = torch.from_numpy(yw[i, :ysz[i]].astype(np.float64))
    mw = torch.zeros(ow, w, dtype=torch.float64)
    for i in range(ow):
        mw[i, xmin[i]:xmin[i] + xsz[i]] = torch.from_numpy(xw[i, :xsz[i]].astype(np.float64))
    y = torch.einsum("oh,bchw->bcow", mh, x.double())
    y = torch.einsum("pw,bcow->bcop", mw, y)
    return y.to(x.dtype)


# --------------------------------------------------------------------------------------
# TransformerModel.forward -- models/FastTransformer/model.py:231-327
# --------------------------------------------------------------------------------------
def forward(sd: Dict[str, Tensor], x: Tensor, res_out: Tuple[int, int] = (1080, 1920),
            upscale_factor: Optional[int] = None, require_ratio: bool = True,
            capture: Optional[dict] = None, clamp: bool = True, masks: Optional[dict] = None) -> Tensor:
    """Eval-mode forward.  ``capture`` (dict) receives named intermediates for per-kernel tests.

    ``masks`` (gradient tests only): {"feat1", "feat", "upscaled_input", "dec", "clamp"} -> 0/1 tensors that REPLACE the four
    ReLU gates and the clamp gate (``relu(z)`` becomes ``z * mask``, ``clamp(z)`` becomes ``z * mask``).  Evaluated at the gates
    another forward took (the bf16 HIP path), the autograd of this graph is the gradient that path should produce, with the
    ReLU / clamp decisions factored out -- what remains is kernel arithmetic error."""
    def gate(z, key):
        return F.relu(z) if masks is None else z * masks[key].to(z.dtype)
    cap = capture if capture is not None else {}
    res_out, s = resolve_scale(x.shape[2], x.shape[3], res_out, upscale_factor)

    # encoder, model.py:251-252 (shared in-place ReLU)
    feat = gate(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1), "feat1")
    cap["feat1"] = feat
    feat = gate(F.conv2d(feat, sd["conv2.weight"], sd["conv2.bias"], padding=1), "feat")
    cap["feat"] = feat
    b, c, hf, wf = feat.shape

    # reflect pad bottom/right to multiples of 8, model.py:256-261
    ph, pw = (PATCH - hf % PATCH) % PATCH, (PATCH - wf % PATCH) % PATCH
    feat_pad = F.pad(feat, (0, pw, 0, ph), mode="reflect") if (ph or pw) else feat

    # branch A, model.py:264-265 ; BasicConv = conv(no bias)+ReLU, utils.py:32-40
    up = upsampler(sd, "up1", feat, s)
    cap["up1"] = up
    upscaled_input = gate(F.conv2d(up, sd["up1_conv.conv.weight"], None, padding=1), "upscaled_input")
    cap["upscaled_input"] = upscaled_input

    # patch embed, model.py:268-270
    tokens = F.conv2d(feat_pad, sd["patch_embed.weight"], sd["patch_embed.bias"], stride=PATCH)
    ht, wt = tokens.shape[2], tokens.shape[3]
    tokens = tokens.permute(0, 2, 3, 1).contiguous()
    cap["tokens"] = tokens

    # zero-pad token grid to multiples of the window (NOT masked), model.py:273-280
    pb, pr = (WINDOW - ht % WINDOW) % WINDOW, (WINDOW - wt % WINDOW) % WINDOW
    oh_t, ow_t = ht, wt
    if pb or pr:
        tokens = F.pad(tokens.permute(0, 3, 1, 2), (0, pr, 0, pb)).permute(0, 2, 3, 1).contiguous()
        ht, wt = tokens.shape[1], tokens.shape[2]

    # windows + blocks, model.py:283-289
    win = window_partition(tokens, WINDOW)
    bw, nw, n, d = win.shape
    win = win.view(bw * nw, n, d)
    cap["win_in"] = win
    for i in range(NUM_BLOCKS):
        blk_cap = {} if capture is not None else None
        win = window_block(sd, f"window_blocks.{i}", win, capture=blk_cap)
        if capture is not None:
            cap[f"block{i}"] = win
            cap[f"block{i}_qkv"] = blk_cap["qkv"]
            cap[f"block{i}_attn_out"] = blk_cap["attn_out"]

    # reverse + crop + NCHW, model.py:292-299
    tokens = window_reverse(win.view(bw, nw, n, d), WINDOW, ht, wt)
    if pb or pr:
        tokens = tokens[:, :oh_t, :ow_t, :]
    tokens = tokens.permute(0, 3, 1, 2).contiguous()

    # patch unembed + crop + skip, model.py:302-309
    feat_trans = F.conv_transpose2d(tokens, sd["patch_unembed.weight"], sd["patch_unembed.bias"], stride=PATCH)
    feat_trans = feat_trans[:, :, :hf, :wf]
    combined = feat + feat_trans
    cap["combined"] = combined

    # decoder, model.py:312-313
    dec = gate(F.conv2d(combined, sd["decoder_conv1.weight"], sd["decoder_conv1.bias"], padding=1), "dec")
    cap["dec"] = dec
    residual = F.conv2d(dec, sd["decoder_conv2.weight"], sd["decoder_conv2.bias"], padding=1)
    cap["residual"] = residual

    # final upscale, model.py:316-317
    residual_up = upsampler(sd, "final_upscale", residual, s)
    residual_up = F.conv2d(residual_up, sd["final_upscale_conv.weight"], sd["final_upscale_conv.bias"], padding=1)
    cap["residual_up"] = residual_up

    out = upscaled_input + residual_up  # model.py:320
    cap["sum"] = out

    # model.py:323-325 -- note the (H, H) comparison quirk is reproduced as written
    if require_ratio and tuple(res_out) != (out.shape[2], out.shape[2]):
        out = aa_resize(out, res_out)
    cap["pre_clamp"] = out
    if masks is not None and clamp:
        return out * masks["clamp"].to(out.dtype)
    return torch.clamp(out, 0.0, 1.0) if clamp else out  # model.py:327


# --------------------------------------------------------------------------------------
# train step -- train.py:113-140 (per-sample forward, resize-to-target, L1, mean, Adam)
# --------------------------------------------------------------------------------------
def train_loss(sd: Dict[str, Tensor], lr_batch: Tensor, hr_batch: Tensor) -> Tensor:
    """Loss of one train.py step with dropout off (eval graph): mean over samples of L1 means."""
    losses = []
    for i in range(lr_batch.shape[0]):
        lr, hr = lr_batch[i:i + 1], hr_batch[i:i + 1]
        out = forward(sd, lr, res_out=(hr.shape[2], hr.shape[3]), require_ratio=False)
        if tuple(out.shape[2:]) != tuple(hr.shape[2:]):
            out = aa_resize(out, tuple(hr.shape[2:]))       # train.py:127-130
        losses.append(F.l1_loss(out, hr))                   # train.py:132
    return sum(losses) / len(losses)                        # train.py:136


def train_step_grads(sd: Dict[str, Tensor], lr_batch: Tensor, hr_batch: Tensor):
    """Returns (loss, {name: grad or None}) using autograd, as train.py:138 does."""
    leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v)
            for k, v in sd.items()}
    loss = train_loss(leaf, lr_batch, hr_batch)
    loss.backward()
    grads = {k: (v.grad if v.is_floating_point() else None) for k, v in leaf.items()}
    return loss.detach(), grads


def adam_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-4,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam defaults (train.py:104): no weight decay, no amsgrad."""
    m = b1 * m + (1 - b1) * grad
    v = b2 * v + (1 - b2) * grad * grad
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return param - lr * mhat / (vhat.sqrt() + eps), m, v
