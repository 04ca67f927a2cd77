"""CPU check of packing.pack_stream_block, the operand layout of the streamed whole-block kernel (csrc/block_stream.hip,
tup_blocks_stream_fwd): the packed tensors are unpacked again by the index maps the KERNEL uses (LDS image swizzles, tile rows, the
K order behind a LayerNorm and behind an accumulator, the accumulator-tile form of the relative position bias, the bias tables and
bias K-step words) and the block is evaluated from them in fp32 -- against the reference block (model.py:153-172) in plain torch.
What differs is only the bf16 / fp16 rounding of the packed weights.  No GPU, no library."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from transformerupscaler_amd import packing as P


def _rho(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def _unswizzle192(img):
    """[n][3][32][8][8] LDS image -> [n][32][192]: chunk c of a 128-byte row sits at position c ^ ((row >> 1) & 7)."""
    n = img.shape[0]
    out = torch.empty((n, 32, 192), dtype=torch.float32)
    v = img.float().reshape(n, 3, 32, 8, 8)
    for row in range(32):
        for c in range(8):
            out[:, row].view(n, 3, 8, 8)[:, :, c] = v[:, :, row, c ^ ((row >> 1) & 7)]
    return out


def _unswizzle32(img):
    """[n][32][32] image with 64-byte rows -> logical [n][32][32]: chunk c at position c ^ ((row >> 2) & 3)."""
    n = img.shape[0]
    v = img.float().reshape(n, 32, 4, 8)
    out = torch.empty_like(v)
    for row in range(32):
        for c in range(4):
            out[:, row, c] = v[:, row, c ^ ((row >> 2) & 3)]
    return out.reshape(n, 32, 32)


def _words_to_f32(words):
    """bias K-step words: bf16 hi | bf16 lo << 16 -> hi + lo"""
    w = words.view(torch.int32)
    hi = (w << 16).view(torch.float32)
    lo = (w & ~0xFFFF).view(torch.float32)
    return hi + lo


def _block_from_packed(x, packed):
    """One WindowTransformerBlock on x [nwin * 64][192], evaluated from the packed operands with the kernel's index maps."""
    wqk, wv, wproj, w1, w2, tab, sbias = packed
    och = P._stream_out_channel()            # tile row (rt, rho) -> residual channel
    kln = P._stream_k_after_ln()             # K column after a LayerNorm -> residual channel
    kacc192, kacc32 = P._stream_k_acc(192), P._stream_k_acc(32)
    Tqk, Tv, Tp, T1 = _unswizzle192(wqk), _unswizzle192(wv), _unswizzle192(wproj), _unswizzle192(w1)
    T2 = _unswizzle32(w2.reshape(24 * 6, 32, 32)).reshape(24, 6, 32, 32)
    qkb = tab[0:384].view(12, 2, 16); b1t = tab[384:1152].view(24, 2, 16)
    bp = _words_to_f32(tab[1152:1344]); b2 = _words_to_f32(tab[1344:1536])
    i16 = torch.arange(16)
    nwin = x.shape[0] // 64
    out = torch.empty_like(x)
    for w in range(nwin):
        xs = x[w * 64:(w + 1) * 64]                                   # [64 tokens][192]
        xh = F.layer_norm(xs, (192,))                                 # scale / shift are folded into the weights
        xk = xh[:, kln]                                               # tokens x packed K columns
        O = torch.zeros(64, 192)                                      # attention output, channel = 16 head + c
        for hd in range(12):
            # q | k tile: rows 0-15 q, 16-31 k; bias of tile row rho = table[h][i] with rho = rho(i, h)
            acc = Tqk[hd] @ xk.t()                                    # [32 rows][64 tokens]
            for h in range(2):
                acc[_rho(i16, h)] += qkb[hd, h][:, None]
            q, k = acc[:16], acc[16:]
            pair, half = hd // 2, hd % 2
            v = (Tv[pair] @ xk.t())[16 * half:16 * half + 16]         # [16 channels][64 tokens], no bias (folded into the proj's)
            # relative position bias as S^T accumulator tiles [query half][key half][lane][16]
            S = k.t() @ q                                             # [key][query]
            for hf in range(2):
                for kt in range(2):
                    for lane in range(64):
                        S[32 * kt + _rho(i16, lane >> 5), 32 * hf + (lane & 31)] += sbias[hd, hf, kt, lane]
            Pm = torch.exp2(S - S.max(dim=0, keepdim=True).values)    # log2 units: q and the bias carry log2(e)
            O[:, 16 * hd:16 * hd + 16] = ((v @ Pm) / Pm.sum(dim=0, keepdim=True)).t()
        # proj: tile rt row rho = out channel och[32 rt + rho]; K column kappa = in channel kacc192[kappa]
        x1 = xs.clone()
        Ok = O[:, kacc192]
        for rt in range(6):
            x1[:, och[32 * rt:32 * rt + 32]] += (Tp[rt] @ Ok.t()).t() + bp[32 * rt:32 * rt + 32][None, :]
        xh2 = F.layer_norm(x1, (192,))[:, kln]
        y = x1.clone()
        for c in range(24):
            hpre = T1[c] @ xh2.t()                                    # [32 hidden rows][64 tokens] = mlp.0 / 4
            for h in range(2):
                hpre[_rho(i16, h)] += b1t[c, h][:, None]
            hid = F.gelu(4.0 * hpre) / 4.0                            # the kernel's hidden tile: gelu(x) / 4 (fp16)
            hk = hid[kacc32]                                          # K order behind an accumulator
            for rt in range(6):
                y[:, och[32 * rt:32 * rt + 32]] += (T2[c, rt] @ hk).t()          # 4 W2 x gelu / 4
        for rt in range(6):
            y[:, och[32 * rt:32 * rt + 32]] += b2[32 * rt:32 * rt + 32][None, :]
        out[w * 64:(w + 1) * 64] = y
    return out


def _block_torch(raw, nwin):
    x = raw["x"]
    M = nwin * 64
    y = F.layer_norm(x, (192,), raw["gm1"], raw["bt1"], 1e-5)
    qkv = (y @ raw["w"].t() + raw["b"]).view(nwin, 64, 3, 12, 16).permute(2, 0, 3, 1, 4)
    bias = P.relpos_bias_matrix(raw["table"])
    attn = torch.softmax((qkv[0] * 0.25) @ qkv[1].transpose(-2, -1) + bias, dim=-1)
    att = (attn @ qkv[2]).transpose(1, 2).reshape(M, 192)
    x1 = x + att @ raw["wp"].t() + raw["bp"]
    y2 = F.layer_norm(x1, (192,), raw["gm2"], raw["bt2"], 1e-5)
    return x1 + F.linear(F.gelu(F.linear(y2, raw["w1"], raw["b1"])), raw["w2"], raw["b2"])


def test_stream_block_packing_reproduces_the_block_on_the_cpu():
    g = torch.Generator().manual_seed(23)
    nwin = 2
    raw = dict(x=torch.randn((nwin * 64, 192), generator=g),
               gm1=1 + 0.1 * torch.randn(192, generator=g), bt1=0.1 * torch.randn(192, generator=g),
               gm2=1 + 0.1 * torch.randn(192, generator=g), bt2=0.1 * torch.randn(192, generator=g),
               w=torch.randn((576, 192), generator=g) / 192 ** 0.5, b=0.1 * torch.randn(576, generator=g),
               wp=torch.randn((192, 192), generator=g) / 192 ** 0.5, bp=0.1 * torch.randn(192, generator=g),
               w1=torch.randn((768, 192), generator=g) * 0.08, b1=0.2 * torch.randn(768, generator=g),
               w2=torch.randn((192, 768), generator=g) * 0.05, b2=0.2 * torch.randn(192, generator=g),
               table=0.5 * torch.randn((225, 12), generator=g))
    packed = P.pack_stream_block(raw["w"], raw["b"], raw["gm1"], raw["bt1"], raw["table"], raw["wp"], raw["bp"],
                                 raw["w1"], raw["b1"], raw["gm2"], raw["bt2"], raw["w2"], raw["b2"])
    shapes = [tuple(t.shape) for t in packed]
    assert [t.numel() for t in packed] == [12 * 6144, 6 * 6144, 6 * 6144, 24 * 6144, 24 * 6144, 1536, 12 * 2 * 2 * 64 * 16], shapes
    assert packed[0].dtype == packed[3].dtype == torch.bfloat16 and packed[4].dtype == torch.float16
    got = _block_from_packed(raw["x"], packed)
    ref = _block_torch(raw, nwin)
    err = (got - ref).abs().max().item()
    print(f"block from the packed operands vs the reference block: max |diff| {err:.3e} (|ref| max {ref.abs().max().item():.2f})")
    # bf16 weights (2^-9 relative) through K = 192 / 768 products: measured 1.5e-2; a wrong index map is O(1)
    assert err <= 5e-2, err
    assert (got - raw["x"]).abs().max().item() > 0.1


def test_stream_index_maps_are_permutations():
    assert sorted(P._stream_out_channel().tolist()) == list(range(192))
    assert sorted(P._stream_k_after_ln().tolist()) == list(range(192))
    assert sorted(P._stream_k_acc(768).tolist()) == list(range(768))
    # a lane's 16 registers of a residual tile are 16 consecutive channels (the 64-byte pieces of the kernel's x loads / stores)
    och = P._stream_out_channel().view(6, 32)
    for h in range(2):
        rows = [_rho(i, h) for i in range(16)]
        ch = och[3][rows].tolist()
        assert ch == list(range(96 + 16 * h, 96 + 16 * h + 16)), ch


def test_stream_gelu_polynomial_from_the_kernel_source():
    """The GELU of tup_blocks_stream_fwd (csrc/block_stream.hip gelu_op; reference nn.GELU(), model.py:100-104): the coefficients are read
    out of the kernel source and the eight packed-fp16 instructions per pair of values are replayed in numpy with one fp16 rounding
    per fma -- error bound over |x| <= 12, exact saturation outside (the clamp bit of the last fma is the only clamp), no NaN from
    the overflow of the Horner chain."""
    import os
    import re
    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(__file__), "..", "transformerupscaler_amd", "csrc", "block_stream.hip")).read()
    body = src[src.index("TUP_DEVICE void gelu_op("):src.index("template <int K> TUP_DEVICE void gelu_pin(")]
    c_tail = [float(v) for v in re.search(r"constexpr float C\[3\] = \{([^}]*)\}", body).group(1).replace("f", "").split(",")]
    lead = [float(v) for v in re.findall(r"\(_Float16\)(-?[0-9.]+)f", body[body.index("t == 1"):body.index("t < 5")])]
    assert len(c_tail) == 3 and len(lead) == 4 and lead[0] == lead[1] and lead[2] == lead[3]
    coef = [lead[0], lead[2]] + c_tail                                     # highest degree first
    f16 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)
    fma = lambda a, b, c: f16(np.float32(a) * np.float32(b) + np.float32(c))
    x = np.concatenate([np.linspace(-12, 12, 96001), [-1e4, -300.0, -64.0, 64.0, 300.0, 1e4]]).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        u = f16(x / 4)
        s = fma(u, u, -0.5)
        q = fma(s, f16(coef[0]), f16(coef[1]))
        for c in coef[2:]:
            q = fma(q, s, f16(c))
        phi = np.clip(fma(u, q, 0.5), 0.0, 1.0)                              # v_pk_fma_f16 ... clamp
        got = f16(u * phi) * 4
    assert np.isfinite(got).all()
    want = x * 0.5 * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    err = np.abs(got - want)
    inner = np.abs(x) <= 12
    print("stream GELU: max |error| %.2e on |x| <= 12" % err[inner].max())
    assert err[inner].max() <= 5e-3
    assert (phi[x >= 4] == 1).all() and (phi[x <= -4] == 0).all()             # exact saturation, overflow included
    assert (got[~inner & (x < 0)] == 0).all() and np.allclose(got[~inner & (x > 0)], f16(f16(x[~inner & (x > 0)] / 4)) * 4)
