"""Host-side weight packing into the layouts the HIP kernels consume (include/tupscale_hip.h).

Pure index shuffles + bf16 casts with torch ops on whatever device the parameters live on
(plumbing; a few MB per call).  The reference state_dict layout is the source of truth.
"""
from __future__ import annotations

from typing import Dict

import torch

from .weights import BLOCKS, upsampler_layout

# row n_local = ct*16 + 4g + e of every 64-row weight group holds output feature g*16 + ct*4 + e,
# so that an MFMA lane (group g) ends up with 16 consecutive features (see csrc/common.h).
_PERM64 = torch.tensor([((n >> 2) & 3) * 16 + (n >> 4) * 4 + (n & 3) for n in range(64)], dtype=torch.long)


_IDX_CACHE: Dict[tuple, torch.Tensor] = {}


def _dev_index(key, build, device) -> torch.Tensor:
    """Index tensors are built once per device: the training loop repacks every weight after every optimizer step, and
    a host-built index means one host-to-device copy per packing call."""
    k = (key, str(device))
    t = _IDX_CACHE.get(k)
    if t is None:
        t = _IDX_CACHE[k] = build().to(device)
    return t


def _perm64(device) -> torch.Tensor:
    return _dev_index("perm64", lambda: _PERM64, device)


def perm_rows64(w: torch.Tensor) -> torch.Tensor:
    n = w.shape[0]
    assert n % 64 == 0
    idx = _dev_index(("rows64", n), lambda: (torch.arange(n // 64).view(-1, 1) * 64 + _PERM64.view(1, -1)).reshape(-1), w.device)
    return w.index_select(0, idx)


def pack_conv_c64(weight: torch.Tensor, bias, r: int):
    """Conv2d(64, 64*r*r, 3) (+PixelShuffle r) -> bf16 [r*r][9][64][64], bias fp32 [r*r][64]."""
    cout, cin = weight.shape[0], weight.shape[1]
    assert cin == 64 and cout == 64 * r * r and weight.shape[2:] == (3, 3)
    w = weight.reshape(64, r * r, 64, 9)                 # [c][sp][cin][tap]  (cout = c*r*r + sp)
    w = w.permute(1, 3, 0, 2)                            # [sp][tap][c][cin]
    w = w.index_select(2, _perm64(w.device))          # row n_local <- channel perm[n_local]
    b = None if bias is None else bias.reshape(64, r * r).t().contiguous().float()
    return w.unsqueeze(1).contiguous().to(torch.bfloat16), b       # [ntile][in-chunk = 1][9][64][64]


def pack_conv_c64_thin(weight: torch.Tensor):
    """Conv2d(64, co<=16, 3) -> bf16 [1][9][16][64] (rows >= co zero)."""
    co = weight.shape[0]
    assert weight.shape[1] == 64 and co <= 16
    w = torch.zeros(9, 16, 64, dtype=weight.dtype, device=weight.device)
    w[:, :co, :] = weight.reshape(co, 64, 9).permute(2, 0, 1)
    return w.view(1, 1, 9, 16, 64).contiguous().to(torch.bfloat16)


def pack_conv1(weight: torch.Tensor):
    """Conv2d(3, 64, 3) -> bf16 [64][32], k = tap*3 + cin, rows permuted."""
    assert tuple(weight.shape) == (64, 3, 3, 3)
    w = torch.zeros(64, 32, dtype=weight.dtype, device=weight.device)
    w[:, :27] = weight.permute(0, 2, 3, 1).reshape(64, 27)
    return perm_rows64(w).contiguous().to(torch.bfloat16)


def pack_planar(weight: torch.Tensor):
    """Conv2d(3, cout, 3) -> fp32 [cout][28] ((cin, ky, kx) order + one pad)."""
    cout = weight.shape[0]
    assert weight.shape[1:] == (3, 3, 3)
    w = torch.zeros(cout, 28, dtype=torch.float32, device=weight.device)
    w[:, :27] = weight.reshape(cout, 27).float()
    return w.contiguous()


def pack_planar_t(weight: torch.Tensor):
    """Conv2d(3, cout, 3) -> fp32 [27][cout padded to a multiple of 4], k = cin*9 + ky*3 + kx: the tap-major order in which the
    streaming output tail (csrc/tail_stream.hip) reads its wave-uniform weights (one scalar load per tap covers all outputs)."""
    cout = weight.shape[0]
    assert weight.shape[1:] == (3, 3, 3)
    w = torch.zeros(27, (cout + 3) // 4 * 4, dtype=torch.float32, device=weight.device)
    w[:, :cout] = weight.reshape(cout, 27).float().t()
    return w.contiguous()


def pack_linear(weight: torch.Tensor):
    """nn.Linear weight [N][K] -> bf16 with rows permuted per 64-group."""
    return perm_rows64(weight).contiguous().to(torch.bfloat16)


def pack_linear_stack(weights, transpose: bool = False):
    """pack_linear of several same-shaped nn.Linear weights with three launches in all (stack, row gather, bf16 cast)
    instead of two or three per weight: the training loop repacks every weight after every optimizer step, and ~150 tiny
    launches per step were a measurable slice of it.  transpose=True packs W^T (the input-gradient GEMM's operand).
    Returns one bf16 [n][rows][cols] tensor; entry i is contiguous."""
    w = torch.stack([t.detach() for t in weights])                     # [n][N][K]
    if transpose:
        w = w.transpose(1, 2)                                          # view; the gather below writes it out contiguous
    n = w.shape[1]
    assert n % 64 == 0
    idx = _dev_index(("rows64", n), lambda: (torch.arange(n // 64).view(-1, 1) * 64 + _PERM64.view(1, -1)).reshape(-1), w.device)
    return w.index_select(1, idx).to(torch.bfloat16)


def pack_fc1_fused(weight: torch.Tensor):
    """mlp.0 weight for tup_fused_mlp_fwd: pack_linear plus a column permutation.  K-step st of the kernel's FC1, lane
    group g, element j contracts over channel 64*(st>>1) + 16g + 8*(st&1) + j -- the channels whose residual the same
    lane carries in its FC2 accumulators -- so packed column 32st + 8g + j holds that channel."""
    return pack_linear(weight).index_select(1, _fused_k_order(weight.device)).contiguous()


def pack_fc1_fused_q(weight: torch.Tensor, bias: torch.Tensor):
    """mlp.0 for the fused inference MLPs (tup_fused_mlp_fwd, tup_fused_block_fwd, tup_fused_blocks32_fwd): pack_fc1_fused of
    W1 / 4 and b1 / 4.  A power of two, so exact in bf16 / fp32: FC1's accumulators then hold x / 4, the range in which the GELU
    polynomial is evaluated in packed fp16 (csrc/common.h, gelu16_batch)."""
    return pack_fc1_fused(weight.detach() * 0.25), (bias.detach().float() * 0.25).contiguous()


def pack_fc2_h4(weight: torch.Tensor):
    """mlp.2 for the same kernels: 4 W2 in FP16 (rows permuted per 64-group as pack_linear): the hidden tile gelu(x) / 4 stays fp16
    and FC2 runs on v_mfma_f32_16x16x32_f16; fp16 carries three more mantissa bits than the bf16 tile it replaces."""
    return perm_rows64(weight.detach() * 4.0).contiguous().to(torch.float16)


def fold_layernorm(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """Linear(LayerNorm(x)) with the LayerNorm's scale and shift moved into the Linear (model.py:163 norm1 -> attn.qkv, :168 norm2 ->
    mlp.0): W (xhat * gamma + beta) + b = (W diag(gamma)) xhat + (b + W beta), formed in fp32.  The whole-block kernels
    (tup_fused_block_fwd, tup_fused_blocks32_fwd) then normalise with one FMA per value and never load gamma / beta."""
    w = weight.detach().float()
    return w * gamma.detach().float()[None, :], bias.detach().float() + w @ beta.detach().float()


def _fused_k_order(device):
    """Column order of the weights that follow a LayerNorm inside the fused block kernels: K-step st, lane group g, element j
    contracts over channel 64*(st>>1) + 16g + 8*(st&1) + j (the channels whose residual the same lane carries in its accumulators)."""
    def build():
        k = torch.arange(192)
        st, g, j = k // 32, (k % 32) // 8, k % 8
        return 64 * (st // 2) + 16 * g + 8 * (st % 2) + j
    return _dev_index("fc1_fused", build, device)


def pack_qkv_heads(weight: torch.Tensor, bias: torch.Tensor, heads: int = 12, natural_k: bool = False):
    """attn.qkv Linear [3*dim][dim] -> per head the 16 q, 16 k, 16 v weight rows + 16 zero rows: bf16 [heads][64][dim], and the
    matching biases fp32 [heads][48] (tup_fused_qkv_attn_fwd and the kernels built on it).  The K columns are in the order of
    pack_fc1_fused (LayerNorm1 reads the residual stream in the layout of the accumulators that produce it); natural_k=True
    keeps the channel order (dim = 192 only for the permuted form)."""
    dim = weight.shape[1]
    w = weight.detach().reshape(3, heads, 16, dim).permute(1, 0, 2, 3).reshape(heads, 48, dim)
    if not natural_k and dim == 192:
        w = w.index_select(2, _fused_k_order(weight.device))
    wh = torch.zeros(heads, 64, dim, dtype=weight.dtype, device=weight.device)
    wh[:, :48] = w
    bh = bias.detach().reshape(3, heads, 16).permute(1, 0, 2).reshape(heads, 48).float().contiguous()
    return wh.contiguous().to(torch.bfloat16), bh


def pack_proj_pairs(weight: torch.Tensor):
    """attn.proj weight [dim][dim] for tup_fused_attn_block_fwd: rows permuted per 64-group (as pack_linear) and, inside every
    32-column block (= two heads), columns reordered so that lane group g's eight K values are contiguous:
    new column 32p + 8g + j  <-  old 32p + 4g + j (j < 4, head 2p) | 32p + 16 + 4g + (j - 4) (j >= 4, head 2p + 1)."""
    dim = weight.shape[1]
    idx = torch.empty(dim, dtype=torch.long)
    for p in range(dim // 32):
        for g in range(4):
            for j in range(8):
                idx[32 * p + 8 * g + j] = 32 * p + 4 * g + j if j < 4 else 32 * p + 16 + 4 * g + (j - 4)
    return perm_rows64(weight.detach()[:, idx.to(weight.device)]).contiguous().to(torch.bfloat16)


# ---- the streamed whole-block kernel (csrc/block_stream.hip, tup_blocks_stream_fwd): 32x32x16 MFMA tiles ----
# A wave owns 32 tokens (lane r = l & 31, half h = l >> 5).  An accumulator tile [32 rows][32 tokens] keeps row
# rho(i, h) = (i & 3) + 8 (i >> 2) + 4 h in register i of a lane.  The residual stream uses six such tiles with tile row rho of tile rt
# = channel 32 rt + 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3), so that a lane's 16 registers of a tile are 16 consecutive channels
# (64-byte loads / stores of x) and pack pairwise into the B fragments of the product that follows a LayerNorm: K-step s = 2 rt + u,
# half h, element j  <->  channel 32 rt + 16 h + 8 u + j.
def _stream_rho(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def _stream_out_channel():
    """tile row (rt, rho) -> residual channel, [192]"""
    rho = torch.arange(32)
    ch = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3)
    return (torch.arange(6).view(6, 1) * 32 + ch.view(1, 32)).reshape(-1)


def _stream_k_after_ln():
    """packed K column kappa = 16 s + 8 h + j -> residual channel (the order in which LayerNorm's fragments hold the channels)"""
    k = torch.arange(192)
    s_, h, j = k // 16, (k % 16) // 8, k % 8
    return 32 * (s_ // 2) + 16 * h + 8 * (s_ % 2) + j


def _stream_k_acc(n):
    """packed K column kappa = 16 s + 8 h + j -> row 16 s + rho(j, h) of an accumulator tile that is the B operand (n columns)"""
    k = torch.arange(n)
    s_, h, j = k // 16, (k % 16) // 8, k % 8
    return 16 * s_ + (j & 3) + 8 * (j >> 2) + 4 * h


def _stream_image192(t: torch.Tensor) -> torch.Tensor:
    """[n][32 rows][192 K] (bf16 / fp16) -> the byte-exact LDS image [n][3 k-tiles][32][64]: 128-byte rows whose 16-byte chunk c sits at
    position c ^ ((row >> 1) & 7) (conflict-free ds_read_b128 of the 32x32x16 A fragments; the DMA is then a linear copy)."""
    n = t.shape[0]
    v = t.reshape(n, 32, 3, 8, 8).permute(0, 2, 1, 3, 4)                 # [n][kt][row][chunk][8]
    row = torch.arange(32).view(32, 1)
    pos = torch.arange(8).view(1, 8)
    src = (pos ^ ((row >> 1) & 7)).view(1, 1, 32, 8, 1).expand(n, 3, 32, 8, 8).to(t.device)
    return torch.gather(v, 3, src).contiguous()


def _stream_image32(t: torch.Tensor) -> torch.Tensor:
    """[n][32 rows][32 K] -> LDS image with 64-byte rows, chunk c at position c ^ ((row >> 2) & 3)."""
    n = t.shape[0]
    v = t.reshape(n, 32, 4, 8)
    row = torch.arange(32).view(32, 1)
    pos = torch.arange(4).view(1, 4)
    src = (pos ^ ((row >> 2) & 3)).view(1, 32, 4, 1).expand(n, 32, 4, 8).to(t.device)
    return torch.gather(v, 2, src).contiguous()


def _bf16_hi_lo_words(v: torch.Tensor) -> torch.Tensor:
    """fp32 -> int32 words  bf16(v) | bf16(v - bf16(v)) << 16  (the two K columns of a bias K-step: v to 2^-17 relative)."""
    hi = v.to(torch.bfloat16)
    lo = (v - hi.float()).to(torch.bfloat16)
    w = hi.view(torch.int16).to(torch.int32) & 0xFFFF | (lo.view(torch.int16).to(torch.int32) << 16)
    return w.contiguous()


def relpos_bias_matrix(table: torch.Tensor) -> torch.Tensor:
    """relative_position_bias_table [225][heads] -> bias [heads][64 queries][64 keys] (model.py:89-100,120-124) for 8 x 8 windows."""
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    idx = ((ys[:, None] - ys[None, :] + 7) * 15 + (xs[:, None] - xs[None, :] + 7)).to(table.device)
    return table.detach().float()[idx.view(-1)].view(64, 64, -1).permute(2, 0, 1).contiguous()


def pack_stream_block(wqkv, bqkv, gamma1, beta1, table, wproj, bproj, w1, b1, gamma2, beta2, w2, b2):
    """One WindowTransformerBlock (model.py:153-172; heads 12 x 16, dim 192, hidden 768) for tup_blocks_stream_fwd.  Returns the seven
    tensors of a table row: wqk bf16 [12][3][32][64] (per head a tile of 16 q rows, pre-scaled by head_dim^-0.5 log2(e), model.py:117,
    and 16 k rows), wv bf16 [6][3][32][64] (per head pair 16 + 16 v rows), wproj bf16 [6][3][32][64], w1 bf16 [24][3][32][64]
    (mlp.0 / 4 in chunks of 32 hidden units), w2 fp16 [24][6][32][32] (4 mlp.2), tab fp32 [1536] (qk biases [12][2][16], mlp.0 biases
    [24][2][16], bias K-step words of proj [6][32] and mlp.2 [6][32]), sbias fp32 [12][2][2][64][16] (relative position bias as the
    accumulator input of S^T).  LayerNorm scale / shift are folded into attn.qkv / mlp.0 (fold_layernorm); the v bias goes through the
    softmax (rows of P sum to 1) and the proj, so it is folded into the proj bias."""
    dev = wqkv.device
    wq, bq = fold_layernorm(wqkv, bqkv, gamma1, beta1)                  # fp32 [576][192], [576]
    w1f, b1f = fold_layernorm(w1, b1, gamma2, beta2)
    kln = _stream_k_after_ln().to(dev)
    wq = wq[:, kln]
    # q carries head_dim^-0.5 (model.py:117) AND log2(e): the scores come out in log2 units and the kernel's softmax is exp2(s - max)
    LOG2E = 1.4426950408889634
    q = wq[0:192].reshape(12, 16, 192) * (0.25 * LOG2E)
    k = wq[192:384].reshape(12, 16, 192)
    v = wq[384:576].reshape(6, 32, 192)
    wqk = _stream_image192(torch.cat([q, k], dim=1).to(torch.bfloat16))
    wv = _stream_image192(v.to(torch.bfloat16))
    och = _stream_out_channel().to(dev)
    wp_b = wproj.detach().float().to(torch.bfloat16)
    wp = wp_b[och][:, _stream_k_acc(192).to(dev)].reshape(6, 32, 192)
    wpi = _stream_image192(wp)
    w1c = (w1f[:, kln] * 0.25).reshape(24, 32, 192).to(torch.bfloat16)
    w1i = _stream_image192(w1c)
    w2p = (w2.detach().float() * 4.0)[och][:, _stream_k_acc(768).to(dev)]            # [192][768]: columns in chunks of 32
    w2c = w2p.reshape(6, 32, 24, 32).permute(2, 0, 1, 3).reshape(24 * 6, 32, 32).to(torch.float16)
    w2i = _stream_image32(w2c).reshape(24, 6, 32, 32)
    # tables
    i = torch.arange(16)
    rho = torch.stack([_stream_rho(i, 0), _stream_rho(i, 1)])                        # [2][16]
    qb = (bq[0:192].reshape(12, 16) * (0.25 * LOG2E))
    kb = bq[192:384].reshape(12, 16)
    qkb = torch.cat([qb, kb], dim=1)[:, rho.to(dev)]                                  # [12][2][16]
    b1t = (b1f * 0.25).reshape(24, 32)[:, rho.to(dev)]                                # [24][2][16]
    bp = bproj.detach().float() + wp_b.float() @ bq[384:576]                          # v bias through softmax and proj
    tab = torch.cat([qkb.reshape(-1), b1t.reshape(-1),
                     _bf16_hi_lo_words(bp[och]).view(torch.float32), _bf16_hi_lo_words(b2.detach().float()[och]).view(torch.float32)])
    assert tab.numel() == 384 + 768 + 192 + 192
    # relative position bias as S^T accumulator tiles: [head][query half][key half][lane][16]: key 32 KT + rho(i, l >> 5), query 32 hf + (l & 31)
    bias = relpos_bias_matrix(table) * LOG2E                                          # [12][q][k], log2 units
    lane = torch.arange(64)
    qi = (torch.arange(2).view(2, 1, 1, 1) * 32 + (lane & 31).view(1, 1, 64, 1)).expand(2, 2, 64, 16)
    ki = (torch.arange(2).view(1, 2, 1, 1) * 32 + _stream_rho(i.view(1, 1, 1, 16), (lane >> 5).view(1, 1, 64, 1))).expand(2, 2, 64, 16)
    sb = bias[:, qi.to(dev), ki.to(dev)].contiguous()                                 # [12][2][2][64][16]
    return (wqk.contiguous(), wv.contiguous(), wpi.contiguous(), w1i.contiguous(), w2i.contiguous(), tab.contiguous().float(), sb)


def pack_patch_embed(weight: torch.Tensor):
    """Conv2d(64,192,k8,s8) weight [192][64][8][8] -> bf16 [192][4096], k = (i*8+j)*64 + c."""
    return pack_linear(weight.permute(0, 2, 3, 1).reshape(192, 4096))


def pack_patch_unembed(weight: torch.Tensor):
    """ConvTranspose2d(192,64,k8,s8) weight [192][64][8][8] -> bf16 [4096][192], n = (i*8+j)*64 + o."""
    return pack_linear(weight.permute(2, 3, 1, 0).reshape(4096, 192))


# ---- composed branch A (inference): last Upsampler conv + PixelShuffle + up1_conv as one 5x5 conv ----
def compose_branch_a(wu: torch.Tensor, bu: torch.Tensor, w3: torch.Tensor, r: int, rowmode: int = 0, colmode: int = 0):
    """Exact composition (fp32) of Conv2d(64, 64*r*r, 3)+bias -> PixelShuffle(r) -> Conv2d(64, 3, 3, bias=False)
    (utils.py:62-63/74-75/83-84 followed by utils.py:32-40; no activation in between, utils.py:50).
    Returns (wc [3*r*r][5][5][64] with output n = c*r*r + si*r + sj and LR taps at offsets -2..2, bc [3*r*r]).
    rowmode/colmode 1 (2) drop the HR taps above (below) / left (right) of the output pixel: the variants
    for the first (last) HR row / column, where the reference zero-pads the HR intermediate."""
    # Composed on the host in fp64 (a one-time weight transform, ~10 ms): no device GEMM library is involved and the
    # composition error stays far below the bf16 rounding applied afterwards.
    dev = wu.device
    wu5 = wu.detach().double().cpu().reshape(64, r, r, 64, 3, 3)          # [ch][si'][sj'][ci][ky][kx]
    bu3 = bu.detach().double().cpu().reshape(64, r, r)
    w3 = w3.detach().double().cpu()
    wc = torch.zeros(3, r, r, 5, 5, 64, dtype=torch.float64)
    bc = torch.zeros(3, r, r, dtype=torch.float64)
    for si in range(r):
        for dy in range(3):
            if (rowmode == 1 and dy == 0) or (rowmode == 2 and dy == 2):
                continue
            oy, si2 = divmod(si + dy - 1, r)               # LR row offset and sub-row of the HR intermediate
            for sj in range(r):
                for dx in range(3):
                    if (colmode == 1 and dx == 0) or (colmode == 2 and dx == 2):
                        continue
                    ox, sj2 = divmod(sj + dx - 1, r)
                    m = (w3[:, :, dy, dx].reshape(3, 64, 1) * wu5[:, si2, sj2].reshape(1, 64, 64 * 9)).sum(1)   # [3][ci*ky*kx]
                    wc[:, si, sj, oy + 1:oy + 4, ox + 1:ox + 4, :] += m.reshape(3, 64, 3, 3).permute(0, 2, 3, 1)
                    bc[:, si, sj] += (w3[:, :, dy, dx] * bu3[:, si2, sj2].reshape(1, 64)).sum(1)
    wc = wc.float().to(dev)
    bc = bc.float().to(dev)
    return wc.reshape(3 * r * r, 5, 5, 64), bc.reshape(3 * r * r)


def pack_branch_a(wu, bu, w3, r: int):
    """-> (wp bf16 [1][1][25][rows][64], bias fp32 [3rr], wv bf16 [9][3rr][25][64], bv fp32 [9][3rr])."""
    rows = {2: 16, 3: 32, 6: 112}[r]
    n = 3 * r * r
    wc, bc = compose_branch_a(wu, bu, w3, r)
    wp = torch.zeros(25, rows, 64, dtype=torch.float32, device=wu.device)
    wp[:, :n, :] = wc.reshape(n, 25, 64).permute(1, 0, 2)
    wv, bv = [], []
    for rowmode in range(3):
        for colmode in range(3):
            a, b = compose_branch_a(wu, bu, w3, r, rowmode, colmode)
            wv.append(a.reshape(n, 25, 64)); bv.append(b)
    return (wp.view(1, 1, 25, rows, 64).contiguous().to(torch.bfloat16), bc.contiguous(),
            torch.stack(wv).contiguous().to(torch.bfloat16), torch.stack(bv).contiguous())


# ---- backward (input-gradient) packings: the same kernels run with transposed / flipped weights ----
def pack_conv_c64_dgrad(weight: torch.Tensor, r: int):
    """Conv2d(64, 64*r*r, 3) -> weights of its input-gradient conv (64*r*r -> 64, read through
    PixelShuffle^-1): bf16 [1][r*r][9][64 rows = cin][64 = c]."""
    w = weight.reshape(64, r * r, 64, 3, 3).flip(3, 4).reshape(64, r * r, 64, 9)    # [c][sp][cin][tap']
    w = w.permute(1, 3, 2, 0)                                                       # [sp][tap'][cin][c]
    w = w.index_select(2, _perm64(w.device))
    return w.unsqueeze(0).contiguous().to(torch.bfloat16)


def pack_conv_thin_dgrad(weight: torch.Tensor):
    """Conv2d(64, 3, 3) -> conv1-format weights (3 -> 64) of its input-gradient conv."""
    return pack_conv1(weight.flip(2, 3).transpose(0, 1).contiguous())


def pack_planar_dgrad(weight: torch.Tensor):
    """Conv2d(3, 3, 3) -> planar weights of its input-gradient conv."""
    return pack_planar(weight.flip(2, 3).transpose(0, 1).contiguous())


def unpack_conv_c64_wgrad(dwp: torch.Tensor, db, r: int):
    """[r*r][64 c][9][64 ci] (+ bias [r*r][64]) -> reference layout [64*r*r][64][3][3] (+ [64*r*r])."""
    dw = dwp.permute(1, 0, 3, 2).reshape(64 * r * r, 64, 3, 3)
    return dw, (None if db is None else db.t().reshape(-1))


def pack_state_dict(sd: Dict[str, torch.Tensor], scale: int, backward: bool = False) -> Dict[str, torch.Tensor]:
    """Everything one forward (and, with backward=True, one backward) at `scale` needs, keyed by short names."""
    pk: Dict[str, torch.Tensor] = {}
    f32 = lambda t: t.detach().float().contiguous()
    pk["conv1.w"] = pack_conv1(sd["conv1.weight"].detach()); pk["conv1.b"] = f32(sd["conv1.bias"])
    pk["conv2.w"], pk["conv2.b"] = pack_conv_c64(sd["conv2.weight"].detach(), sd["conv2.bias"].detach(), 1)
    for si, (idx, r) in enumerate(upsampler_layout(scale)):
        k = f"up1.upsamplers.{scale}.{idx}"
        pk[f"up1.{si}.w"], pk[f"up1.{si}.b"] = pack_conv_c64(sd[k + ".weight"].detach(), sd[k + ".bias"].detach(), r)
        k = f"final_upscale.upsamplers.{scale}.{idx}"
        pk[f"fu.{si}.w"] = pack_planar(sd[k + ".weight"].detach()); pk[f"fu.{si}.b"] = f32(sd[k + ".bias"])
    pk["up1_conv.w"] = pack_conv_c64_thin(sd["up1_conv.conv.weight"].detach())
    if not backward:      # inference: composed branch A (last up stage + up1_conv)
        idx, r = upsampler_layout(scale)[-1]
        k = f"up1.upsamplers.{scale}.{idx}"
        pk["bra.w"], pk["bra.b"], pk["bra.wv"], pk["bra.bv"] = pack_branch_a(
            sd[k + ".weight"].detach(), sd[k + ".bias"].detach(), sd["up1_conv.conv.weight"].detach(), r)
    pk["fuc.w"] = pack_planar(sd["final_upscale_conv.weight"].detach()); pk["fuc.b"] = f32(sd["final_upscale_conv.bias"])
    if not backward and upsampler_layout(scale)[-1][1] == 2:      # inference, last stage x2: the streaming output tail's tap-major weights
        li = len(upsampler_layout(scale)) - 1
        pk["tail.wfu_t"] = pack_planar_t(sd[f"final_upscale.upsamplers.{scale}.{upsampler_layout(scale)[-1][0]}.weight"].detach())
        pk["tail.wfc_t"] = pack_planar_t(sd["final_upscale_conv.weight"].detach())
    pk["pe.w"] = pack_patch_embed(sd["patch_embed.weight"].detach()); pk["pe.b"] = f32(sd["patch_embed.bias"])
    for i in range(BLOCKS):
        p = f"window_blocks.{i}"
        for nm in ("norm1", "norm2"):
            pk[f"b{i}.{nm}.w"] = f32(sd[f"{p}.{nm}.weight"]); pk[f"b{i}.{nm}.b"] = f32(sd[f"{p}.{nm}.bias"])
        pk[f"b{i}.table"] = f32(sd[f"{p}.attn.relative_position_bias_table"])
        for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            pk[f"b{i}.{nm}.b"] = f32(sd[f"{p}.{key}.bias"])
    for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
        ws = [sd[f"window_blocks.{i}.{key}.weight"] for i in range(BLOCKS)]
        wp = pack_linear_stack(ws)
        for i in range(BLOCKS):
            pk[f"b{i}.{nm}.w"] = wp[i]
        if backward:
            wt = pack_linear_stack(ws, transpose=True)
            for i in range(BLOCKS):
                pk[f"b{i}.{nm}.wd"] = wt[i]
    if not backward:      # inference fusion of norm1 + qkv + attention
        for i in range(BLOCKS):
            p = f"window_blocks.{i}"
            pk[f"b{i}.qkv.wh"], pk[f"b{i}.qkv.bh"] = pack_qkv_heads(sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"])
            pk[f"b{i}.proj.wpp"] = pack_proj_pairs(sd[f"{p}.attn.proj.weight"])
            pk[f"b{i}.fc1.wfq"], pk[f"b{i}.fc1.bq"] = pack_fc1_fused_q(sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"])
            # the whole-block kernels: norm1 / norm2 folded into the Linear that follows them
            pk[f"b{i}.qkv.whn"], pk[f"b{i}.qkv.bhn"] = pack_qkv_heads(*fold_layernorm(
                sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"], sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"]))
            pk[f"b{i}.fc1.wfqn"], pk[f"b{i}.fc1.bqn"] = pack_fc1_fused_q(*fold_layernorm(
                sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"], sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]))
            pk[f"b{i}.fc2.wh4"] = pack_fc2_h4(sd[f"window_blocks.{i}.mlp.2.weight"])
            # the streamed 32x32x16 whole-block kernel (csrc/block_stream.hip)
            for j, t in enumerate(pack_stream_block(
                    sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"], sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"],
                    sd[f"{p}.attn.relative_position_bias_table"], sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"],
                    sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"], sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"],
                    sd[f"{p}.mlp.2.weight"], sd[f"{p}.mlp.2.bias"])):
                pk[f"b{i}.stream.{j}"] = t
    pk["pu.w"] = pack_patch_unembed(sd["patch_unembed.weight"].detach()); pk["pu.b"] = f32(sd["patch_unembed.bias"])
    pk["dec1.w"], pk["dec1.b"] = pack_conv_c64(sd["decoder_conv1.weight"].detach(), sd["decoder_conv1.bias"].detach(), 1)
    pk["dec2.w"] = pack_conv_c64_thin(sd["decoder_conv2.weight"].detach()); pk["dec2.b"] = f32(sd["decoder_conv2.bias"])
    if not backward:
        return pk
    # ---- extra packings the backward needs ----
    t = lambda k: sd[k].detach()
    pk["conv1.wd"] = pack_conv_c64_thin(t("conv1.weight").flip(2, 3).transpose(0, 1).contiguous())    # d/dx: a 64 -> 3 conv
    pk["conv2.wd"] = pack_conv_c64_dgrad(t("conv2.weight"), 1)
    pk["dec1.wd"] = pack_conv_c64_dgrad(t("decoder_conv1.weight"), 1)
    pk["dec2.wd"] = pack_conv_thin_dgrad(t("decoder_conv2.weight"))
    pk["up1_conv.wd"] = pack_conv_thin_dgrad(t("up1_conv.conv.weight"))
    pk["fuc.wd"] = pack_planar_dgrad(t("final_upscale_conv.weight"))
    for si, (idx, r) in enumerate(upsampler_layout(scale)):
        pk[f"up1.{si}.wd"] = pack_conv_c64_dgrad(t(f"up1.upsamplers.{scale}.{idx}.weight"), r)
        pk[f"fu.{si}.raw"] = f32(sd[f"final_upscale.upsamplers.{scale}.{idx}.weight"])
    pk["pe.wd"] = pack_linear(t("patch_embed.weight").permute(2, 3, 1, 0).reshape(4096, 192))     # rows (i,j,c), cols n
    pk["pu.wd"] = pack_linear(t("patch_unembed.weight").permute(0, 2, 3, 1).reshape(192, 4096))   # rows k, cols (i,j,o)
    return pk


# ------------------------------------------------------------------------------------------------
# ResidualTransformer
# ------------------------------------------------------------------------------------------------
def pack_conv_c64_stride2(weight: torch.Tensor, bias):
    """Conv2d(64, 64, 3, stride=2, padding=1) as a 3x3 conv over the space-to-depth (2x2 -> 256 channel) input that
    tup_conv3x3_c64_fwd reads with in_r = 2: chunk (si, sj), block tap (dy, dx) <-> original tap
    (ky, kx) = (2(dy-1)+si+1, 2(dx-1)+sj+1) when that lies in 0..2, zero otherwise.  bf16 [1][4][9][64][64]."""
    assert tuple(weight.shape) == (64, 64, 3, 3)
    w = torch.zeros(4, 9, 64, 64, dtype=weight.dtype, device=weight.device)          # [chunk][tap][cout][cin]
    for si in range(2):
        for sj in range(2):
            for dy in range(3):
                for dx in range(3):
                    ky, kx = 2 * (dy - 1) + si + 1, 2 * (dx - 1) + sj + 1
                    if 0 <= ky <= 2 and 0 <= kx <= 2:
                        w[si * 2 + sj, dy * 3 + dx] = weight[:, :, ky, kx]
    w = w.index_select(2, _perm64(w.device))
    b = None if bias is None else bias.reshape(1, 64).contiguous().float()
    return w.unsqueeze(0).contiguous().to(torch.bfloat16), b


def pack_conv_c64_stride2_dgrad(weight: torch.Tensor):
    """Input-gradient conv of Conv2d(64, 64, 3, stride=2, padding=1): a 3x3 conv over the output gradient with one
    output tile per input sub-pixel (si, sj), stored through PixelShuffle(2).  Block tap (dy, dx) of sub-pixel (si, sj)
    carries original tap ky = si + 1 - 2(dy - 1) (same in x) when that lies in 0..2.  bf16 [4][1][9][64 = ci][64 = co]."""
    assert tuple(weight.shape) == (64, 64, 3, 3)
    w = torch.zeros(4, 9, 64, 64, dtype=weight.dtype, device=weight.device)          # [sub-pixel][tap][ci][co]
    for si in range(2):
        for sj in range(2):
            for dy in range(3):
                for dx in range(3):
                    ky, kx = si + 1 - 2 * (dy - 1), sj + 1 - 2 * (dx - 1)
                    if 0 <= ky <= 2 and 0 <= kx <= 2:
                        w[si * 2 + sj, dy * 3 + dx] = weight[:, :, ky, kx].t()
    w = w.index_select(2, _perm64(w.device))
    return w.unsqueeze(1).contiguous().to(torch.bfloat16)


def unpack_conv_c64_stride2_wgrad(dwp: torch.Tensor):
    """[4 planes][64 co][9 block taps][64 ci] -> reference layout [64][64][3][3] (inverse of pack_conv_c64_stride2)."""
    dw = torch.zeros(64, 64, 3, 3, dtype=dwp.dtype, device=dwp.device)
    for si in range(2):
        for sj in range(2):
            for dy in range(3):
                for dx in range(3):
                    ky, kx = 2 * (dy - 1) + si + 1, 2 * (dx - 1) + sj + 1
                    if 0 <= ky <= 2 and 0 <= kx <= 2:
                        dw[:, :, ky, kx] = dwp[si * 2 + sj, :, dy * 3 + dx, :]
    return dw


def pack_rt_state_dict(sd: Dict[str, torch.Tensor], backward: bool = False) -> Dict[str, torch.Tensor]:
    pk: Dict[str, torch.Tensor] = {}
    f32 = lambda t: t.detach().float().contiguous()
    t = lambda k: sd[k].detach()
    pk["conv1.w"] = pack_conv1(t("conv1.weight")); pk["conv1.b"] = f32(sd["conv1.bias"])
    pk["conv2.w"], pk["conv2.b"] = pack_conv_c64(t("conv2.weight"), t("conv2.bias"), 1)
    pk["ds.w"], pk["ds.b"] = pack_conv_c64_stride2(t("downsample.weight"), t("downsample.bias"))
    pk["pe.w"] = pack_linear(t("patch_embed.weight").permute(0, 2, 3, 1).reshape(128, 4096)); pk["pe.b"] = f32(sd["patch_embed.bias"])
    pk["pos"] = f32(sd["pos_embed"]).reshape(-1, 128).contiguous()
    nb = 0
    while f"transformer_blocks.{nb}.norm1.weight" in sd:
        p = f"transformer_blocks.{nb}"
        for nm in ("norm1", "norm2"):
            pk[f"b{nb}.{nm}.w"] = f32(sd[f"{p}.{nm}.weight"]); pk[f"b{nb}.{nm}.b"] = f32(sd[f"{p}.{nm}.bias"])
        pk[f"b{nb}.in.w"] = pack_linear(t(f"{p}.attn.in_proj_weight")); pk[f"b{nb}.in.b"] = f32(sd[f"{p}.attn.in_proj_bias"])
        pk[f"b{nb}.out.w"] = pack_linear(t(f"{p}.attn.out_proj.weight")); pk[f"b{nb}.out.b"] = f32(sd[f"{p}.attn.out_proj.bias"])
        pk[f"b{nb}.fc1.w"] = pack_linear(t(f"{p}.mlp.0.weight")); pk[f"b{nb}.fc1.b"] = f32(sd[f"{p}.mlp.0.bias"])
        pk[f"b{nb}.fc2.w"] = pack_linear(t(f"{p}.mlp.2.weight")); pk[f"b{nb}.fc2.b"] = f32(sd[f"{p}.mlp.2.bias"])
        nb += 1
    pk["nblocks"] = nb
    pk["pu.w"] = pack_linear(t("patch_unembed.weight").permute(2, 3, 1, 0).reshape(4096, 128)); pk["pu.b"] = f32(sd["patch_unembed.bias"])
    pk["dec1.w"], pk["dec1.b"] = pack_conv_c64(t("decoder_conv1.weight"), t("decoder_conv1.bias"), 1)
    pk["dec2.w"] = pack_conv_c64_thin(t("decoder_conv2.weight")); pk["dec2.b"] = f32(sd["decoder_conv2.bias"])
    if not backward:
        return pk
    pk["conv2.wd"] = pack_conv_c64_dgrad(t("conv2.weight"), 1)
    pk["ds.wd"] = pack_conv_c64_stride2_dgrad(t("downsample.weight"))
    pk["dec1.wd"] = pack_conv_c64_dgrad(t("decoder_conv1.weight"), 1)
    pk["dec2.wd"] = pack_conv_thin_dgrad(t("decoder_conv2.weight"))
    pk["pe.wd"] = pack_linear(t("patch_embed.weight").permute(2, 3, 1, 0).reshape(4096, 128))     # rows (i,j,c), cols n
    pk["pu.wd"] = pack_linear(t("patch_unembed.weight").permute(0, 2, 3, 1).reshape(128, 4096))   # rows k, cols (i,j,o)
    for i in range(nb):
        p = f"transformer_blocks.{i}"
        for nm, key in (("in", "attn.in_proj_weight"), ("out", "attn.out_proj.weight"), ("fc1", "mlp.0.weight"), ("fc2", "mlp.2.weight")):
            pk[f"b{i}.{nm}.wd"] = pack_linear(t(f"{p}.{key}").t().contiguous())
    return pk


# ------------------------------------------------------------------------------------------------
# WindowTransformer
# ------------------------------------------------------------------------------------------------
def pack_wt_state_dict(sd: Dict[str, torch.Tensor], backward: bool = False) -> Dict[str, torch.Tensor]:
    pk: Dict[str, torch.Tensor] = {}
    f32 = lambda t: t.detach().float().contiguous()
    t = lambda k: sd[k].detach()
    dim = sd["patch_embed.weight"].shape[0]
    pk["conv1.w"] = pack_conv1(t("conv1.weight")); pk["conv1.b"] = f32(sd["conv1.bias"])
    pk["conv2.w"], pk["conv2.b"] = pack_conv_c64(t("conv2.weight"), t("conv2.bias"), 1)
    pk["ds.w"], pk["ds.b"] = pack_conv_c64_stride2(t("downsample.weight"), t("downsample.bias"))
    pk["pe.w"] = pack_linear(t("patch_embed.weight").permute(0, 2, 3, 1).reshape(dim, 4096)); pk["pe.b"] = f32(sd["patch_embed.bias"])
    nb = 0
    while f"window_blocks.{nb}.norm1.weight" in sd:
        p = f"window_blocks.{nb}"
        for nm in ("norm1", "norm2"):
            pk[f"b{nb}.{nm}.w"] = f32(sd[f"{p}.{nm}.weight"]); pk[f"b{nb}.{nm}.b"] = f32(sd[f"{p}.{nm}.bias"])
        pk[f"b{nb}.table"] = f32(sd[f"{p}.attn.relative_position_bias_table"])
        for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            pk[f"b{nb}.{nm}.w"] = pack_linear(t(f"{p}.{key}.weight")); pk[f"b{nb}.{nm}.b"] = f32(sd[f"{p}.{key}.bias"])
        nb += 1
    pk["nblocks"] = nb
    pk["pu.w"] = pack_linear(t("patch_unembed.weight").permute(2, 3, 1, 0).reshape(4096, dim)); pk["pu.b"] = f32(sd["patch_unembed.bias"])
    pk["dec1.w"], pk["dec1.b"] = pack_conv_c64(t("decoder_conv1.weight"), t("decoder_conv1.bias"), 1)
    pk["dec2.w"] = pack_conv_c64_thin(t("decoder_conv2.weight")); pk["dec2.b"] = f32(sd["decoder_conv2.bias"])
    if not backward:
        return pk
    pk["conv2.wd"] = pack_conv_c64_dgrad(t("conv2.weight"), 1)
    pk["ds.wd"] = pack_conv_c64_stride2_dgrad(t("downsample.weight"))
    pk["dec1.wd"] = pack_conv_c64_dgrad(t("decoder_conv1.weight"), 1)
    pk["dec2.wd"] = pack_conv_thin_dgrad(t("decoder_conv2.weight"))
    pk["pe.wd"] = pack_linear(t("patch_embed.weight").permute(2, 3, 1, 0).reshape(4096, dim))      # rows (i,j,c), cols n
    pk["pu.wd"] = pack_linear(t("patch_unembed.weight").permute(0, 2, 3, 1).reshape(dim, 4096))    # rows k, cols (i,j,o)
    for i in range(nb):
        p = f"window_blocks.{i}"
        for nm, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            pk[f"b{i}.{nm}.wd"] = pack_linear(t(f"{p}.{key}.weight").t().contiguous())
    return pk
