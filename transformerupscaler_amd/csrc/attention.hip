// LayerNorm and the fused 8x8-window multi-head attention core (gfx950).
//
//   tup_layernorm_fwd       nn.LayerNorm(192), eps 1e-5     models/FastTransformer/model.py:142,144,163,169
//   tup_relpos_bias_expand  table[index] gather -> dense per-head bias   model.py:120-123
//   tup_window_attn_fwd     q*scale, q k^T + bias, softmax, P v, head concat   model.py:114-130
//
// Attention: one wave per (window, head).  N = 64 tokens, head_dim = 16, so S^T = K Q^T is 4x4
// tiles of v_mfma_f32_16x16x16_bf16 (K = head_dim exactly) and stays in registers: a lane owns
// one query column; its 16 in-lane values + 2 cross-lane shuffles give the softmax row
// statistics, and the bf16 P^T tiles are already in the B-operand layout of the second product
// O^T = V^T P^T, so P never touches LDS or HBM.
#include "common.h"

namespace {

constexpr int DIM = 192, HD = 16, NTOK = 64;

__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int M)
{
    // 16 lanes per row, 12 elements per lane; a 256-thread block normalises 16 rows
    const int sub = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < M;
    const float* xr = x + (size_t)(ok ? row : 0) * DIM;
    f32x4 v[3];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        v[q] = *reinterpret_cast<const f32x4*>(xr + q * 64 + sub * 4);
        s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / DIM);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mean; ss += d * d; }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = q * 64 + sub * 4;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[q][e] - mean) * rstd * gm[e] + bt[e];
        *reinterpret_cast<u32x2*>(y + (size_t)row * DIM + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
    if (sub == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// bias_frag[h][kt][qt][lane][e] = table[index(query, key)][h], query = 16qt + (lane&15),
// key = 16kt + 4(lane>>4) + e: the C/D fragment layout of the S^T tiles.
template <int NH>
__global__ void relpos_expand_kernel(const float* __restrict__ table, float* __restrict__ frag)
{
    constexpr int HEADS = NH;
    const int idx = blockIdx.x * 256 + threadIdx.x;        // over NH*4*4*64*4
    if (idx >= HEADS * 16 * 256) return;
    const int e = idx & 3, lane = (idx >> 2) & 63, qt = (idx >> 8) & 3, kt = (idx >> 10) & 3, h = idx >> 12;
    const int qi = 16 * qt + (lane & 15), kj = 16 * kt + 4 * (lane >> 4) + e;
    const int rel = ((qi >> 3) - (kj >> 3) + 7) * 15 + ((qi & 7) - (kj & 7) + 7);   // model.py:89-100
    frag[idx] = table[rel * HEADS + h];
}

template <int NH>       // heads of 16: 12 (FastTransformer, dim 192) or 8 (WindowTransformer, dim 128)
__global__ __launch_bounds__(256) void window_attn_kernel(
    const bf16_t* __restrict__ qkv, const float* __restrict__ bias_frag, bf16_t* __restrict__ out, float* __restrict__ lse, int npairs,
    uint32_t drop_thresh, float drop_inv_keep, uint32_t drop_seed)
{
    constexpr int HEADS = NH, DIM = NH * HD;
    __shared__ __attribute__((aligned(16))) bf16_t vlds[4][NTOK * HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    const int pair = blockIdx.x * 4 + wave;
    const bool active = pair < npairs;
    const int win = active ? pair / HEADS : 0, h = active ? pair % HEADS : 0;
    const bf16_t* base = qkv + (size_t)win * NTOK * (3 * DIM) + h * HD;

    // stage V [key][hd] (32-byte rows) for the transposed fragment reads
    {
        const int key = lane;
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(base + (size_t)key * (3 * DIM) + 2 * DIM);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(base + (size_t)key * (3 * DIM) + 2 * DIM + 8);
        *reinterpret_cast<u32x4*>(&vlds[wave][key * HD]) = v0;
        *reinterpret_cast<u32x4*>(&vlds[wave][key * HD + 8]) = v1;
    }

    s16x4 kf[4], qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t roff = (size_t)(16 * t + p) * (3 * DIM) + 4 * g;
        qf[t] = *reinterpret_cast<const s16x4*>(base + roff);
        kf[t] = *reinterpret_cast<const s16x4*>(base + roff + DIM);
    }

    // S^T tiles: st[kt][qt][e] = S[query 16qt+p][key 16kt+4g+e]
    f32x4 st[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(bias_frag + ((((size_t)h * 4 + kt) * 4 + qt) * 64 + lane) * 4);
            const f32x4 s = mfma16x16x16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int e = 0; e < 4; ++e) st[kt][qt][e] = s[e] * 0.25f + bf[e];   // q*scale, scale = 16^-0.5
        }

    float inv[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, st[kt][qt][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = __expf(st[kt][qt][e] - mx);
                st[kt][qt][e] = ex;
                sum += ex;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[qt] = 1.0f / sum;
        // training: the row's log-sum-exp is all the backward needs to rebuild P (csrc/attention_bwd.hip)
        if (lse != nullptr && active && g == 0) lse[(size_t)pair * NTOK + 16 * qt + p] = mx + __logf(sum);
    }

    __syncthreads();   // V staged (per-wave region, but keep it simple: one barrier)
    // V^T fragments: A[row = hd p][k = key 16kt + 4g + j]
    s16x4 vf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        bf16x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = vlds[wave][(16 * kt + 4 * g + j) * HD + p];
        vf[kt] = __builtin_bit_cast(s16x4, t);
    }

#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            // normalise before the bf16 rounding of P (softmax output is what the reference multiplies by v);
            // attn_drop (model.py:127): element index = (pair*64 + query)*64 + key.
            // Two key tiles per v_mfma_f32_16x16x32_bf16 (twice the rate of the x16 form): k index 8g + j <-> key
            // 4g + j of tile 2kp (j < 4) or of tile 2kp + 1 (j >= 4), the same map on both operands.
            uint32_t pw[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int kt = 2 * kp + hh;
                float pv[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop_thresh) drop_pair4(drop_seed, ((uint32_t)pair * 64u + 16u * qt + p) * 64u + 16u * kt + 4u * g, drop_thresh, drop_inv_keep, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[e] = st[kt][qt][e] * inv[qt] * dm[e];
                pw[2 * hh] = pack_bf16x2(pv[0], pv[1]);
                pw[2 * hh + 1] = pack_bf16x2(pv[2], pv[3]);
            }
            const u32x2 va = __builtin_bit_cast(u32x2, vf[2 * kp]), vb = __builtin_bit_cast(u32x2, vf[2 * kp + 1]);
            o = mfma16x16x32(__builtin_bit_cast(bf16x8, u32x4{va[0], va[1], vb[0], vb[1]}),
                             __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]}), o);
        }
        // O^T tile: rows = hd 4g+e, col = query p  ->  out[win][16qt+p][h*16 + 4g .. +3]
        if (active) {
            bf16_t* op = out + ((size_t)win * NTOK + 16 * qt + p) * DIM + h * HD + 4 * g;
            *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

}  // namespace

// x: fp32 [M][192]; y: bf16 [M][192]; mean/rstd: optional fp32 [M] (saved for backward).
extern "C" int tup_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int M, void* stream)
{
    if (M <= 0) return 0;
    layernorm_kernel<<<dim3((M + 15) / 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        x, gamma, beta, (bf16_t*)y, mean, rstd, M);
    TUP_CHECK_LAUNCH();
    return 0;
}

// table: fp32 [225][12] (relative_position_bias_table); frag: fp32 [12*16*256] in S^T fragment order.
extern "C" int tup_relpos_bias_expand(const float* table, float* frag, void* stream)
{
    relpos_expand_kernel<12><<<dim3(12 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(table, frag);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Same for `heads` = 8 (WindowTransformer: table fp32 [225][8], frag fp32 [8*16*256]) or 12.
extern "C" int tup_relpos_bias_expand_h(const float* table, float* frag, int heads, void* stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (heads == 12) relpos_expand_kernel<12><<<dim3(12 * 16), dim3(256), 0, s>>>(table, frag);
    else if (heads == 8) relpos_expand_kernel<8><<<dim3(8 * 16), dim3(256), 0, s>>>(table, frag);
    else return (int)hipErrorInvalidValue;
    TUP_CHECK_LAUNCH();
    return 0;
}

// qkv: bf16 [nwin][64][576] (q | k | v, each head-major 12 x 16); out: bf16 [nwin][64][192].
// drop_p > 0 applies attn_drop (model.py:80,127) with the stateless mask of common.h keyed by drop_seed.
// lse: NULL, or fp32 [nwin][12][64] <- log-sum-exp of every score row (what tup_window_attn_bwd rebuilds P from).
extern "C" int tup_window_attn_fwd(const void* qkv, const float* bias_frag, void* out, float* lse, int nwin, float drop_p,
                                   unsigned int drop_seed, void* stream)
{
    if (nwin <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f) return (int)hipErrorInvalidValue;
    const int npairs = nwin * 12;
    uint32_t thresh; float inv_keep;
    drop_pair_params(drop_p, thresh, inv_keep);
    window_attn_kernel<12><<<dim3((npairs + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)qkv, bias_frag, (bf16_t*)out, lse, npairs, thresh, inv_keep, drop_seed);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Same kernel for `heads` x 16 channels (8: WindowTransformer, models/WindowTransformer/model.py:67-143): qkv bf16
// [nwin][64][3*16*heads], out bf16 [nwin][64][16*heads], lse NULL or fp32 [nwin][heads][64].
extern "C" int tup_window_attn_fwd_h(const void* qkv, const float* bias_frag, void* out, float* lse, int nwin, int heads, float drop_p,
                                     unsigned int drop_seed, void* stream)
{
    if (nwin <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f) return (int)hipErrorInvalidValue;
    const int npairs = nwin * heads;
    uint32_t thresh; float inv_keep;
    drop_pair_params(drop_p, thresh, inv_keep);
    const dim3 grid((npairs + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (heads == 12)
        window_attn_kernel<12><<<grid, dim3(256), 0, s>>>((const bf16_t*)qkv, bias_frag, (bf16_t*)out, lse, npairs, thresh, inv_keep, drop_seed);
    else if (heads == 8)
        window_attn_kernel<8><<<grid, dim3(256), 0, s>>>((const bf16_t*)qkv, bias_frag, (bf16_t*)out, lse, npairs, thresh, inv_keep, drop_seed);
    else return (int)hipErrorInvalidValue;
    TUP_CHECK_LAUNCH();
    return 0;
}
