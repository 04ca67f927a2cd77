// Token-side GEMMs of the FastTransformer path on MFMA (gfx950), with gather / scatter
// addressing and fused epilogues.  out[m][n] = epi( sum_k A[m][k] * Wt[n][k] + bias[n] ).
//
// Reference call sites replaced (aten linear / conv2d k8s8 / conv_transpose2d k8s8):
//   qkv, proj            models/FastTransformer/model.py:79,81,115,131
//   mlp.0 (+erf GELU), mlp.2   model.py:146-151,170
//   residual adds        model.py:164,171
//   patch_embed + NCHW->NHWC + zero token pad + window_partition   model.py:31-45,215,268-285
//   window_reverse + crop + patch_unembed + crop + skip add        model.py:47-63,225,292-309
//
// One workgroup = 4 waves = 128 token rows x 64 output features, K streamed in 64-wide
// chunks through double-buffered, XOR-swizzled LDS.  MFMA A operand = weight rows, B operand =
// token rows, so each lane finishes with 16 consecutive output features of one token.
// Weight rows are pre-permuted on the host inside every 64-row group (row ct*16+4g+e holds
// feature g*16+ct*4+e) to make that true.
#include "common.h"
#include <stdlib.h>

namespace {

enum { A_BF16 = 0, A_F32 = 1, A_PATCH = 2, A_LN = 3 };
enum { E_BF16 = 0, E_GELU_BF16 = 1, E_RES_F32 = 2, E_PATCH_EMBED = 3, E_UNEMBED = 4, E_GELU_BWD = 5, E_UNEMBED_MERGE = 6 };

struct GemmParams {
    const void* A; int lda;
    const bf16_t* Wt;            // [N][K] bf16, rows permuted per 64-group
    const float* bias;           // [N] natural order (E_UNEMBED: [64])
    void* out; int ldo;
    const float* res;            // E_RES_F32: [M][ldo] fp32
    const bf16_t* skip;          // E_UNEMBED: NHWC feat to add (or null); E_GELU_BWD: pre-activation [M][ldo]
    const bf16_t* skip2; const bf16_t* relu_src;      // E_UNEMBED_MERGE (panel kernel): a second map to add (or null) and the map whose sign gates the sum
    float* skip_colsum;                               // E_UNEMBED_MERGE: optional fp32 [16][64] += per-channel sums of `skip` (16 replicas, spread by workgroup)
    int reflect;                 // A_PATCH: 1 = reflect-pad beyond the map, 0 = zeros
    uint32_t drop_thresh, drop_seed; float drop_inv_keep;      // E_RES_F32: dropout on (acc + bias) before "+ res"
    const float* ln_gamma; const float* ln_beta;               // panel kernel, A_LN: LayerNorm fused into the A load
    int linear_tokens;            // patch modes: token rows are a plain [B][Ht][Wt] grid (ResidualTransformer), not windows
    const float* pos;             // E_PATCH_EMBED with linear tokens: + pos_embed[token][n]
    int M, N, K;
    int H, W, Ht, Wt_, nWx, nWy; // geometry for the patch modes (token rows are in window layout)
};

constexpr int BM = 128, BN = 64, BK = 64;
constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128;

struct TokPos { int b, ty, tx; bool valid; };

TUP_DEVICE TokPos token_of_row(int m, const GemmParams& p) {
    TokPos t;
    if (p.linear_tokens) {
        const int per = p.Ht * p.Wt_;
        t.b = m / per;
        const int tok = m - t.b * per;
        t.ty = tok / p.Wt_;
        t.tx = tok - t.ty * p.Wt_;
        t.valid = true;
        return t;
    }
    const int tok = m & 63;
    int win = m >> 6;
    const int wx = win % p.nWx; win /= p.nWx;
    const int wy = win % p.nWy;
    t.b = win / p.nWy;
    t.ty = wy * 8 + (tok >> 3);
    t.tx = wx * 8 + (tok & 7);
    t.valid = (t.ty < p.Ht) && (t.tx < p.Wt_);
    return t;
}

// Epilogue of one token row: the lane holds v[16] = features n0 + g*16 .. +15 of row m (accumulators), bvec = the
// matching bias values.  Shared by the K-streaming kernel and the A-resident panel kernel.
template <int EPI>
TUP_DEVICE void gemm_store_row(const GemmParams& p, int m, int n0, int g, const float (&v)[16], const float (&bvec)[16])
{
    const int nb = n0 + g * 16;
    if constexpr (EPI == E_BF16 || EPI == E_GELU_BF16) {
        uint32_t pk[8], ppre[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a = v[2 * q] + bvec[2 * q], b = v[2 * q + 1] + bvec[2 * q + 1];
            if constexpr (EPI == E_GELU_BF16) {
                ppre[q] = pack_bf16x2(a, b);      // pre-activation, saved (bf16) by the training path
                const f32x2 gv = gelu_erf2(f32x2{a, b});
                a = gv[0];
                b = gv[1];
            }
            pk[q] = pack_bf16x2(a, b);
        }
        if constexpr (EPI == E_GELU_BF16) {
            if (p.skip) {
                bf16_t* po = const_cast<bf16_t*>(p.skip) + (size_t)m * p.ldo + nb;
                *reinterpret_cast<u32x4*>(po) = u32x4{ppre[0], ppre[1], ppre[2], ppre[3]};
                *reinterpret_cast<u32x4*>(po + 8) = u32x4{ppre[4], ppre[5], ppre[6], ppre[7]};
            }
        }
        bf16_t* o = (bf16_t*)p.out + (size_t)m * p.ldo + nb;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    } else if constexpr (EPI == E_RES_F32) {
        float* o = (float*)p.out + (size_t)m * p.ldo + nb;
        const float* rs = p.res + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rs + 4 * q);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[4 * q + e] + bvec[4 * q + e];
                if (p.drop_thresh)      // proj_drop / mlp Dropout (model.py:132,150): element index m*ldo + n
                    t *= drop_scale(p.drop_seed, (uint32_t)m * (uint32_t)p.ldo + nb + 4 * q + e, p.drop_thresh, p.drop_inv_keep);
                ov[e] = t + rv[e];
            }
            *reinterpret_cast<f32x4*>(o + 4 * q) = ov;
        }
    } else if constexpr (EPI == E_PATCH_EMBED) {
        // zero-padded tokens are exact zeros (no bias): model.py:273-280
        const TokPos t = token_of_row(m, p);
        float* o = (float*)p.out + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = t.valid ? v[4 * q + e] + bvec[4 * q + e] : 0.f;
            if (p.pos) {       // ResidualTransformer: tokens + pos_embed (model.py:140)
                const f32x4 pv = *reinterpret_cast<const f32x4*>(p.pos + (size_t)(m % (p.Ht * p.Wt_)) * p.ldo + nb + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] += pv[e];
            }
            *reinterpret_cast<f32x4*>(o + 4 * q) = ov;
        }
    } else if constexpr (EPI == E_GELU_BWD) {
        // out = acc * gelu'(pre), pre = saved fc1 output before the activation (model.py:148)
        const bf16_t* pr = p.skip + (size_t)m * p.ldo + nb;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(pr), a1 = *reinterpret_cast<const u32x4*>(pr + 8);
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t sw = (q < 4) ? a0[q & 3] : a1[q & 3];
            const float xa = __builtin_bit_cast(float, sw << 16), xb = __builtin_bit_cast(float, sw & 0xffff0000u);
            pk[q] = pack_bf16x2(v[2 * q] * gelu_erf_grad(xa), v[2 * q + 1] * gelu_erf_grad(xb));
        }
        bf16_t* o = (bf16_t*)p.out + (size_t)m * p.ldo + nb;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    } else {  // E_UNEMBED: n tile = patch pixel (i, j); features = base channel o
        const TokPos t = token_of_row(m, p);
        const int pix = n0 >> 6, i = pix >> 3, j = pix & 7;
        const int py = t.ty * 8 + i, px = t.tx * 8 + j;
        if (!t.valid || py >= p.H || px >= p.W) return;
        const size_t off = (((size_t)t.b * p.H + py) * p.W + px) * 64 + g * 16;
        u32x4 s0 = {0u, 0u, 0u, 0u}, s1 = {0u, 0u, 0u, 0u};
        if (p.skip) {
            s0 = *reinterpret_cast<const u32x4*>(p.skip + off);
            s1 = *reinterpret_cast<const u32x4*>(p.skip + off + 8);
        }
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t sw = (q < 4) ? s0[q & 3] : s1[q & 3];
            const float sa = __builtin_bit_cast(float, sw << 16);
            const float sb = __builtin_bit_cast(float, sw & 0xffff0000u);
            pk[q] = pack_bf16x2(v[2 * q] + bvec[2 * q] + sa, v[2 * q + 1] + bvec[2 * q + 1] + sb);
        }
        bf16_t* o = (bf16_t*)p.out + off;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
}

// The same epilogues split in two for the panel kernel: the operands an epilogue READS (residual row, skip pixels,
// saved pre-activation) are requested before the K loop of the tile and consumed after it, so their HBM round trip
// hides under the MFMAs instead of being paid between the K loop and the stores of every 64-column tile.
// store sink for rows outside the output (keeps the number of store instructions per wave fixed); never read
__device__ __attribute__((aligned(16))) unsigned int tup_gemm_sink[64 * 8];
// 16-byte global loads the compiler does not track (no s_waitcnt of its own): the caller waits with a counted vmcnt and must
// touch the outputs (asm "+v") only after that wait
TUP_DEVICE u32x4 global_load_b128_async(const void* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
TUP_DEVICE u32x4 global_load_b128_async_16(const void* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

struct EpiPre {
    u32x4 a[6];
    size_t off;          // E_UNEMBED: element offset of this lane's 16 channels
    bool ok;
};

template <int EPI>
TUP_DEVICE EpiPre epi_prefetch(const GemmParams& p, int m, int n0, int g)
{
    EpiPre r;
    r.ok = true; r.off = 0;
    const int nb = n0 + g * 16;
    if constexpr (EPI == E_RES_F32) {
        const float* rs = p.res + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int q = 0; q < 4; ++q) r.a[q] = *reinterpret_cast<const u32x4*>(rs + 4 * q);
    } else if constexpr (EPI == E_GELU_BWD) {
        const bf16_t* pr = p.skip + (size_t)m * p.ldo + nb;
        r.a[0] = *reinterpret_cast<const u32x4*>(pr); r.a[1] = *reinterpret_cast<const u32x4*>(pr + 8);
    } else if constexpr (EPI == E_UNEMBED) {
        const TokPos t = token_of_row(m, p);
        const int pix = n0 >> 6, i = pix >> 3, j = pix & 7;
        const int py = t.ty * 8 + i, px = t.tx * 8 + j;
        r.ok = t.valid && py < p.H && px < p.W;
        r.off = (((size_t)t.b * p.H + py) * p.W + px) * 64 + g * 16;
        r.a[0] = u32x4{0u, 0u, 0u, 0u}; r.a[1] = u32x4{0u, 0u, 0u, 0u};
        if (p.skip && r.ok) {
            r.a[0] = *reinterpret_cast<const u32x4*>(p.skip + r.off);
            r.a[1] = *reinterpret_cast<const u32x4*>(p.skip + r.off + 8);
        }
    } else if constexpr (EPI == E_UNEMBED_MERGE) {
        // training: the gradient merge at `feat` (model.py:264,268,308 fan-out + conv2's ReLU) in the epilogue of patch_embed's input
        // gradient: out = (acc + skip + skip2) * (relu_src > 0).  Every map is the UNPADDED H x W map (H, W multiples of 8).
        const TokPos t = token_of_row(m, p);
        const int pix = n0 >> 6, i = pix >> 3, j = pix & 7;
        const int py = t.ty * 8 + i, px = t.tx * 8 + j;
        r.ok = t.valid && py < p.H && px < p.W;
        r.off = (((size_t)t.b * p.H + py) * p.W + px) * 64 + g * 16;
#pragma unroll
        for (int q = 0; q < 6; ++q) r.a[q] = u32x4{0u, 0u, 0u, 0u};
        if (r.ok) {
            r.a[0] = *reinterpret_cast<const u32x4*>(p.skip + r.off);     r.a[1] = *reinterpret_cast<const u32x4*>(p.skip + r.off + 8);
            if (p.skip2) { r.a[2] = *reinterpret_cast<const u32x4*>(p.skip2 + r.off); r.a[3] = *reinterpret_cast<const u32x4*>(p.skip2 + r.off + 8); }
            r.a[4] = *reinterpret_cast<const u32x4*>(p.relu_src + r.off); r.a[5] = *reinterpret_cast<const u32x4*>(p.relu_src + r.off + 8);
        }
    }
    return r;
}

template <int EPI>
TUP_DEVICE void epi_finish(const GemmParams& p, int m, int n0, int g, const float (&v)[16], const float (&bvec)[16], const EpiPre& pre)
{
    const int nb = n0 + g * 16;
    if constexpr (EPI == E_RES_F32) {
        float* o = (float*)p.out + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 rv = __builtin_bit_cast(f32x4, pre.a[q]);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[4 * q + e] + bvec[4 * q + e];
                if (p.drop_thresh)
                    t *= drop_scale(p.drop_seed, (uint32_t)m * (uint32_t)p.ldo + nb + 4 * q + e, p.drop_thresh, p.drop_inv_keep);
                ov[e] = t + rv[e];
            }
            *reinterpret_cast<f32x4*>(o + 4 * q) = ov;
        }
    } else if constexpr (EPI == E_GELU_BWD) {
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t sw = (q < 4) ? pre.a[0][q & 3] : pre.a[1][q & 3];
            const float xa = __builtin_bit_cast(float, sw << 16), xb = __builtin_bit_cast(float, sw & 0xffff0000u);
            pk[q] = pack_bf16x2(v[2 * q] * gelu_erf_grad(xa), v[2 * q + 1] * gelu_erf_grad(xb));
        }
        bf16_t* o = (bf16_t*)p.out + (size_t)m * p.ldo + nb;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    } else if constexpr (EPI == E_UNEMBED) {
        if (!pre.ok) return;
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t sw = (q < 4) ? pre.a[0][q & 3] : pre.a[1][q & 3];
            const float sa = __builtin_bit_cast(float, sw << 16);
            const float sb = __builtin_bit_cast(float, sw & 0xffff0000u);
            pk[q] = pack_bf16x2(v[2 * q] + bvec[2 * q] + sa, v[2 * q + 1] + bvec[2 * q + 1] + sb);
        }
        bf16_t* o = (bf16_t*)p.out + pre.off;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    } else if constexpr (EPI == E_UNEMBED_MERGE) {
        if (!pre.ok) return;
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t s1 = (q < 4) ? pre.a[0][q & 3] : pre.a[1][q & 3], s2 = (q < 4) ? pre.a[2][q & 3] : pre.a[3][q & 3];
            const uint32_t mw = (q < 4) ? pre.a[4][q & 3] : pre.a[5][q & 3];
            // (a + b) + this GEMM's value, which stays fp32 here (feat_grad_combine_kernel read it back rounded to bf16)
            float ra = __builtin_bit_cast(float, s1 << 16) + __builtin_bit_cast(float, s2 << 16);
            float rb = __builtin_bit_cast(float, s1 & 0xffff0000u) + __builtin_bit_cast(float, s2 & 0xffff0000u);
            ra += v[2 * q]; rb += v[2 * q + 1];
            if (!(__builtin_bit_cast(float, mw << 16) > 0.f)) ra = 0.f;
            if (!(__builtin_bit_cast(float, mw & 0xffff0000u) > 0.f)) rb = 0.f;
            pk[q] = pack_bf16x2(ra, rb);
        }
        bf16_t* o = (bf16_t*)p.out + pre.off;
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    } else {
        gemm_store_row<EPI>(p, m, n0, g, v, bvec);          // nothing to read ahead
    }
}

TUP_DEVICE void gemm_load_bias(const GemmParams& p, int n0, int g, bool unembed, float (&bvec)[16])
{
    const int bb = unembed ? g * 16 : n0 + g * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + bb + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) bvec[4 * q + e] = t[e];
    }
}

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tokens_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_lds = smem;                       // 2 x A_STAGE
    char* w_lds = smem + 2 * A_STAGE;         // 2 x W_STAGE

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int nk = p.K / BK;

    // ---- per-thread A row bookkeeping: thread loads chunk (tid&7) of rows (tid>>3) + 32u ----
    const int achunk = tid & 7;
    const char* arow[4];
    bool avalid[4];
    int apy[4], apx[4], abat[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int m = m0 + (tid >> 3) + 32 * u;
        avalid[u] = m < p.M;
        if (m >= p.M) m = p.M - 1;
        if constexpr (AMODE == A_BF16) arow[u] = (const char*)p.A + (size_t)m * p.lda * 2;
        else if constexpr (AMODE == A_F32) arow[u] = (const char*)p.A + (size_t)m * p.lda * 4;
        else {
            const TokPos t = token_of_row(m, p);
            avalid[u] = avalid[u] && t.valid;
            apy[u] = t.ty * 8; apx[u] = t.tx * 8; abat[u] = t.b;
            arow[u] = nullptr;
        }
    }

    u32x4 areg[(AMODE == A_F32) ? 8 : 4];
    u32x4 wreg[2];

    auto load_stage = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (AMODE == A_BF16) {
                areg[u] = avalid[u] ? *reinterpret_cast<const u32x4*>(arow[u] + (size_t)(kc * BK + achunk * 8) * 2)
                                    : u32x4{0u, 0u, 0u, 0u};
            } else if constexpr (AMODE == A_F32) {
                const char* s = arow[u] + (size_t)(kc * BK + achunk * 8) * 4;
                areg[2 * u] = avalid[u] ? *reinterpret_cast<const u32x4*>(s) : u32x4{0u, 0u, 0u, 0u};
                areg[2 * u + 1] = avalid[u] ? *reinterpret_cast<const u32x4*>(s + 16) : u32x4{0u, 0u, 0u, 0u};
            } else {
                // k chunk kc = patch pixel (i, j); reflect-pad rows/cols beyond the map (model.py:256-261)
                const int i = kc >> 3, j = kc & 7;
                int py = apy[u] + i, px = apx[u] + j;
                bool ok = avalid[u];
                if (p.reflect) {
                    if (py >= p.H) py = 2 * p.H - 2 - py;
                    if (px >= p.W) px = 2 * p.W - 2 - px;
                } else if (py >= p.H || px >= p.W) {
                    ok = false; py = 0; px = 0;
                }
                const bf16_t* s = (const bf16_t*)p.A + (((size_t)abat[u] * p.H + py) * p.W + px) * 64 + achunk * 8;
                areg[u] = ok ? *reinterpret_cast<const u32x4*>(s) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 256, row = idx >> 3, c = idx & 7;
            wreg[u] = *reinterpret_cast<const u32x4*>(p.Wt + (size_t)(n0 + row) * p.K + kc * BK + c * 8);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = (tid >> 3) + 32 * u;
            u32x4 v;
            if constexpr (AMODE == A_F32) {
                const f32x4 lo = __builtin_bit_cast(f32x4, areg[2 * u]);
                const f32x4 hi = __builtin_bit_cast(f32x4, areg[2 * u + 1]);
                v = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]),
                          pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
            } else {
                v = areg[u];
            }
            *reinterpret_cast<u32x4*>(a_lds + buf * A_STAGE + swz128(row, achunk)) = v;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 256;
            *reinterpret_cast<u32x4*>(w_lds + buf * W_STAGE + swz128(idx >> 3, idx & 7)) = wreg[u];
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_stage(kc + 1);
        const char* ab = a_lds + buf * A_STAGE;
        const char* wb = w_lds + buf * W_STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int chunk = kh * 4 + g;
            bf16x8 tf[2], wf[4];
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
                tf[tg] = *reinterpret_cast<const bf16x8*>(ab + swz128(32 * wave + 16 * tg + pl, chunk));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                wf[ct] = *reinterpret_cast<const bf16x8*>(wb + swz128(ct * 16 + pl, chunk));
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = mfma16x16x32(wf[ct], tf[tg], acc[tg][ct]);
        }
        if (kc + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds features n0 + g*16 + ct*4 + e of token row m ----
    float bvec[16];
    gemm_load_bias(p, n0, g, EPI == E_UNEMBED, bvec);
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const int m = m0 + 32 * wave + 16 * tg + pl;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[ct * 4 + e] = acc[tg][ct][e];
        gemm_store_row<EPI>(p, m, n0, g, v, bvec);
    }
}

// ------------------------------------------------------------------------------------------------
// A-resident "panel" kernel for K = 192 (every Linear that reads the 192-wide token stream, and
// patch_unembed): the [128][192] token tile is loaded ONCE per workgroup -- optionally through a fused
// LayerNorm (A_LN) -- and stays in LDS while the workgroup walks all N/64 weight tiles, each streamed
// global -> registers -> LDS one tile ahead.  The K-streaming kernel above pays a global-load round trip
// per 16 MFMAs at K = 192 (3 iterations per output tile) and was latency-bound; here it is paid once
// per 128 rows.  LDS: 48 KB (A) + 24 KB (W) -> two workgroups per CU.  Fragment reads are hand-pipelined.
// ------------------------------------------------------------------------------------------------
constexpr int PK = 192, PBM = 128;
constexpr int PA_BYTES = 3 * PBM * 128, PW_BYTES = 3 * 64 * 128;

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_panel_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_lds = smem;
    char* w_lds = smem + PA_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * PBM;
    const int ntiles = p.N / 64;

    u32x4 wreg[6];
    auto load_w = [&](int nt) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = tid + u * 256;                  // 1536 chunks: row = idx / 24, c = idx % 24
            const int row = idx / 24, c = idx - row * 24;
            wreg[u] = *reinterpret_cast<const u32x4*>(p.Wt + (size_t)(nt * 64 + row) * PK + c * 8);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = tid + u * 256;
            const int row = idx / 24, c = idx - row * 24;
            *reinterpret_cast<u32x4*>(w_lds + (c >> 3) * (64 * 128) + swz128(row, c & 7)) = wreg[u];
        }
    };
    load_w(0);

    // ---- A tile prologue ----
    if constexpr (AMODE == A_LN) {
        const int sub = tid & 15;
        f32x4 gm[3], bt[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            gm[q] = *reinterpret_cast<const f32x4*>(p.ln_gamma + q * 64 + sub * 4);
            bt[q] = *reinterpret_cast<const f32x4*>(p.ln_beta + q * 64 + sub * 4);
        }
#pragma unroll 2
        for (int pass = 0; pass < PBM / 16; ++pass) {
            const int r = pass * 16 + (tid >> 4);
            const int m = min(m0 + r, p.M - 1);
            f32x4 v[3];
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                v[q] = *reinterpret_cast<const f32x4*>((const float*)p.A + (size_t)m * p.lda + q * 64 + sub * 4);
                sum += v[q][0] + v[q][1] + v[q][2] + v[q][3];
            }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.0f / PK);
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mean; ss += d * d; }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            const float rstd = rsqrtf(ss * (1.0f / PK) + 1e-5f);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (v[q][e] - mean) * rstd * gm[q][e] + bt[q][e];
                *reinterpret_cast<u32x2*>(a_lds + q * (PBM * 128) + swz128(r, sub >> 1) + (sub & 1) * 8) =
                    u32x2{pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3])};
            }
        }
    } else {
#pragma unroll 4
        for (int u = 0; u < 12; ++u) {
            const int idx = tid + u * 256;                  // 3072 chunks of 8 elements: row = idx / 24
            const int row = idx / 24, c = idx - row * 24;
            const int m = min(m0 + row, p.M - 1);
            u32x4 v;
            if constexpr (AMODE == A_BF16) {
                v = *reinterpret_cast<const u32x4*>((const bf16_t*)p.A + (size_t)m * p.lda + c * 8);
            } else {
                const float* src = (const float*)p.A + (size_t)m * p.lda + c * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                v = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
            }
            *reinterpret_cast<u32x4*>(a_lds + (c >> 3) * (PBM * 128) + swz128(row, c & 7)) = v;
        }
    }

    const uint32_t a_tok0 = lds_addr(a_lds) + (uint32_t)swz128(32 * wave + pl, g);
    const uint32_t a_tok1 = lds_addr(a_lds) + (uint32_t)swz128(32 * wave + 16 + pl, g);
    const uint32_t w_frag = lds_addr(w_lds) + (uint32_t)swz128(pl, g);

    for (int nt = 0; nt < ntiles; ++nt) {
        store_w();
        __syncthreads();                        // W tile nt (and, first time, the A tile) visible
        if (nt + 1 < ntiles) load_w(nt + 1);
        // epilogue operands of THIS tile: requested now, consumed after the K loop
        const int n0 = nt * 64;
        float bvec[16];
        gemm_load_bias(p, n0, g, EPI == E_UNEMBED, bvec);
        EpiPre pre[2];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) pre[tg] = epi_prefetch<EPI>(p, min(m0 + 32 * wave + 16 * tg + pl, p.M - 1), n0, g);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // fragments two K-steps ahead (3-slot ring), addresses = 3 base registers + immediate offsets: one step is
            // 8 MFMAs = 128 cycles, the LDS round trip under load is several hundred (stamped in the conv kernel)
            bf16x8 tf[3][2], wf[3][4];
            auto ld = [&](int step, int slot) {
                const int kc = step >> 1;
                if (step & 1) {
                    tf[slot][0] = lds_read_b128_asm_off_x64(a_tok0, kc * (PBM * 128));
                    tf[slot][1] = lds_read_b128_asm_off_x64(a_tok1, kc * (PBM * 128));
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) wf[slot][ct] = lds_read_b128_asm_off_x64(w_frag, kc * (64 * 128) + ct * 2048);
                } else {
                    tf[slot][0] = lds_read_b128_asm_off(a_tok0, kc * (PBM * 128));
                    tf[slot][1] = lds_read_b128_asm_off(a_tok1, kc * (PBM * 128));
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) wf[slot][ct] = lds_read_b128_asm_off(w_frag, kc * (64 * 128) + ct * 2048);
                }
            };
            ld(0, 0);
            ld(1, 1);
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<12>(); }
                else if (step + 1 < 6) { lds_wait<6>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = mfma16x16x32(wf[cur][ct], tf[cur][tg], acc[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                        // everyone done reading W tile nt
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const int m = m0 + 32 * wave + 16 * tg + pl;
            if (m >= p.M) continue;
            float v[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[ct * 4 + e] = acc[tg][ct][e];
            epi_finish<EPI>(p, m, n0, g, v, bvec, pre[tg]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Panel kernel v2 (K = 192): the token operand never touches LDS.  Each lane loads its 32 rows directly in MFMA
// B-fragment order (token 16tg+pl, channels 32*step + 8g .. +8; A_LN: statistics by two cross-lane-group shuffles) and
// keeps the 12 bf16x8 fragments in registers for all N/64 weight tiles.  LDS holds only the weight stream: two 24 KB
// buffers filled by LDS-DMA one tile ahead, ONE workgroup barrier per tile (v1: global -> registers -> ds_write, two
// barriers).  K loop: weight fragments two K-steps ahead, immediate-offset reads.  Epilogue operands are requested
// before the K loop (epi_prefetch).  48 KB of LDS, <= 256 VGPRs: two 4-wave workgroups per CU.
// ------------------------------------------------------------------------------------------------
// (Measured: the A_F32 / E_UNEMBED instance built for THREE workgroups per CU -- 168 registers, 17 of them spilled -- makes the forward
// 0.2 ms slower, 4.12 vs 3.91 ms; two workgroups per CU it stays.)
template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_panel2_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x PW_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * PBM;
    const int ntiles = p.N / 64;

    // weight tile nt -> LDS buffer: slot s = u*256 + tid -> k-tile u >> 1, row (u & 1)*32 + (tid >> 3), logical chunk
    // (tid & 7) ^ ((tid >> 4) & 7) (swizzle on the source side); six pieces per thread, compile-time offsets
    const bf16_t* w_thr = p.Wt + (size_t)(tid >> 3) * PK + ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    auto dma_w = [&](int nt, int buf) {
        char* dst = smem + buf * PW_BYTES + wave * 1024;
        const bf16_t* src = w_thr + (size_t)nt * 64 * PK;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * PK + (u >> 1) * 64),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    dma_w(0, 0);

    // ---- token fragments ----
    bf16x8 tf[2][6];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const int m = min(m0 + 32 * wave + 16 * tg + pl, p.M - 1);
        if constexpr (AMODE == A_BF16) {
            const bf16_t* ar = (const bf16_t*)p.A + (size_t)m * p.lda + 8 * g;
#pragma unroll
            for (int st = 0; st < 6; ++st) tf[tg][st] = *reinterpret_cast<const bf16x8*>(ar + 32 * st);
        } else {
            const float* xr = (const float*)p.A + (size_t)m * p.lda + 8 * g;
            f32x4 v[6][2];
            float sum = 0.f;
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                v[st][0] = *reinterpret_cast<const f32x4*>(xr + 32 * st);
                v[st][1] = *reinterpret_cast<const f32x4*>(xr + 32 * st + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += v[st][0][e] + v[st][1][e];
            }
            float mean = 0.f, rstd = 1.f;
            if constexpr (AMODE == A_LN) {
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                mean = sum * (1.0f / PK);
                float ss = 0.f;
#pragma unroll
                for (int st = 0; st < 6; ++st)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float d = v[st][h][e] - mean; ss += d * d; }
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                rstd = rsqrtf(ss * (1.0f / PK) + 1e-5f);
            }
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                uint32_t pk[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float o4[4];
                    if constexpr (AMODE == A_LN) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_gamma + 32 * st + 8 * g + 4 * h);
                        const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_beta + 32 * st + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = (v[st][h][e] - mean) * rstd * gm[e] + bt[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = v[st][h][e];
                    }
                    pk[2 * h] = pack_bf16x2(o4[0], o4[1]);
                    pk[2 * h + 1] = pack_bf16x2(o4[2], o4[3]);
                }
                tf[tg][st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
            }
        }
    }

    const uint32_t w_frag = lds_addr(smem) + (uint32_t)swz128(pl, g);
    if constexpr (EPI == E_UNEMBED) {
        // patch_unembed: 64 column tiles (one patch pixel each), every tile reads 16 KB of `skip` and writes 16 KB per workgroup.
        // The `skip` lines of tile nt + 1 are requested at the top of tile nt: requested in their own tile they had one K loop
        // (0.4 us) to arrive, and every tile waited out the HBM latency with only the CU's second workgroup to cover it.  The
        // loads are asm (untracked), all waits counted: per tile and lane exactly 6 weight pieces, 4 skip loads, 4 stores (rows
        // outside the map load the map's first pixel / store to the sink).  Two operand sets, the tile loop unrolled by two.
        float bvec[16];
        gemm_load_bias(p, 0, g, true, bvec);                   // [64] base channels: the same for every tile
        size_t base[2];                                        // element offset of the patch origin (+ this lane's 16 channels)
        int py0[2], px0[2];
        bool tok_ok[2];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const int m = m0 + 32 * wave + 16 * tg + pl;
            const TokPos t = token_of_row(min(m, p.M - 1), p);
            tok_ok[tg] = t.valid && m < p.M;
            py0[tg] = t.ty * 8; px0[tg] = t.tx * 8;
            base[tg] = (((size_t)t.b * p.H) * p.W) * 64 + g * 16;
        }
        struct Ops { u32x4 a[2][2]; size_t off[2]; bool ok[2]; };
        auto issue = [&](int nt, Ops& o) {
            const int i = nt >> 3, j = nt & 7;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                const int py = py0[tg] + i, px = px0[tg] + j;
                o.ok[tg] = tok_ok[tg] && py < p.H && px < p.W;
                o.off[tg] = base[tg] + ((size_t)py * p.W + px) * 64;
                const bf16_t* sp = (p.skip ? p.skip : (const bf16_t*)p.out) + (o.ok[tg] ? o.off[tg] : (size_t)(g * 16));
                o.a[tg][0] = global_load_b128_async(sp);
                o.a[tg][1] = global_load_b128_async_16(sp);
            }
        };
        auto tile = [&](int nt, Ops& cur, Ops& nx) {
            // own pieces of W tile nt (requested a tile ago); younger: the 4 skip loads of tile nt, the 4 stores of tile nt-1
            if (nt > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone finished reading tile nt-1
            if (nt + 1 < ntiles) { dma_w(nt + 1, (nt + 1) & 1); issue(nt + 1, nx); }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2][4];
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint32_t wb = w_frag + (uint32_t)((nt & 1) * PW_BYTES);
            bf16x8 wf[3][4];
            auto ld = [&](int step, int slot) {
                const int kc = step >> 1;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    wf[slot][ct] = (step & 1) ? lds_read_b128_asm_off_x64(wb, kc * (64 * 128) + ct * 2048)
                                              : lds_read_b128_asm_off(wb, kc * (64 * 128) + ct * 2048);
            };
            ld(0, 0);
            ld(1, 1);
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int c3 = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<8>(); }
                else if (step + 1 < 6) { lds_wait<4>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = mfma16x16x32(wf[c3][ct], tf[tg][step], acc[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // this tile's skip lines, requested a tile ago; younger: the stores of tile nt-1 (4), W pieces (6) + skip loads (4) of nt+1
            if (nt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                asm volatile("" : "+v"(cur.a[tg][0]), "+v"(cur.a[tg][1]));      // the loads' outputs are live only from here
                uint32_t pk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    uint32_t sw = (q < 4) ? cur.a[tg][0][q & 3] : cur.a[tg][1][q & 3];
                    if (!p.skip) sw = 0u;
                    const int ct = q >> 1, e = (q & 1) * 2;
                    pk[q] = pack_bf16x2(acc[tg][ct][e] + bvec[2 * q] + __builtin_bit_cast(float, sw << 16),
                                        acc[tg][ct][e + 1] + bvec[2 * q + 1] + __builtin_bit_cast(float, sw & 0xffff0000u));
                }
                bf16_t* o = cur.ok[tg] ? (bf16_t*)p.out + cur.off[tg] : reinterpret_cast<bf16_t*>(tup_gemm_sink) + lane * 16;
                *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
        };
        Ops A, B;
        issue(0, A);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own pieces of W tile 0 (and tile 0's skip lines)
        for (int nt = 0; nt < ntiles; nt += 2) {
            tile(nt, A, B);
            if (nt + 1 < ntiles) tile(nt + 1, B, A);
        }
        return;
    }
    float cs[EPI == E_UNEMBED_MERGE ? 16 : 1] = {};            // E_UNEMBED_MERGE: column sums of `skip` (this lane's 16 channels)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // own pieces of W tile 0
    for (int nt = 0; nt < ntiles; ++nt) {
        __syncthreads();                                       // everyone's pieces of tile nt landed; everyone finished reading tile nt-1
        if (nt + 1 < ntiles) dma_w(nt + 1, (nt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int n0 = nt * 64;
        float bvec[16];
        gemm_load_bias(p, n0, g, EPI == E_UNEMBED, bvec);
        EpiPre pre[2];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) pre[tg] = epi_prefetch<EPI>(p, min(m0 + 32 * wave + 16 * tg + pl, p.M - 1), n0, g);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const uint32_t wb = w_frag + (uint32_t)((nt & 1) * PW_BYTES);
            bf16x8 wf[3][4];
            auto ld = [&](int step, int slot) {
                const int kc = step >> 1;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    wf[slot][ct] = (step & 1) ? lds_read_b128_asm_off_x64(wb, kc * (64 * 128) + ct * 2048)
                                              : lds_read_b128_asm_off(wb, kc * (64 * 128) + ct * 2048);
            };
            ld(0, 0);
            ld(1, 1);
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<8>(); }
                else if (step + 1 < 6) { lds_wait<4>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[tg][ct] = mfma16x16x32(wf[cur][ct], tf[tg][step], acc[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // Own pieces of W tile nt+1 (requested a K loop ago) and the epilogue operands: waited for HERE, before this
        // tile's stores are issued -- a vmcnt(0) at the top of the next iteration would also wait for those stores.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const int m = m0 + 32 * wave + 16 * tg + pl;
            if (m >= p.M) continue;
            float v[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[ct * 4 + e] = acc[tg][ct][e];
            epi_finish<EPI>(p, m, n0, g, v, bvec, pre[tg]);
            if constexpr (EPI == E_UNEMBED_MERGE) {
                if (pre[tg].ok) {          // every element of `skip` passes through exactly one lane here: its column sums ride along
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint32_t sw = (q < 4) ? pre[tg].a[0][q & 3] : pre[tg].a[1][q & 3];
                        cs[2 * q] += __builtin_bit_cast(float, sw << 16);
                        cs[2 * q + 1] += __builtin_bit_cast(float, sw & 0xffff0000u);
                    }
                }
            }
        }
    }
    if constexpr (EPI == E_UNEMBED_MERGE) {
        if (p.skip_colsum) {
            // lane (g, pl) holds channels 16 g .. 16 g + 15: sum over the 16 lanes of a group and the four waves through LDS (the
            // weight buffers are free now), then one atomic per channel onto this workgroup's replica (blockIdx & 15): atomics on
            // the same address serialise in L2, 16 replicas keep that to 60 per address
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int q = 0; q < 16; ++q) red[(wave * 64 + lane) * 16 + q] = cs[q];
            __syncthreads();
            if (tid < 64) {
                const int gg = tid >> 4, q = tid & 15;
                float t = 0.f;
                for (int w = 0; w < 4; ++w)
                    for (int l = 0; l < 16; ++l) t += red[(w * 64 + gg * 16 + l) * 16 + q];
                atomicAdd(p.skip_colsum + (blockIdx.x & 15) * 64 + tid, t);
            }
        }
    }
}

template <int AMODE, int EPI>
int launch_panel(const GemmParams& p, hipStream_t s)
{
    if (p.M <= 0) return 0;
    if (p.N % 64 != 0 || p.K != PK) return (int)hipErrorInvalidValue;
    static const bool use_v1 = TUP_ENV_FLAG("TUP_GEMM_PANEL_V1");
    if (!use_v1) {
        gemm_panel2_kernel<AMODE, EPI><<<dim3((p.M + PBM - 1) / PBM), dim3(256), 2 * PW_BYTES, s>>>(p);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    constexpr size_t lds = PA_BYTES + PW_BYTES;
    TUP_SET_DYN_LDS((gemm_panel_kernel<AMODE, EPI>), lds);
    gemm_panel_kernel<AMODE, EPI><<<dim3((p.M + PBM - 1) / PBM), dim3(256), lds, s>>>(p);
    TUP_CHECK_LAUNCH();
    return 0;
}

template <int AMODE, int EPI>
int launch(const GemmParams& p, hipStream_t s)
{
    if (p.M <= 0) return 0;
    if (p.N % BN != 0 || p.K % BK != 0) return (int)hipErrorInvalidValue;
    const int mt = (p.M + BM - 1) / BM;
    if (mt > 65535) return (int)hipErrorInvalidValue;
    gemm_tokens_kernel<AMODE, EPI><<<dim3(p.N / BN, mt), dim3(256), 2 * (A_STAGE + W_STAGE), s>>>(p);
    TUP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// patch_embed (K = 4096 gathered from the NHWC map) as its own kernel: one workgroup = 8 waves = 256 tokens x ALL N/64
// feature groups, so every 8x8 patch is fetched once (the generic 128 x 64 tile above reads it N/64 times through L2 and
// runs 16 MFMAs per barrier).  One K-step = one patch pixel (64 channels): token tile 32 KB + weight tile N x 128 B, both
// by LDS-DMA (swizzle and the reflect / zero padding on the source side) into a two-stage ring, one barrier per K-step;
// a wave owns 32 tokens x N: per quarter K-step 2 token + 6 (4) weight fragments for 12 (8) MFMAs, fragments two quarter
// steps ahead, immediate-offset reads.
// ------------------------------------------------------------------------------------------------
constexpr int PE_BM = 256, PE_A_BYTES = PE_BM * 128;
__device__ __attribute__((aligned(16))) unsigned int tup_pe_zero_line[4] = {0u, 0u, 0u, 0u};

// AMODE = A_PATCH (gathered patches) or A_BF16 (plain bf16 rows [M][lda]: the K = 576 / 768 Linear layers of the training
// path with N = 192), EPI = any epilogue of gemm_store_row.
template <int NT64, int AMODE, int EPI>
__global__ __launch_bounds__(512, 2) void patch_embed_kernel(const GemmParams p)
{
    constexpr int NTILE = 4 * NT64;                        // 16-row weight tiles
    constexpr int W_BYTES = NT64 * 64 * 128, STAGE = PE_A_BYTES + W_BYTES;
    constexpr int QN = NTILE / 2;                          // weight tiles per quarter step
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x STAGE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * PE_BM;
    const int nk = p.K / 64;                               // 64 patch pixels

    // ---- DMA bookkeeping: token-tile slot s = u*512 + tid -> row u*64 + (tid >> 3), logical chunk below ----
    const int dc = (tid & 7) ^ ((tid >> 4) & 7);
    uint32_t tokoff[4];                                    // byte offset of the patch origin (+ chunk) in the map
    int tpy[4], tpx[4];
    bool tval[4];
    const char* arow[4];                                   // A_BF16: this thread's four rows (+ chunk)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int m = min(m0 + u * 64 + (tid >> 3), p.M - 1);
        tval[u] = m0 + u * 64 + (tid >> 3) < p.M;
        if constexpr (AMODE == A_PATCH) {
            const TokPos t = token_of_row(m, p);
            tval[u] = tval[u] && t.valid;
            tpy[u] = t.ty * 8; tpx[u] = t.tx * 8;
            tokoff[u] = (uint32_t)(((size_t)t.b * p.H) * p.W * 128) + dc * 16;
        } else {
            arow[u] = (const char*)p.A + (size_t)m * p.lda * 2 + dc * 16;
        }
    }
    const bf16_t* w_thr = p.Wt + (size_t)(tid >> 3) * p.K + dc * 8;
    auto dma_stage = [&](int kc, int buf) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const int i = kc >> 3, j = kc & 7;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char* src;
            if constexpr (AMODE == A_PATCH) {
                int py = tpy[u] + i, px = tpx[u] + j;
                bool ok = tval[u];
                if (p.reflect) {
                    if (py >= p.H) py = 2 * p.H - 2 - py;
                    if (px >= p.W) px = 2 * p.W - 2 - px;
                } else if (py >= p.H || px >= p.W) {
                    ok = false;
                }
                src = ok ? (const char*)p.A + tokoff[u] + (size_t)(py * p.W + px) * 128 : (const char*)tup_pe_zero_line;
            } else {
                src = tval[u] ? arow[u] + (size_t)kc * 128 : (const char*)tup_pe_zero_line;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + u * 8192), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NT64; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_thr + (size_t)u * 64 * p.K + kc * 64),
                                             (__attribute__((address_space(3))) void*)(dst + PE_A_BYTES + u * 8192), 16, 0, 0);
    };
    dma_stage(0, 0);

    f32x4 acc[2][NTILE];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int n = 0; n < NTILE; ++n) acc[tg][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t sbase = lds_addr(smem);
    const uint32_t a_off0 = (uint32_t)swz128(32 * wave + pl, g), a_off1 = (uint32_t)swz128(32 * wave + 16 + pl, g);
    const uint32_t w_off = (uint32_t)(PE_A_BYTES + swz128(pl, g));

    for (int kc = 0; kc < nk; ++kc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own pieces of stage kc
        __syncthreads();                                       // everyone's; everyone finished reading stage kc-1
        if (kc + 1 < nk) dma_stage(kc + 1, (kc + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t sb = sbase + (uint32_t)((kc & 1) * STAGE);
        // quarter steps q = 0..3: K half kh = q >> 1, weight tiles (q & 1)*QN .. +QN
        bf16x8 tf[3][2], wf[3][QN];
        auto ld1 = [&](int q, int slot, int j) {               // j = 0, 1: token fragments; 2..: weight tiles of the quarter
            const int nb = (q & 1) * QN;
            if (j < 2) {
                const uint32_t a = sb + (j ? a_off1 : a_off0);
                tf[slot][j] = (q >> 1) ? lds_read_b128_asm_off_x64(a, 0) : lds_read_b128_asm_off(a, 0);
            } else {
                wf[slot][j - 2] = (q >> 1) ? lds_read_b128_asm_off_x64(sb + w_off, (nb + j - 2) * 2048)
                                           : lds_read_b128_asm_off(sb + w_off, (nb + j - 2) * 2048);
            }
        };
        constexpr int PER = 2 + QN, NM = 2 * QN;
#pragma unroll
        for (int j = 0; j < PER; ++j) ld1(0, 0, j);
#pragma unroll
        for (int j = 0; j < PER; ++j) ld1(1, 1, j);
        // the requests of quarter q + 2 are spread over quarter q's MFMAs (see conv_c64_persistent_kernel)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q % 3;
            if (q + 1 < 4) lds_wait<PER>(); else lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            const int nb = (q & 1) * QN;
            int rd = 0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (q + 2 < 4) {
#pragma unroll
                    for (int j = 0; j < PER; ++j)
                        if (j == rd && j * NM <= m * PER) { ld1(q + 2, (q + 2) % 3, j); ++rd; }
                }
                const int tg = m / QN, n = m % QN;
                acc[tg][nb + n] = mfma16x16x32(wf[cur][n], tf[cur][tg], acc[tg][nb + n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: per 64-feature group the lane holds features n0 + g*16 + ct*4 + e of token row m ----
#pragma unroll
    for (int ng = 0; ng < NT64; ++ng) {
        float bvec[16];
        gemm_load_bias(p, ng * 64, g, false, bvec);
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const int m = m0 + 32 * wave + 16 * tg + pl;
            if (m >= p.M) continue;
            float v[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[ct * 4 + e] = acc[tg][ng * 4 + ct][e];
            gemm_store_row<EPI>(p, m, ng * 64, g, v, bvec);
        }
    }
}

template <int NT64, int AMODE = A_PATCH, int EPI = E_PATCH_EMBED>
int launch_patch_embed(const GemmParams& p, hipStream_t s)
{
    if (p.M <= 0) return 0;
    constexpr size_t lds = 2 * (size_t)(PE_A_BYTES + NT64 * 64 * 128);
    TUP_SET_DYN_LDS((patch_embed_kernel<NT64, AMODE, EPI>), lds);
    patch_embed_kernel<NT64, AMODE, EPI><<<dim3((p.M + PE_BM - 1) / PE_BM), dim3(512), lds, s>>>(p);
    TUP_CHECK_LAUNCH();
    return 0;
}

int launch_patch_embed_any(const GemmParams& p, hipStream_t s)
{
    static const bool use_old = TUP_ENV_FLAG("TUP_PATCH_EMBED_V1");
    // the kernel keeps 32-bit byte offsets into the map: fine up to 4 GB of NHWC bf16 (B*H*W < 33.5 M pixels)
    const long long per_img = p.linear_tokens ? (long long)p.Ht * p.Wt_ : (long long)p.nWy * p.nWx * 64;
    const long long nimg = per_img > 0 ? p.M / per_img : 0;
    const bool fits = nimg * (long long)p.H * p.W * 128 < (1LL << 32);
    if (!use_old && fits && p.K == 4096) {
        if (p.N == 192) return launch_patch_embed<3>(p, s);
        if (p.N == 128) return launch_patch_embed<2>(p, s);
    }
    return launch<A_PATCH, E_PATCH_EMBED>(p, s);
}

}  // namespace

// epilogue: 0 = (+bias) -> bf16, 1 = +bias, erf-GELU -> bf16, 2 = +bias +res(fp32) -> fp32,
//           3 = * gelu'(aux) -> bf16 (aux = bf16 [M][ldo] pre-activation; backward of model.py:148).
//           With epilogue 1 a non-NULL aux is an OUTPUT: the bf16 pre-activation is saved there.
// a_dtype: 0 = bf16 A, 1 = fp32 A (converted to bf16 on the way into LDS).  bias may be NULL for 0 and 3.
// drop_p > 0 (epilogue 2 only): out = dropout(acc + bias) + res with the stateless mask of common.h.
extern "C" int tup_gemm_tokens_fwd(const void* A, int a_dtype, int lda, const void* Wt, const float* bias,
                                   const float* res, const void* aux, void* out, int ldo, int M, int N, int K,
                                   int epilogue, float drop_p, unsigned int drop_seed, void* stream)
{
    GemmParams p{};
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && epilogue != 2)) return (int)hipErrorInvalidValue;
    if (drop_p > 0.f) {
        p.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
        p.drop_inv_keep = 1.0f / (1.0f - drop_p);
        p.drop_seed = drop_seed;
    }
    p.A = A; p.lda = lda; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = out; p.ldo = ldo; p.res = res;
    p.skip = (const bf16_t*)aux;
    p.M = M; p.N = N; p.K = K;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((epilogue == 1 || epilogue == 2) && !bias) return (int)hipErrorInvalidValue;
    if (epilogue == 2 && !res) return (int)hipErrorInvalidValue;
    if (epilogue == 3 && !aux) return (int)hipErrorInvalidValue;
    static const bool use_panel = !TUP_ENV_FLAG("TUP_GEMM_NOPANEL");
    if (K == PK && use_panel) {
        if (a_dtype == 0) {
            if (epilogue == 0) return launch_panel<A_BF16, E_BF16>(p, s);
            if (epilogue == 1) return launch_panel<A_BF16, E_GELU_BF16>(p, s);
            if (epilogue == 2) return launch_panel<A_BF16, E_RES_F32>(p, s);
            if (epilogue == 3) return launch_panel<A_BF16, E_GELU_BWD>(p, s);
        } else if (a_dtype == 1) {
            if (epilogue == 0) return launch_panel<A_F32, E_BF16>(p, s);
            if (epilogue == 3) return launch_panel<A_F32, E_GELU_BWD>(p, s);
        }
        return (int)hipErrorInvalidValue;
    }
    static const bool use_big = !TUP_ENV_FLAG("TUP_GEMM_NOBIG");
    if (use_big && a_dtype == 0 && N == 192 && K > PK && K % 64 == 0 && M >= 4096) {       // long-K Linear layers onto the big tile
        if (epilogue == 0) return launch_patch_embed<3, A_BF16, E_BF16>(p, s);
        if (epilogue == 2) return launch_patch_embed<3, A_BF16, E_RES_F32>(p, s);
    }
    if (a_dtype == 0) {
        if (epilogue == 0) return launch<A_BF16, E_BF16>(p, s);
        if (epilogue == 1) return launch<A_BF16, E_GELU_BF16>(p, s);
        if (epilogue == 2) return launch<A_BF16, E_RES_F32>(p, s);
        if (epilogue == 3) return launch<A_BF16, E_GELU_BWD>(p, s);
    } else if (a_dtype == 1) {
        if (epilogue == 0) return launch<A_F32, E_BF16>(p, s);
        if (epilogue == 3) return launch<A_F32, E_GELU_BWD>(p, s);
    }
    return (int)hipErrorInvalidValue;
}

// feat: [B][H][W][64] bf16.  Wt: [192][4096] bf16, k = (i*8+j)*64 + c.  x_out: fp32 [B*nWy*nWx*64][192]
// in window layout (zero rows for padded tokens).
extern "C" int tup_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, float* x_out,
                                   int B, int H, int W, void* stream)
{
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = feat; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = x_out; p.ldo = 192;
    p.M = B * p.nWy * p.nWx * 64; p.N = 192; p.K = 4096; p.reflect = 1;
    // reflect padding needs pad < dim (same constraint as F.pad(mode='reflect'))
    if ((p.Ht * 8 - H) >= H || (p.Wt_ * 8 - W) >= W) return (int)hipErrorInvalidValue;
    return launch_patch_embed_any(p, reinterpret_cast<hipStream_t>(stream));
}

// x: fp32 [B*nWy*nWx*64][192] window layout.  Wt: [4096][192] bf16, n = (i*8+j)*64 + o (rows permuted
// per 64-group).  bias: [64].  skip / out: [B][H][W][64] bf16 NHWC; out = skip + unembed(x) + bias.
// x_bf16 != 0: x is bf16 [..][192] (tup_blocks_stream_fwd's out_bf16) -- the same values the fp32 form is rounded to on load.
extern "C" int tup_patch_unembed_fwd(const void* x, int x_bf16, const void* Wt, const float* bias, const void* skip,
                                     void* out, int B, int H, int W, void* stream)
{
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = x; p.lda = 192; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = out; p.skip = (const bf16_t*)skip;
    p.M = B * p.nWy * p.nWx * 64; p.N = 4096; p.K = 192;
    static const bool use_panel = !TUP_ENV_FLAG("TUP_GEMM_NOPANEL");
    if (x_bf16) return launch_panel<A_BF16, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
    if (use_panel) return launch_panel<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
    return launch<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
}

// LayerNorm fused into a Linear (norm1 -> attn.qkv, model.py:163 + :115): out bf16 [M][N] = LN(x) Wt^T + bias.
// x fp32 [M][192]; Wt bf16 [N][192] (rows permuted per 64-group).
extern "C" int tup_ln_gemm_fwd(const float* x, const float* gamma, const float* beta, const void* Wt,
                               const float* bias, void* out, int M, int N, void* stream)
{
    GemmParams p{};
    p.A = x; p.lda = 192; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = out; p.ldo = N;
    p.M = M; p.N = N; p.K = 192; p.ln_gamma = gamma; p.ln_beta = beta;
    return launch_panel<A_LN, E_BF16>(p, reinterpret_cast<hipStream_t>(stream));
}

// Backward of patch_unembed w.r.t. its input tokens (model.py:292-305 under autograd): gather the 8x8
// patches of the gradient map (zero beyond the cropped H x W) and multiply by W^T.
// gmap bf16 NHWC [B][H][W][64]; Wt bf16 [192][4096] (row k, col (i*8+j)*64+o, rows permuted per
// 64-group); gx fp32 window layout [M][192] (zero rows for padded tokens).
extern "C" int tup_patch_unembed_bwd(const void* gmap, const void* Wt, float* gx, int B, int H, int W, void* stream)
{
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = gmap; p.Wt = (const bf16_t*)Wt; p.bias = nullptr; p.out = gx; p.ldo = 192;
    p.M = B * p.nWy * p.nWx * 64; p.N = 192; p.K = 4096; p.reflect = 0;
    return launch_patch_embed_any(p, reinterpret_cast<hipStream_t>(stream));
}

// Backward of patch_embed w.r.t. the (reflect-padded) feature map (model.py:256-285 under autograd):
// gx fp32 window layout [M][192]; Wt bf16 [4096][192] (row (i*8+j)*64+c, rows permuted per 64-group);
// gmap_pad bf16 NHWC [B][Ht*8][Wt*8][64] -- the PADDED map; the caller folds the reflected rows/cols back.
extern "C" int tup_patch_embed_bwd(const float* gx, const void* Wt, void* gmap_pad, int B, int H, int W, void* stream)
{
    GemmParams p{};
    p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8; p.H = p.Ht * 8; p.W = p.Wt_ * 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = gx; p.lda = 192; p.Wt = (const bf16_t*)Wt; p.bias = nullptr; p.out = gmap_pad; p.skip = nullptr;
    p.M = B * p.nWy * p.nWx * 64; p.N = 4096; p.K = 192;
    static const bool use_panel = !TUP_ENV_FLAG("TUP_GEMM_NOPANEL");
    if (use_panel) return launch_panel<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
    return launch<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
}

// patch_embed's input gradient with the gradient merge at `feat` in its epilogue (training, H and W multiples of 8 so that the
// reflect-padded map IS the map): out bf16 NHWC [B][H][W][64] = (gx Wt^T + add1 + add2) * (relu_src > 0) -- the three gradient paths
// into `feat` (model.py:264 up branch, :268 patch_embed, :308 skip) and conv2's ReLU backward (model.py:252) without the separate
// tup_feat_grad_combine pass over five 64-channel maps.  add2 may be NULL.  add1_colsum (optional, fp32 [16][64], zeroed by the caller)
// += the per-channel sums of add1 in 16 replicas (the caller adds them up): add1 is the gradient entering patch_unembed's bias
// (model.py:302-308), and every element of it passes through this kernel once anyway.
extern "C" int tup_patch_embed_bwd_merge(const float* gx, const void* Wt, const void* add1, const void* add2, const void* relu_src,
                                         void* out, float* add1_colsum, int B, int H, int W, void* stream)
{
    if (H % 8 || W % 8 || add1 == nullptr || relu_src == nullptr) return (int)hipErrorInvalidValue;
    GemmParams p{};
    p.Ht = H / 8; p.Wt_ = W / 8; p.H = H; p.W = W;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = gx; p.lda = 192; p.Wt = (const bf16_t*)Wt; p.bias = nullptr; p.out = out;
    p.skip = (const bf16_t*)add1; p.skip2 = (const bf16_t*)add2; p.relu_src = (const bf16_t*)relu_src; p.skip_colsum = add1_colsum;
    p.M = B * p.nWy * p.nWx * 64; p.N = 4096; p.K = 192;
    return launch_panel<A_F32, E_UNEMBED_MERGE>(p, reinterpret_cast<hipStream_t>(stream));
}

// ---- ResidualTransformer token entry / exit (plain [B][45][80] token grid, no windows) ----
// feat bf16 NHWC [B][H][W][64] (H, W multiples of 8); Wt bf16 [128][4096]; pos fp32 [H/8*W/8][128]; x_out fp32 [B*T][128].
// patch_embed + flatten/transpose + pos_embed: models/ResidualTransformer/model.py:135-140.
extern "C" int tup_rt_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, const float* pos, float* x_out,
                                      int B, int H, int W, void* stream)
{
    if (H % 8 || W % 8) return (int)hipErrorInvalidValue;
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = H / 8; p.Wt_ = W / 8; p.linear_tokens = 1; p.pos = pos; p.reflect = 0;
    p.A = feat; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = x_out; p.ldo = 128;
    p.M = B * p.Ht * p.Wt_; p.N = 128; p.K = 4096;
    return launch_patch_embed_any(p, reinterpret_cast<hipStream_t>(stream));
}

// x fp32 [B*T][128]; Wt bf16 [4096][128] (n = (i*8+j)*64 + o); out = skip + ConvTranspose(k8,s8)(x) + bias, NHWC bf16.
// transpose/view + patch_unembed + skip add: model.py:147-153.
extern "C" int tup_rt_patch_unembed_fwd(const float* x, const void* Wt, const float* bias, const void* skip, void* out,
                                        int B, int H, int W, void* stream)
{
    if (H % 8 || W % 8) return (int)hipErrorInvalidValue;
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = H / 8; p.Wt_ = W / 8; p.linear_tokens = 1;
    p.A = x; p.lda = 128; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = out; p.skip = (const bf16_t*)skip;
    p.M = B * p.Ht * p.Wt_; p.N = 4096; p.K = 128;
    return launch<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
}

// ---- WindowTransformer token entry / exit (window layout, embedding width N = 128 or 192) ----
// patch_embed without reflect padding (models/WindowTransformer/model.py:247-264: a stride-8 conv drops the remainder
// rows / columns, the token grid is then zero-padded to whole windows): feat bf16 NHWC [B][H][W][64], Wt bf16 [N][4096],
// x_out fp32 [B*nWy*nWx*64][N] window layout, token grid floor(H/8) x floor(W/8).
extern "C" int tup_wt_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, float* x_out,
                                      int B, int H, int W, int N, void* stream)
{
    if (N % 64 || H < 8 || W < 8) return (int)hipErrorInvalidValue;
    GemmParams p{};
    p.H = H; p.W = W; p.Ht = H / 8; p.Wt_ = W / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8; p.reflect = 0;
    p.A = feat; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = x_out; p.ldo = N;
    p.M = B * p.nWy * p.nWx * 64; p.N = N; p.K = 4096;
    return launch_patch_embed_any(p, reinterpret_cast<hipStream_t>(stream));
}

// window_reverse + crop + patch_unembed + skip (model.py:272-291): x fp32 window layout [M][K], Wt bf16 [4096][K],
// skip / out bf16 NHWC [B][Ht*8][Wt*8][64] with Ht = Hs/8, Wt = Ws/8 given as the map size Hs x Ws (multiples of 8).
extern "C" int tup_wt_patch_unembed_fwd(const float* x, const void* Wt, const float* bias, const void* skip, void* out,
                                        int B, int Hs, int Ws, int K, void* stream)
{
    if (K % 64 || Hs % 8 || Ws % 8) return (int)hipErrorInvalidValue;
    GemmParams p{};
    p.H = Hs; p.W = Ws; p.Ht = Hs / 8; p.Wt_ = Ws / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8;
    p.A = x; p.lda = K; p.Wt = (const bf16_t*)Wt; p.bias = bias; p.out = out; p.skip = (const bf16_t*)skip;
    p.M = B * p.nWy * p.nWx * 64; p.N = 4096; p.K = K;
    return launch<A_F32, E_UNEMBED>(p, reinterpret_cast<hipStream_t>(stream));
}
