// Fused halves of a WindowTransformerBlock for inference (gfx950): the hidden / qkv / attention tensors
// never leave the CU; each kernel reads the fp32 residual stream once and writes it once.
//
//   tup_fused_mlp_fwd    x += mlp.2(GELU(mlp.0(LayerNorm2(x))))        models/FastTransformer/model.py:144-151,168-171
//
// One workgroup = 8 waves (two per SIMD, so one wave's LDS / barrier waits hide under the other's MFMAs)
// = 256 token rows, 32 per wave.  LN2(x) is built once in LDS (bf16, 48 KB) and is the MFMA
// token operand of all 12 hidden chunks; per chunk of 64 hidden units the 24 KB slice of W1 and the 24 KB
// slice of W2 are streamed global -> registers -> LDS (prefetched one chunk ahead); FC1's accumulators,
// after bias + erf-GELU, ARE the token operand of the FC2 partial product (the lane already holds the 16
// hidden units its MFMA lane group contracts over), accumulated in 96 registers per lane.  MFMA convention as everywhere: A operand = weight rows, B operand = token rows.
#include "common.h"

namespace {

constexpr int DIM = 192, HID = 768, BM = 256;
constexpr int A_BYTES = 3 * BM * 128;          // LN(x) as three [256][64] swizzled sub-tiles (96 KB)
constexpr int W1_BYTES = 3 * 64 * 128;         // W1 chunk: [64 hidden rows] x 3 k-tiles of 64
constexpr int W2_BYTES = 3 * 64 * 128;         // W2 chunk: 3 n-tiles of [64 out rows][64 k]

__global__ __launch_bounds__(512, 2) void fused_mlp_kernel(
    float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2,
    const float* __restrict__ b2, int M)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_lds = smem;
    char* w1_lds = smem + A_BYTES;
    char* w2_lds = w1_lds + W1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * BM;

    // ---- weight chunk prefetch: W1 rows j*64.. (64 x 192), W2 columns j*64.. of all 192 rows ----
    u32x4 wreg[6];
    auto load_w = [&](int j) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + u * 512;               // 1536 chunks: row = idx / 24, c24 = idx % 24
            const int row = idx / 24, c = idx - row * 24;
            wreg[u] = *reinterpret_cast<const u32x4*>(w1 + (size_t)(j * 64 + row) * DIM + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + u * 512;               // 1536 chunks: row = idx / 8 (0..191), c = idx % 8
            const int row = idx >> 3, c = idx & 7;
            wreg[3 + u] = *reinterpret_cast<const u32x4*>(w2 + (size_t)row * HID + j * 64 + c * 8);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + u * 512;
            const int row = idx / 24, c = idx - row * 24;
            *reinterpret_cast<u32x4*>(w1_lds + (c >> 3) * (64 * 128) + swz128(row, c & 7)) = wreg[u];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + u * 512;
            const int row = idx >> 3, c = idx & 7;
            *reinterpret_cast<u32x4*>(w2_lds + (row >> 6) * (64 * 128) + swz128(row & 63, c)) = wreg[3 + u];
        }
    };
    load_w(0);

    // ---- LayerNorm2 prologue: 16 lanes per row, 32 rows per pass ----
    {
        const int sub = tid & 15;
        f32x4 gm[3], bt[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            gm[q] = *reinterpret_cast<const f32x4*>(gamma + q * 64 + sub * 4);
            bt[q] = *reinterpret_cast<const f32x4*>(beta + q * 64 + sub * 4);
        }
#pragma unroll 2
        for (int pass = 0; pass < BM / 32; ++pass) {
            const int r = pass * 32 + (tid >> 4);
            const int m = min(m0 + r, M - 1);
            f32x4 v[3];
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                v[q] = *reinterpret_cast<const f32x4*>(x + (size_t)m * DIM + q * 64 + sub * 4);
                sum += v[q][0] + v[q][1] + v[q][2] + v[q][3];
            }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.0f / DIM);
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mean; ss += d * d; }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (v[q][e] - mean) * rstd * gm[q][e] + bt[q][e];
                *reinterpret_cast<u32x2*>(a_lds + q * (BM * 128) + swz128(r, sub >> 1) + (sub & 1) * 8) =
                    u32x2{pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3])};
            }
        }
    }

    // loop-invariant LDS byte addresses of this lane's fragments
    //   FC1: token rows 32*wave + 16*tg + pl, k-chunk kh*4+g of sub-tile kc; weight row ct*16+pl (kh = 1 is "^ 64")
    //   FC2: the token operand never touches LDS (see below); weight row n*16+pl, k-chunk 2g+s
    const uint32_t a_tok0 = lds_addr(a_lds) + (uint32_t)swz128(32 * wave + pl, g);
    const uint32_t a_tok1 = lds_addr(a_lds) + (uint32_t)swz128(32 * wave + 16 + pl, g);
    const uint32_t w1_frag = lds_addr(w1_lds) + (uint32_t)swz128(pl, g);
    const uint32_t w2_frag0 = lds_addr(w2_lds) + (uint32_t)swz128(pl, 2 * g);
    const uint32_t w2_frag1 = lds_addr(w2_lds) + (uint32_t)swz128(pl, 2 * g + 1);

    f32x4 acc2[2][12];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int n = 0; n < 12; ++n) acc2[tg][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < HID / 64; ++j) {
        store_w();
        __syncthreads();                       // W chunk j (and, first time, the LN tile) visible
        if (j + 1 < HID / 64) load_w(j + 1);

        // ---- FC1 chunk: [256][64] = LN(x) [256][192] . W1_j^T : 6 K-steps of (2 token + 4 weight) fragments
        // and 8 MFMAs, fragments of step k+1 requested before the MFMAs of step k (common.h) ----
        f32x4 acc1[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc1[tg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 w2f[12];
        {
            bf16x8 tf[2][2], wf[2][4];
            auto ld1 = [&](int step, int slot) {
                const int kc = step >> 1;
                const uint32_t khx = (step & 1) << 6;
                tf[slot][0] = lds_read_b128_asm((a_tok0 ^ khx) + kc * (BM * 128));
                tf[slot][1] = lds_read_b128_asm((a_tok1 ^ khx) + kc * (BM * 128));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) wf[slot][ct] = lds_read_b128_asm((w1_frag ^ khx) + kc * (64 * 128) + ct * 2048);
            };
            ld1(0, 0);
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step & 1;
                if (step + 1 < 6) {
                    ld1(step + 1, cur ^ 1);
                    lds_wait<6>();
                } else {
                    lds_wait<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc1[tg][ct] = mfma16x16x32(wf[cur][ct], tf[cur][tg], acc1[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- bias + erf-GELU in registers.  The lane holds hidden units 16g .. 16g+15 of its token (weight rows
        // are permuted that way), i.e. exactly two 8-wide k-groups: FC2 is run with the K order "lane group g <->
        // hidden 16g + 8s + j" (s = 0, 1), so its token operand is this lane's own packed values and its weight
        // fragment is k-chunk 2g+s of W2 -- the hidden tile is never written anywhere. ----
        // FC2's first 12 weight fragments are requested now and land under the GELU math
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 12; ++n) w2f[n] = lds_read_b128_asm(w2_frag0 + (n >> 2) * (64 * 128) + (n & 3) * 2048);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 hf[2][2];
        {
            f32x4 bb[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) bb[ct] = *reinterpret_cast<const f32x4*>(b1 + j * 64 + g * 16 + ct * 4);
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 pk;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ct = 2 * s + h;
                        const f32x2 g0 = gelu_erf2(f32x2{acc1[tg][ct][0] + bb[ct][0], acc1[tg][ct][1] + bb[ct][1]});
                        const f32x2 g1 = gelu_erf2(f32x2{acc1[tg][ct][2] + bb[ct][2], acc1[tg][ct][3] + bb[ct][3]});
                        pk[2 * h + 0] = pack_bf16x2(g0[0], g0[1]);
                        pk[2 * h + 1] = pack_bf16x2(g1[0], g1[1]);
                    }
                    hf[tg][s] = __builtin_bit_cast(bf16x8, pk);
                }
        }

        // ---- FC2 partial: [256][192] += hidden [256][64] . W2[:, chunk j]^T : 2 K-steps x 24 MFMAs ----
        {
            __builtin_amdgcn_sched_barrier(0);
            lds_wait<0>();                     // step 0 fragments (requested before the GELU) have landed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 12; ++n)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg) acc2[tg][n] = mfma16x16x32(w2f[n], hf[tg][0], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
            // step 1 reuses the fragment registers: the MFMAs above have issued (operands are read at issue),
            // the reads below return tens of cycles later
#pragma unroll
            for (int n = 0; n < 12; ++n) w2f[n] = lds_read_b128_asm(w2_frag1 + (n >> 2) * (64 * 128) + (n & 3) * 2048);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 12; ++n)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg) acc2[tg][n] = mfma16x16x32(w2f[n], hf[tg][1], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                       // everyone done with W chunk j before it is overwritten
    }

    // ---- epilogue: x = x + acc2 + b2 (lane holds features nt*64 + g*16 + ct*4 + e) ----
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const int m = m0 + 32 * wave + 16 * tg + pl;
        if (m >= M) continue;
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            const int col = (n >> 2) * 64 + g * 16 + (n & 3) * 4;
            float* xp = x + (size_t)m * DIM + col;
            const f32x4 rv = *reinterpret_cast<const f32x4*>(xp);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + col);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = acc2[tg][n][e] + bv[e] + rv[e];
            *reinterpret_cast<f32x4*>(xp) = ov;
        }
    }
}

}  // namespace

// x fp32 [M][192] updated in place: x += W2 GELU(W1 LN(x) + b1) + b2.  w1 bf16 [768][192], w2 bf16 [192][768]
// (both with rows permuted per 64-group, packing.pack_linear), biases / LayerNorm parameters fp32.
extern "C" int tup_fused_mlp_fwd(float* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                                 const void* w2, const float* b2, int M, void* stream)
{
    if (M <= 0) return 0;
    constexpr size_t lds = A_BYTES + W1_BYTES + W2_BYTES;      // 144 KB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    fused_mlp_kernel<<<dim3((M + BM - 1) / BM), dim3(512), lds, reinterpret_cast<hipStream_t>(stream)>>>(
        x, gamma, beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, M);
    TUP_CHECK_LAUNCH();
    return 0;
}
