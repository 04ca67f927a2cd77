// Fused halves of a WindowTransformerBlock for inference (gfx950): the hidden / qkv / attention tensors
// never leave the CU; each kernel reads the fp32 residual stream once and writes it once.
//
//   tup_fused_mlp_fwd    x += mlp.2(GELU(mlp.0(LayerNorm2(x))))        models/FastTransformer/model.py:144-151,168-171
//
// FC1's accumulators, after bias + erf-GELU, ARE the token operand of the FC2 partial product (the lane already holds
// the 16 hidden units its MFMA lane group contracts over), accumulated in 96 registers per lane.  MFMA convention as
// everywhere: A operand = weight rows, B operand = token rows.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int DIM = 192, HID = 768;
constexpr int W1_BYTES = 3 * 64 * 128;         // W1 chunk: [64 hidden rows] x 3 k-tiles of 64
constexpr int W2_BYTES = 3 * 64 * 128;         // W2 chunk: 3 n-tiles of [64 out rows][64 k]

// ------------------------------------------------------------------------------------------------
// The token operand never touches LDS.  Each lane loads its rows of x directly in MFMA B-fragment order
// (token 16tg+pl, channels 32*step + 8g .. +8), LayerNorm statistics are two cross-lane-group shuffles, and the
// 12 bf16x8 fragments of LN2(x) stay in registers for all 12 hidden chunks.  LDS holds only the weight stream,
// filled by LDS-DMA (global_load_lds_dwordx4, swizzle applied on the source side): W1 chunks double-buffered (the
// next one is requested at the top of a chunk), the W2 chunk single-buffered and requested at the same point --
// FC1 + GELU of the first half run while it lands.  75 KB per 4-wave workgroup of 128 rows, so TWO workgroups share
// a CU (one wave per SIMD each): they are not synchronised with each other, so one's x read / write-back and
// GELU (VALU) overlap the other's MFMAs -- with a single 8-wave workgroup per CU those phases ran in lockstep and
// simply added up (measured: 41 us of x write-back + 14 us of x read + 96 us of compute = 153 us).
// Each 64-wide hidden chunk is computed in two halves of 32 (FC1 -> bias + GELU -> FC2 K-step).
// ------------------------------------------------------------------------------------------------
constexpr int BM2 = 128;
__device__ unsigned long long tup_mlp_stamps[4][64];       // timing experiments (ABL & 64): s_memtime at phase boundaries
constexpr int V2_W2_OFF = 2 * W1_BYTES, V2_B1_OFF = 2 * W1_BYTES + W2_BYTES, V2_LDS = V2_B1_OFF + HID * 4;     // 75 KB

template <int ABL>      // ABL: timing ablations for scripts/microbench_tokens.py only (0 = the product kernel)
__global__ __launch_bounds__(256, 2) void fused_mlp_v2_kernel(
    float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2,
    const float* __restrict__ b2, int M)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];          // W1 x 2 | W2 | b1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * BM2;
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr ((ABL & 64) != 0) {
            const int slot = blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == 600 ? 2 : blockIdx.x == 959 ? 3 : -1;
            if (slot >= 0 && tid == 0 && nstamp < 64) tup_mlp_stamps[slot][nstamp] = __builtin_amdgcn_s_memtime();
            ++nstamp;
        }
    };
    stamp();
    // mlp.0's bias goes through LDS: a global load inside the chunk loop would tie its vmcnt wait to the weight DMA
    // issued around it (hipcc waits vmcnt(0) = for the whole next chunk) -- LDS reads count on lgkmcnt instead.
    float* b1_lds = reinterpret_cast<float*>(smem + V2_B1_OFF);
    for (int i = tid; i < HID / 4; i += 256) reinterpret_cast<f32x4*>(b1_lds)[i] = reinterpret_cast<const f32x4*>(b1)[i];

    // ---- weight chunk j -> LDS (DMA).  Physical 16-B slot s = u*256 + tid of a 24 KB image: tile s / 512
    // (W1: k-tile, W2: n-tile), row (s % 512) >> 3, physical chunk s & 7 ----
    // slot s = u*256 + tid -> tile u >> 1, row (u & 1)*32 + (tid >> 3), logical chunk (tid & 7) ^ ((tid >> 4) & 7): one
    // per-thread base address per matrix, the six pieces differ by compile-time offsets
    const int drow = tid >> 3, dc = (tid & 7) ^ ((tid >> 4) & 7);
    const bf16_t* w1_thr = w1 + (size_t)drow * DIM + dc * 8;
    const bf16_t* w2_thr = w2 + (size_t)drow * HID + dc * 8;
    auto dma_w1_piece = [&](int j, int buf, int u) {
        char* dst = smem + buf * W1_BYTES + wave * 1024;
        const bf16_t* src = w1_thr + (size_t)j * 64 * DIM;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * DIM + (u >> 1) * 64),
                                         (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    auto dma_w1 = [&](int j, int buf) {
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_w1_piece(j, buf, u);
    };
    auto dma_w2 = [&](int j) {
        char* dst = smem + V2_W2_OFF + wave * 1024;
        const bf16_t* src = w2_thr + j * 64;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)((u >> 1) * 64 + (u & 1) * 32) * HID),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    dma_w1(0, 0);

    // ---- LayerNorm2 straight into B fragments, residual straight into the FC2 accumulators ----
    // K-step st of FC1 contracts, in lane group g, over channels 64*(st>>1) + 16g + 8*(st&1) .. +8 (the columns of the
    // packed W1 are permuted to match, packing.pack_fc1_fused): these are exactly the features this lane's FC2
    // accumulators n = 2*st, 2*st+1 hold, so acc2 starts as x + b2 and the epilogue is a plain store -- the residual is
    // read once, not twice.
    bf16x8 tf[2][6];
    f32x4 acc2[2][12];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const int m = min(m0 + 32 * wave + 16 * tg + pl, M - 1);
        const float* xr = x + (size_t)m * DIM + 16 * g;
        f32x4 v[6][2];
        float sum = 0.f;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int c0 = 64 * (st >> 1) + 8 * (st & 1);
            if constexpr ((ABL & 16) != 0) { v[st][0] = f32x4{1.f * st, 2.f, 3.f, 4.f + g}; v[st][1] = f32x4{0.5f, 1.f * pl, 3.f, 4.f}; continue; }
            v[st][0] = *reinterpret_cast<const f32x4*>(xr + c0);
            v[st][1] = *reinterpret_cast<const f32x4*>(xr + c0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += v[st][0][e] + v[st][1][e];
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / DIM);
        float ss = 0.f;
#pragma unroll
        for (int st = 0; st < 6; ++st)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[st][h][e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            uint32_t pk[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 64 * (st >> 1) + 16 * g + 8 * (st & 1) + 4 * h;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + c);
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (v[st][h][e] - mean) * rstd * gm[e] + bt[e];
                pk[2 * h] = pack_bf16x2(o4[0], o4[1]);
                pk[2 * h + 1] = pack_bf16x2(o4[2], o4[3]);
                acc2[tg][2 * st + h] = v[st][h] + bv;
            }
            tf[tg][st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
        }
    }

    stamp();                   // 1: prologue done
    const uint32_t sbase = lds_addr(smem);
    const uint32_t w1_off = (uint32_t)swz128(pl, g);
    const uint32_t w2_off0 = (uint32_t)(V2_W2_OFF + swz128(pl, 2 * g)), w2_off1 = (uint32_t)(V2_W2_OFF + swz128(pl, 2 * g + 1));
    const uint32_t b1_base = lds_addr(b1_lds) + (uint32_t)(g * 64);

    for (int j = 0; j < HID / 64; ++j) {
        if (j < 3) stamp();        // chunk top
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of W1 chunk j have landed
        if (j < 3) stamp();        // after own vmcnt
        __syncthreads();                                       // everyone's have; everyone is done with chunk j-1 (W2 buffer free)
        if (j < 3) stamp();        // after barrier
        // W2 chunk j now (needed first); the six pieces of W1 chunk j+1 are issued one per K-step of the first FC1 half
        // below, in the shadow of its MFMAs (issued in a block here they held the wave for ~60 cycles each, stamped)
        dma_w2(j);
        const bool more = j + 1 < HID / 64;
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t wb1 = sbase + (uint32_t)((j & 1) * W1_BYTES);

#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // ---- FC1 half: hidden rows ct = 2s, 2s+1 of the chunk; weight fragments two K-steps ahead ----
            f32x4 acc1[2][2];
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc1[tg][h] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 bb[2];           // this half's bias: hidden j*64 + g*16 + (2s+h)*4 .. +4
#pragma unroll
            for (int h = 0; h < 2; ++h)
                bb[h] = __builtin_bit_cast(f32x4, lds_read_b128_asm(b1_base + (uint32_t)((j * 64 + (2 * s + h) * 4) * 4)));
            bf16x8 wf[3][2];
            auto ld1 = [&](int step, int slot) {
                const uint32_t a = wb1 + ((w1_off ^ ((uint32_t)(step & 1) << 6)) + (uint32_t)((step >> 1) * (64 * 128) + 2 * s * 2048));
                wf[slot][0] = lds_read_b128_asm(a);
                wf[slot][1] = lds_read_b128_asm(a + 2048);
            };
            __builtin_amdgcn_sched_barrier(0);
            ld1(0, 0);
            ld1(1, 1);
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld1(step + 2, (step + 2) % 3); lds_wait<4>(); }
                else if (step + 1 < 6) lds_wait<2>();
                else lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if ((ABL & 2) == 0 || step == 0) acc1[tg][h] = mfma16x16x32(wf[cur][h], tf[tg][step], step == 0 ? bb[h] : acc1[tg][h]);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0 && more) { dma_w1_piece(j + 1, (j + 1) & 1, step); __builtin_amdgcn_sched_barrier(0); }
            }
            if (j < 3) stamp();    // FC1 half done
            // ---- FC2 weight fragments (first 6 n tiles) land under the GELU math ----
            if (s == 0) {          // W2 chunk j was requested at the top of the chunk: own pieces, then everyone's
                // all but the youngest 6 (= the W1 prefetch, if any) have landed; raw barrier: __syncthreads() would make
                // hipcc wait vmcnt(0), i.e. for the W1 pieces issued a moment ago
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (j < 3) stamp();    // W2 barrier passed
            }
            // FC2 weight fragments: 8 of the 12 are requested now and land under the GELU math (with 2 MFMAs per
            // fragment the ~400-cycle LDS latency, stamped, cannot be hidden by a short ring); the last 4 reuse the
            // first 4 registers once those MFMAs have issued
            bf16x8 w2f[8];
            const uint32_t w2a = sbase + (s ? w2_off1 : w2_off0);
            auto w2addr = [&](int n) { return w2a + (uint32_t)((n >> 2) * (64 * 128) + (n & 3) * 2048); };
#pragma unroll
            for (int n = 0; n < 8; ++n) w2f[n] = lds_read_b128_asm(w2addr(n));
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 hf[2];
            gelu16_fragments<2>(acc1, hf);          // fp16: gelu(x) / 4; acc1 already holds x / 4 incl. the bias (common.h)
            __builtin_amdgcn_sched_barrier(0);
            lds_wait<4>();
            if (j < 3) stamp();    // GELU done
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
                    if ((ABL & 4) == 0 || n == 0) acc2[tg][n] = mfma16x16x32_f16(w2f[n], hf[tg], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n) w2f[n] = lds_read_b128_asm(w2addr(8 + n));      // operands above were read at issue
            lds_wait<4>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 4; n < 8; ++n)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
                    if ((ABL & 4) == 0) acc2[tg][n] = mfma16x16x32_f16(w2f[n], hf[tg], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
                    if ((ABL & 4) == 0) acc2[tg][8 + n] = mfma16x16x32_f16(w2f[n], hf[tg], acc2[tg][8 + n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    stamp();                   // loop done
    // ---- epilogue: acc2 already holds x + b2 + FC2 (lane holds features (n>>2)*64 + g*16 + (n&3)*4 + e) ----
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const int m = m0 + 32 * wave + 16 * tg + pl;
        if (m >= M) continue;
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            const int col = (n >> 2) * 64 + g * 16 + (n & 3) * 4;
            float* xp = x + (size_t)m * DIM + col;
            if constexpr ((ABL & 32) != 0) { if (acc2[tg][n][0] == 123.456f) *xp = 1.f; continue; }
            *reinterpret_cast<f32x4*>(xp) = acc2[tg][n];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();                   // stores retired
}

}  // namespace

// x fp32 [M][192] updated in place: x += W2 GELU(W1 LN(x) + b1) + b2.  w1 bf16 [768][192] = mlp.0.weight / 4 and b1 = mlp.0.bias / 4
// (packing.pack_fc1_fused_q), w2 FP16 [192][768] = 4 mlp.2.weight (packing.pack_fc2_h4): the GELU runs in packed fp16 on x / 4
// and FC2 on the fp16 MFMA (common.h, gelu16_batch); rows permuted per 64-group, other biases / LayerNorm parameters fp32.
extern "C" int tup_fused_mlp_fwd(float* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                                 const void* w2, const float* b2, int M, void* stream)
{
    if (M <= 0) return 0;
    {
        constexpr size_t lds2 = V2_LDS;
        TUP_SET_DYN_LDS((fused_mlp_v2_kernel<0>), lds2);
        const dim3 grid((M + BM2 - 1) / BM2);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef TUP_DIAG          // `make diag` only: the timing ablations are not instantiated in the product library
        const char* abl = getenv("TUP_MLP_ABLATE");            // timing experiments only (results are wrong)
        if (abl) {
#define TUP_ABL_CASE(V) case V: { (void)hipFuncSetAttribute((const void*)fused_mlp_v2_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
            fused_mlp_v2_kernel<V><<<grid, dim3(256), lds2, st>>>(x, gamma, beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, M); break; }
            switch (atoi(abl)) { TUP_ABL_CASE(1) TUP_ABL_CASE(6) TUP_ABL_CASE(7) TUP_ABL_CASE(48) TUP_ABL_CASE(55) TUP_ABL_CASE(64) default: return (int)hipErrorInvalidValue; }
#undef TUP_ABL_CASE
            TUP_CHECK_LAUNCH();
            return 0;
        }
#endif
        fused_mlp_v2_kernel<0><<<grid, dim3(256), lds2, st>>>(x, gamma, beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, M);
        TUP_CHECK_LAUNCH();
        return 0;
    }
}

#ifdef TUP_DIAG
// Timing experiments only (`make diag`): copies the s_memtime stamps of the last TUP_MLP_ABLATE=64 launch (4 workgroups x 64).
extern "C" int tup_debug_mlp_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tup_mlp_stamps), sizeof(unsigned long long) * 4 * 64);
}
#endif
