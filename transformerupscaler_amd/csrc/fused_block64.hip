// One whole WindowTransformerBlock per launch, ONE WAVE = ONE 8x8 WINDOW (64 tokens = four 16-token MFMA tiles), gfx950.
//     x += proj(WindowAttention(LayerNorm1(x)));  x += mlp.2(GELU(mlp.0(LayerNorm2(x))))        models/FastTransformer/model.py:104-133,153-172
//
// Successor of fused_attn.hip's fused_qkv_attn_kernel<true,true> (32 tokens per wave, two waves per window).  Measured leads
// of round 1 applied:
//   * 64 token columns per wave: every weight fragment read from LDS feeds 4 MFMAs instead of 2, and the weight stream of a
//     workgroup (LDS-DMA, L2 -> LDS) is shared by 256 tokens instead of 128;
//   * the whole window lives in one wave, so K and V never leave the register file: the qkv accumulators (rows = head channel,
//     column = token) are the B operand of q AND the A operand of K for S^T = K Q^T as they stand, and V^T comes out of the
//     qkv GEMM in the A-operand layout of O^T = V^T P^T by swapping that product's operands (tokens as rows).  The two 2 KB
//     LDS tiles per head, their ds_read_b64_tr_b16 read-back and the cross-wave hand-off are gone; the one barrier per head
//     that remains only recycles the shared weight slot;
//   * one 256-thread workgroup per CU with the whole 512-register file per wave (acc2 = the fp32 residual stream of 64 tokens
//     alone is 192 registers).
// Weight packing, LDS slot geometry, fragment addressing and the arithmetic order per token are those of fused_attn.hip: the
// two kernels agree bit for bit (tests/test_hip_kernels.py::test_block64_equals_block32).
#include "common.h"
#include <string.h>

namespace {

constexpr int DIM = 192, HEADS = 12, HD = 16, HID = 768, TG = 4;
constexpr int FW_BYTES = 3 * 64 * 128;                     // one weight slot: 3 k-tiles x [64 rows][128 B]
constexpr int W2_OFF = 2 * FW_BYTES;                       // mlp.2 chunk [3 k... see dma_w2]
constexpr int B1_OFF = 3 * FW_BYTES;                       // mlp.0 bias fp32 [768]
constexpr int QB_OFF = B1_OFF + HID * 4;                   // qkv bias fp32 [12][48]
constexpr int B64_LDS = QB_OFF + HEADS * 48 * 4;           // 79,104 B

struct Mlp64Args {
    const float* gamma2; const float* beta2;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
};
// parameters of one WindowTransformerBlock (device pointers), the element type of the table tup_fused_blocks64_fwd takes
struct Block64Ptrs {
    const float* gamma1; const float* beta1; const bf16_t* wh; const float* bh; const float* bias_frag;
    const bf16_t* wproj; const float* bproj; Mlp64Args ma;
};
static_assert(sizeof(Block64Ptrs) == 13 * 8, "table row = 13 device pointers");
constexpr int MAX_BLK = 8;
// passed BY VALUE: pointer fields of a kernel argument are global pointers to hipcc, pointers read from a table in memory are
// generic (every load through them a flat_load, which counts on lgkmcnt AND vmcnt and drains the LDS-DMA queue at each use)
struct Block64Table { Block64Ptrs b[MAX_BLK]; };

template <class T> TUP_DEVICE const T* as_global(const T* p) {
    return (const T*)(const __attribute__((address_space(1))) T*)p;
}
TUP_DEVICE bf16x8 join4(s16x4 lo, s16x4 hi) {
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}
TUP_DEVICE s16x4 pack4(f32x4 v) {
    return __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
}

// Diagnostic build only (STAMPS = true, TUP_B64_STAMPS=1): s_memtime at phase boundaries, summed per phase kind, for 8 recorded
// workgroups (4 of the first dispatch round, 4 of the second); the values leave through a buffer nothing else reads.
constexpr int NPH = 16;
__device__ unsigned long long tup_b64_stamps[8][4][NPH];
TUP_DEVICE unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
enum { PH_LN1 = 0, PH_SYNC0, PH_QKV, PH_HBAR, PH_ATT, PH_PROJ, PH_LN2, PH_MTOP, PH_FC1, PH_W2BAR, PH_GELU, PH_FC2, PH_STORE, PH_TOTAL };

template <bool STAMPS>
__global__ __launch_bounds__(256, 1) void fused_block64_kernel(
    float* __restrict__ xio, const Block64Table tbl, int nblk, int nwin)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int win = blockIdx.x * 4 + wave;
    const bool active = win < nwin;
    const int row0 = (active ? win : nwin - 1) * 64;                    // first of this wave's 64 token rows
    unsigned long long ph[NPH] = {}, tprev = 0, tstart = 0;
    if constexpr (STAMPS) tprev = tstart = stamp_now();
#define B64_STAMP(K) do { if constexpr (STAMPS) { const unsigned long long t_ = stamp_now(); ph[K] += t_ - tprev; tprev = t_; } } while (0)

    // All blocks of the model in ONE launch: a window never interacts with another window (the partition is the same in every
    // block, model.py:283-289), so this wave carries its window through blk = 0 .. nblk-1; x goes through memory between
    // blocks only as this wave's own store followed by its own loads (served by L2), and HBM sees x once in, once out.
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
    const Block64Ptrs& bp = tbl.b[blk];
    const float* __restrict__ gamma = bp.gamma1; const float* __restrict__ beta = bp.beta1;
    const bf16_t* __restrict__ wh = bp.wh; const float* __restrict__ bh = bp.bh; const float* __restrict__ bias_frag = bp.bias_frag;
    const bf16_t* __restrict__ wproj = bp.wproj; const float* __restrict__ bproj = bp.bproj;
    const Mlp64Args ma = bp.ma;
    if (blk) __syncthreads();          // everyone is done with the previous block's LDS (weight slots, W2 chunk, biases)

    float* qb = reinterpret_cast<float*>(smem + QB_OFF);
    for (int i = tid; i < HEADS * 48; i += 256) qb[i] = bh[i];
    {
        float* b1s = reinterpret_cast<float*>(smem + B1_OFF);
        for (int i = tid; i < HID / 4; i += 256) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(ma.b1)[i];
    }

    // weight slot by DMA: piece u of thread tid -> k-tile u >> 1, row (u & 1)*32 + (tid >> 3), logical chunk
    // (tid & 7) ^ ((tid >> 4) & 7) (swizzle on the source side)
    const int thr_off = (tid >> 3) * DIM + ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    auto dma_rows = [&](const bf16_t* base, int buf) {                   // 64 rows x 192 columns of a [.][192] matrix
        char* dst = smem + buf * FW_BYTES + wave * 1024;
        const bf16_t* src = base + thr_off;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * DIM + (u >> 1) * 64),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    auto dma_w = [&](int h, int buf) { dma_rows(wh + (size_t)h * 64 * DIM, buf); };
    auto dma_wp = [&](int chunk, int buf) { dma_rows(wproj + (size_t)chunk * 64 * DIM, buf); };
    auto dma_w1_piece = [&](int j, int buf, int u) {
        char* dst = smem + buf * FW_BYTES + wave * 1024;
        const bf16_t* src = ma.w1 + (size_t)j * 64 * DIM + thr_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * DIM + (u >> 1) * 64),
                                         (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    const bf16_t* w2_thr = ma.w2 + (size_t)(tid >> 3) * HID + ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    auto dma_w2 = [&](int j) {
        char* dst = smem + W2_OFF + wave * 1024;
        const bf16_t* src = w2_thr + j * 64;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)((u >> 1) * 64 + (u & 1) * 32) * HID),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    dma_w(0, 0);
    dma_w(1, 1);

    // ---- LayerNorm1 straight into B fragments: token 16tg+pl, channels 32*st + 8g .. +8 ----
    bf16x8 tf[TG][6];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
        const float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM + 8 * g;
        f32x4 v[6][2];
        float sum = 0.f;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            v[st][0] = *reinterpret_cast<const f32x4*>(xr + 32 * st);
            v[st][1] = *reinterpret_cast<const f32x4*>(xr + 32 * st + 4);
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
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[st][hh][e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            uint32_t pk[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 32 * st + 8 * g + 4 * hh);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 32 * st + 8 * g + 4 * hh);
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (v[st][hh][e] - mean) * rstd * gm[e] + bt[e];
                pk[2 * hh] = pack_bf16x2(o4[0], o4[1]);
                pk[2 * hh + 1] = pack_bf16x2(o4[2], o4[3]);
            }
            tf[tg][st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
        }
    }

    const uint32_t sbase = lds_addr(smem);
    const uint32_t w_off = (uint32_t)swz128(pl, g);
    const uint32_t qb_addr = sbase + QB_OFF + (uint32_t)(4 * g) * 4;
    B64_STAMP(PH_LN1);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // own pieces of heads 0 / 1; bias staging visible below
    __syncthreads();
    // v bias of every head for this lane's channel column (the V product is computed transposed: bias runs along the lanes)
    float vb[HEADS];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) vb[h] = qb[h * 48 + 32 + pl];
    B64_STAMP(PH_SYNC0);

    // the attention output of this wave's 64 tokens, all heads: of[tg][h] = O^T tile (channels 4g.., token pl) as bf16x4
    s16x4 of[TG][HEADS];

    auto head = [&](const int h) {
        // relative position bias of the (key tile, query tile) pairs, requested BEFORE the DMA below so that the compiler's
        // wait for them counts the DMA pieces instead of draining them
        f32x4 rb[TG][4];
#pragma unroll
        for (int tg = 0; tg < TG; ++tg)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                rb[tg][kt] = *reinterpret_cast<const f32x4*>(bias_frag + ((((size_t)h * 4 + kt) * 4 + tg) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);

        // ---- q^T, k^T [16 ch][64 tok] = W_h LN(x)^T (rows = head channel 4g+e, column = token pl);
        //      v [64 tok][16 ch] = LN(x) W_v^T (rows = token 4g+e of the tile, column = channel pl) ----
        const uint32_t wb = sbase + (uint32_t)((h & 1) * FW_BYTES) + w_off;
        f32x4 acc[TG][3];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            acc[0][ct] = __builtin_bit_cast(f32x4, lds_read_b128_asm_off(qb_addr, (h * 48 + ct * 16) * 4));    // immediate offsets: a
            // per-head address register would be hoisted out of the block loop, spilled, and reloaded behind a vmcnt(0)
        {
            bf16x8 wf[3][3];
            auto ld = [&](int step, int slot) {
                const int kc = step >> 1;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    wf[slot][ct] = (step & 1) ? lds_read_b128_asm_off_x64(wb, kc * (64 * 128) + ct * 2048)
                                              : lds_read_b128_asm_off(wb, kc * (64 * 128) + ct * 2048);
            };
            ld(0, 0);
            ld(1, 1);
            lds_wait<6>();             // the two bias reads have landed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                acc[tg][0] = acc[0][0];
                acc[tg][1] = acc[0][1];
                acc[tg][2] = f32x4{vb[h], vb[h], vb[h], vb[h]};
            }
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<6>(); }
                else if (step + 1 < 6) { lds_wait<3>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < TG; ++tg) {
                    acc[tg][0] = mfma16x16x32(wf[cur][0], tf[tg][step], acc[tg][0]);
                    acc[tg][1] = mfma16x16x32(wf[cur][1], tf[tg][step], acc[tg][1]);
                    acc[tg][2] = mfma16x16x32(tf[tg][step], wf[cur][2], acc[tg][2]);        // operands swapped: V, not V^T
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        B64_STAMP(PH_QKV);
        // ONE barrier per head: it only recycles the weight slot.  Before it every wave waits for its own pieces of the NEXT
        // head's weights (requested a head ago; younger than them: the 16 bias loads above), so passing it means the next
        // head's weights have landed everywhere and everyone is done reading this head's slot, which is refilled two ahead.
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (h + 2 < HEADS) dma_w(h + 2, h & 1);
        else dma_wp(h + 2 - HEADS, h & 1);                      // proj chunks 0 and 1 take the place of "heads 12 and 13"
        __builtin_amdgcn_sched_barrier(0);
        B64_STAMP(PH_HBAR);

        s16x4 kf[TG], vf[TG];
#pragma unroll
        for (int kt = 0; kt < TG; ++kt) { kf[kt] = pack4(acc[kt][1]); vf[kt] = pack4(acc[kt][2]); }
        // ---- S^T = K Q^T + bias, softmax over keys, O^T = V^T P^T; the four query tiles advance in lockstep so that their
        // (independent) MFMA -> max -> cross-lane -> exp -> sum chains cover each other's latencies ----
        f32x4 st[TG][4];
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            const s16x4 qf = pack4(acc[tg][0] * 0.25f);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[tg][kt] = mfma16x16x16(kf[kt], qf, rb[tg][kt]);      // bias as the accumulator input
        }
        float mx[TG], sum[TG];
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            float m = fmaxf(fmaxf(st[tg][0][0], st[tg][0][1]), fmaxf(st[tg][0][2], st[tg][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) m = fmaxf(fmaxf(m, st[tg][kt][0]), fmaxf(st[tg][kt][1], fmaxf(st[tg][kt][2], st[tg][kt][3])));
            mx[tg] = m;
        }
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) mx[tg] = rows_max(mx[tg]) * 1.4426950408889634f;
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            float sm = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[tg][kt][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[tg][kt][e], 1.4426950408889634f, -mx[tg]));
                    sm += st[tg][kt][e];
                }
            sum[tg] = sm;
        }
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) sum[tg] = rows_sum(sum[tg]);
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)        // P enters the product unnormalised (<= 1); O is scaled by 1 / sum in fp32
                o = mfma16x16x32(join4(vf[2 * kp], vf[2 * kp + 1]), join4(pack4(st[tg][2 * kp]), pack4(st[tg][2 * kp + 1])), o);
            of[tg][h] = pack4(o * __builtin_amdgcn_rcpf(sum[tg]));
        }
        B64_STAMP(PH_ATT);
    };
#pragma unroll
    for (int h = 0; h < HEADS; ++h) head(h);               // unrolled: of[tg][h] must be a register, not an indexed array

    // ---- x += att W_proj^T + b: K-step p = heads (2p, 2p+1); the packed weight has its columns ordered to match join4's
    // k map (packing.pack_proj_pairs) ----
    f32x4 acc2[TG][12];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg)
#pragma unroll
        for (int n = 0; n < 12; ++n) acc2[tg][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of chunk c
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // everyone's pieces; everyone finished chunk c-1
        if (c == 1) dma_wp(2, 0);
        if (c == 2) {                                      // slot 1 is free: mlp.0's first chunk lands under the last proj chunk
#pragma unroll
            for (int u = 0; u < 6; ++u) dma_w1_piece(0, 1, u);
        }
        const uint32_t wb = sbase + (uint32_t)((c & 1) * FW_BYTES) + w_off;
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
            for (int tg = 0; tg < TG; ++tg) {
                const bf16x8 tfp = join4(of[tg][2 * step], of[tg][2 * step + 1]);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc2[tg][4 * c + ct] = mfma16x16x32(wf[cur][ct], tfp, acc2[tg][4 * c + ct]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    B64_STAMP(PH_PROJ);
    // ---- second half of the block.  acc2 <- the new residual stream x + proj + b_proj (inactive waves carry a copy of the
    // last window and take part in every barrier; only their final store is skipped) ----
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
        const float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM;
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            const int col = (n >> 2) * 64 + g * 16 + (n & 3) * 4;       // weight rows are permuted per 64-group
            const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + col);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bproj + col);
            acc2[tg][n] = (acc2[tg][n] + bv) + rv;
        }
    }
    // LayerNorm2 from the accumulators: this lane holds 48 of its token's 192 channels, the other three lane groups the
    // rest.  K-step st of FC1 contracts over channels 64*(st>>1) + 16g + 8*(st&1) .. +8 = accumulators n = 2st, 2st+1
    // (packing.pack_fc1_fused), so the B fragments are packed straight from them.
    bf16x8 tf2[TG][6];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
        float sum = 0.f;             // same summation order as fused_mlp_v2_kernel / fused_attn.hip: bit-for-bit agreement
#pragma unroll
        for (int st = 0; st < 6; ++st)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc2[tg][2 * st][e] + acc2[tg][2 * st + 1][e];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / DIM);
        float ss = 0.f;
#pragma unroll
        for (int n = 0; n < 12; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc2[tg][n][e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            uint32_t pk[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int c = 64 * (st >> 1) + 16 * g + 8 * (st & 1) + 4 * hh;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(ma.gamma2 + c);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(ma.beta2 + c);
                const f32x4 b2v = *reinterpret_cast<const f32x4*>(ma.b2 + c);
                const f32x4 v = acc2[tg][2 * st + hh];
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
                pk[2 * hh] = pack_bf16x2(o4[0], o4[1]);
                pk[2 * hh + 1] = pack_bf16x2(o4[2], o4[3]);
                acc2[tg][2 * st + hh] = v + b2v;               // FC2 accumulates onto x + b2
            }
            tf2[tg][st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
        }
    }

    B64_STAMP(PH_LN2);
    // ---- MLP chunk loop: 64 hidden units per chunk j (two 32-unit halves s); W1 chunk j lives in slot (j + 1) & 1 ----
    const uint32_t w2_off0 = (uint32_t)(W2_OFF + swz128(pl, 2 * g)), w2_off1 = (uint32_t)(W2_OFF + swz128(pl, 2 * g + 1));
    const uint32_t b1_base = sbase + B1_OFF + (uint32_t)(g * 64);
    for (int j = 0; j < HID / 64; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of W1 chunk j have landed
        __syncthreads();                                       // everyone's have; everyone is done with chunk j-1 / the proj
        dma_w2(j);
        const bool more = j + 1 < HID / 64;
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t wb1 = sbase + (uint32_t)(((j + 1) & 1) * FW_BYTES);
        B64_STAMP(PH_MTOP);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 acc1[TG][2];
#pragma unroll
            for (int tg = 0; tg < TG; ++tg)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) acc1[tg][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 bb[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                bb[hh] = __builtin_bit_cast(f32x4, lds_read_b128_asm_off(b1_base + (uint32_t)(j * 256), (2 * s + hh) * 16));
            bf16x8 wf[3][2];
            const uint32_t wb1a = wb1 + w_off, wb1b = wb1 + (w_off ^ 64u);
            auto ld1 = [&](int step, int slot) {
                const uint32_t a = (step & 1) ? wb1b : wb1a;
                wf[slot][0] = lds_read_b128_asm_off(a, (step >> 1) * (64 * 128) + 2 * s * 2048);
                wf[slot][1] = lds_read_b128_asm_off(a, (step >> 1) * (64 * 128) + 2 * s * 2048 + 2048);
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
                for (int tg = 0; tg < TG; ++tg)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) acc1[tg][hh] = mfma16x16x32(wf[cur][hh], tf2[tg][step], acc1[tg][hh]);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0 && more) { dma_w1_piece(j + 1, j & 1, step); __builtin_amdgcn_sched_barrier(0); }
            }
            B64_STAMP(PH_FC1);
            if (s == 0) {
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // W2 chunk j landed (younger: the W1 prefetch)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            B64_STAMP(PH_W2BAR);
            bf16x8 w2f[8];
            const uint32_t w2a = sbase + (s ? w2_off1 : w2_off0);
#pragma unroll
            for (int n = 0; n < 8; ++n) w2f[n] = lds_read_b128_asm_off(w2a, (n >> 2) * (64 * 128) + (n & 3) * 2048);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 hfr[TG];
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                f32x2 gv[4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    gv[hh * 2 + 0] = f32x2{acc1[tg][hh][0] + bb[hh][0], acc1[tg][hh][1] + bb[hh][1]};
                    gv[hh * 2 + 1] = f32x2{acc1[tg][hh][2] + bb[hh][2], acc1[tg][hh][3] + bb[hh][3]};
                }
                gelu_erf2_batch<4>(gv);
                u32x4 pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf16x2(gv[q][0], gv[q][1]);
                hfr[tg] = __builtin_bit_cast(bf16x8, pk);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STAMPS) {          // the stamp's lgkmcnt(0) drains the 8 W2 reads: re-issue nothing, just account GELU
                const unsigned long long t_ = stamp_now(); ph[PH_GELU] += t_ - tprev; tprev = t_;
            }
            lds_wait<4>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int tg = 0; tg < TG; ++tg) acc2[tg][n] = mfma16x16x32(w2f[n], hfr[tg], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n) w2f[n] = lds_read_b128_asm_off(w2a, 2 * (64 * 128) + n * 2048);
            lds_wait<4>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 4; n < 8; ++n)
#pragma unroll
                for (int tg = 0; tg < TG; ++tg) acc2[tg][n] = mfma16x16x32(w2f[n], hfr[tg], acc2[tg][n]);
            __builtin_amdgcn_sched_barrier(0);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int tg = 0; tg < TG; ++tg) acc2[tg][8 + n] = mfma16x16x32(w2f[n], hfr[tg], acc2[tg][8 + n]);
            __builtin_amdgcn_sched_barrier(0);
            B64_STAMP(PH_FC2);
        }
    }
    if (active) {
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM;
#pragma unroll
            for (int n = 0; n < 12; ++n)
                *reinterpret_cast<f32x4*>(xr + (n >> 2) * 64 + g * 16 + (n & 3) * 4) = acc2[tg][n];
        }
    }
    }   // blk
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B64_STAMP(PH_STORE);
        ph[PH_TOTAL] = tprev - tstart;
        const int b = blockIdx.x;
        const int rec = b < 4 ? b : (b >= 300 && b < 304 ? 4 + b - 300 : -1);
        if (rec >= 0 && lane == 0)
#pragma unroll
            for (int k = 0; k < NPH; ++k) tup_b64_stamps[rec][wave][k] = ph[k];
    }
#undef B64_STAMP
}

}  // namespace

namespace {
int launch_block64(float* x, const Block64Table& tbl, int nblk, int nwin, void* stream)
{
    static const bool stamps = getenv("TUP_B64_STAMPS") != nullptr;          // diagnostic build (timing shares only)
    if (stamps) {
        TUP_SET_DYN_LDS((fused_block64_kernel<true>), B64_LDS);
        fused_block64_kernel<true><<<dim3((nwin + 3) / 4), dim3(256), B64_LDS, reinterpret_cast<hipStream_t>(stream)>>>(x, tbl, nblk, nwin);
    } else {
        TUP_SET_DYN_LDS((fused_block64_kernel<false>), B64_LDS);
        fused_block64_kernel<false><<<dim3((nwin + 3) / 4), dim3(256), B64_LDS, reinterpret_cast<hipStream_t>(stream)>>>(x, tbl, nblk, nwin);
    }
    TUP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// nblk (<= 8) consecutive WindowTransformerBlocks, in place, in ONE launch (the loop of model.py:288-289), one wave per window.
// table: HOST array [nblk][13] of device pointers in the argument order of tup_fused_block_fwd after x
// (gamma1, beta1, wh, bh, bias_frag, wproj, bproj, gamma2, beta2, w1, b1, w2, b2), same packing.
extern "C" int tup_fused_blocks64_fwd(float* x, const void* const* table, int nblk, int nwin, void* stream)
{
    if (nwin <= 0 || nblk <= 0) return 0;
    if (nblk > MAX_BLK || table == nullptr) return (int)hipErrorInvalidValue;
    Block64Table t{};
    static_assert(sizeof(Block64Ptrs) == 13 * sizeof(void*), "one table row = 13 pointers");
    memcpy(&t, table, (size_t)nblk * sizeof(Block64Ptrs));
    return launch_block64(x, t, nblk, nwin, stream);
}

// Timing experiments only: per-phase cycle sums of the last TUP_B64_STAMPS=1 launch, [8 workgroups][4 waves][16 phases].
extern "C" int tup_debug_block64_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tup_b64_stamps), sizeof(unsigned long long) * 8 * 4 * NPH);
}
