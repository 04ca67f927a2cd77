// Fused first half of a WindowTransformerBlock for inference (gfx950):
//     att = WindowAttention.core(qkv(LayerNorm1(x)))          models/FastTransformer/model.py:104-130,163
// i.e. norm1, the qkv Linear, q*scale, q k^T + relative position bias, softmax and P v in ONE kernel: the 141 MB
// (B = 8) qkv tensor is never written or read.  The unfused sequence (LN+QKV panel GEMM, attention kernel) moves
// 94 + 141 + 141 + 47 MB and both kernels sit at ~4 TB/s; this one moves 94 + 47 MB.
//
// One workgroup = 4 waves = 2 windows; wave w owns half hf = w & 1 (32 query rows = token tiles tg 0, 1) of window
// w >> 1.  LN1(x) of its 32 rows lives in registers as MFMA B fragments (as in the fused MLP v2).  Per head:
//   * the head's 48 weight rows (q | k | v, natural channel order) arrive by LDS-DMA one head ahead (24 KB slots,
//     padded to 64 rows) and are the A operand of  [q; k; v]^T[48][32 tokens] = W_h LN(x)^T  (36 MFMAs 16x16x32);
//   * the accumulator layout (rows = head channel 4g+e, column = token) is already the B-operand layout of q in
//     S^T = K Q^T and the A-operand layout of K, so q never moves and K / V only cross to the window's other wave
//     through a 2 KB LDS tile each (V is read back transposed with ds_read_b64_tr_b16);
//   * S^T, softmax and O^T = V^T P^T as in attention.hip (PV on 16x16x32 by pairing key tiles).
// Every LDS access inside the head loop is inline asm: a compiler-visible LDS access behind an outstanding LDS-DMA makes
// hipcc wait vmcnt(0); barriers are raw s_barrier with counted waits for the same reason.
#include "common.h"
#include <string.h>
#include <stdlib.h>


namespace {

constexpr int DIM = 192, HEADS = 12, HD = 16;
constexpr int FW_BYTES = 3 * 64 * 128;                     // one head's weight slot: 3 k-tiles x [64 rows][128 B]
constexpr int KV_OFF = 2 * FW_BYTES;                       // [2 parities][2 windows][K 2 KB | V 2 KB]
constexpr int QB_OFF = KV_OFF + 2 * 2 * 4096;              // qkv bias, fp32 [12][48]
constexpr int FA_LDS = QB_OFF + HEADS * 48 * 4;

TUP_DEVICE void lds_write_b64_asm(uint32_t addr, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
TUP_DEVICE s16x4 lds_read_b64_asm(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
TUP_DEVICE s16x4 lds_read_tr16_asm(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
TUP_DEVICE bf16x8 join4(s16x4 lo, s16x4 hi) {
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}
__device__ __attribute__((aligned(16))) unsigned int tup_fa_sink[64 * 2];       // store sink of inactive lanes (fixed vmcnt)
// LDS-DMA of one 1 KB piece as a BUFFER load: resource descriptor (SGPRs) + per-lane byte offset (ONE VGPR, the same for every piece
// of a stream) + wave-uniform byte offset of the piece (SGPR).  The global_load_lds form took a per-lane 64-bit pointer per piece:
// one v_lshl_add_u64 each and a write-after-read dependence on that pointer register between consecutive pieces (234 pieces per
// wave and block, ~100 cycles each in the stamps of DESIGN 5b); this form issues no VALU instruction at all.
TUP_DEVICE void dma_piece(__amdgpu_buffer_rsrc_t r, char* lds_dst, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, soff, 0, 0);
}
TUP_DEVICE __amdgpu_buffer_rsrc_t weight_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// PROJ = true additionally runs attn.proj + the residual add (model.py:131,164) on the attention output while it is still
// in registers: the 12 per-head O^T tiles ARE the token fragments of that GEMM (two heads = one K-step of 32, weight
// columns pre-permuted to match), W_proj streams through the two weight slots in 3 chunks, and x is updated in place.
//
// MLP = true (with PROJ) appends the second half of the block, x += mlp.2(GELU(mlp.0(LayerNorm2(x)))) (model.py:165-171): the
// proj accumulators + bias + residual ARE the new residual stream of this wave's 32 tokens, in the layout of the fused
// MLP's FC2 accumulators (fused_blocks.hip), so LayerNorm2 and its MFMA fragments are computed from registers, the chunk
// loop of that kernel follows, and x is written once per block -- one fp32 write + one read of x (94 MB each at B = 8)
// and the load / store phases of a second kernel disappear.  LDS: W1 double buffer = the two weight slots, W2 chunk over
// the (by then dead) K/V tiles and qkv bias, b1 behind it; 75 KB, two workgroups per CU as before.
struct MlpArgs {
    const float* gamma2; const float* beta2;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
};
constexpr int HID = 768;
// parameters of one WindowTransformerBlock; the table is a kernel argument BY VALUE (pointer fields of a kernel argument are
// global pointers to hipcc; pointers read from memory would be generic and every load through them a flat_load)
struct BlockPtrs {
    const float* gamma1; const float* beta1; const bf16_t* wh; const float* bh; const float* bias_frag;
    const bf16_t* wproj; const float* bproj; MlpArgs ma;
};
constexpr int MAX_BLK = 8;
struct BlockTable { BlockPtrs b[MAX_BLK]; };
// ... + the four per-channel vectors of the block's second half (b_proj, gamma2, beta2, b2: [4][192] fp32), staged with the biases:
// read from global where they are used they cost ~45 dependent round trips behind the weight DMA queue (17k cycles per block)
constexpr int M_W2_OFF = 2 * FW_BYTES, M_B1_OFF = 3 * FW_BYTES, M_VEC_OFF = M_B1_OFF + HID * 4;
constexpr int FB_LDS = M_VEC_OFF + 4 * DIM * 4;                                                      // 79,872 B, two per CU

// Diagnostic build only (STAMPS = true, TUP_B32_STAMPS=1): s_memtime at phase boundaries, summed per phase kind, for 8 recorded
// workgroups; the values leave through a buffer nothing else reads.
constexpr int NPH32 = 16;
__device__ unsigned long long tup_b32_stamps[8][4][NPH32];
TUP_DEVICE unsigned long long stamp_now32() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
enum { P_LN1 = 0, P_SYNC0, P_QKV, P_HBAR, P_ATT, P_PROJ, P_LN2, P_MTOP, P_FC1, P_W2BAR, P_GELU, P_FC2, P_STORE, P_TOTAL };

// LayerNorm without scale / shift of one token group, from the accumulator layout (lane (g, pl): token pl, channels
// (n >> 2) * 64 + 16 g + (n & 3) * 4 .. +3) straight into the B fragments of the product that follows (K-step st = values
// n = 2 st, 2 st + 1; the following weight's K columns are packed to match, packing._fused_k_order).
TUP_DEVICE void ln_fragments(const f32x4 (&v)[12], bf16x8 (&tf)[6])
{
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int n = 0; n < 12; n += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0 += v[n][e]; q0 = __builtin_fmaf(v[n][e], v[n][e], q0);
            s1 += v[n + 1][e]; q1 = __builtin_fmaf(v[n + 1][e], v[n + 1][e], q1);
        }
    const float sum = rows_sum(s0 + s1), sq = rows_sum(q0 + q1);
    const float mean = sum * (1.0f / DIM);
    const float rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, sq * (1.0f / DIM)), 0.f) + 1e-5f);
    const float shift = -mean * rstd;
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        uint32_t pk[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 u = v[2 * st + hh];
            pk[2 * hh] = pack_bf16x2(__builtin_fmaf(u[0], rstd, shift), __builtin_fmaf(u[1], rstd, shift));
            pk[2 * hh + 1] = pack_bf16x2(__builtin_fmaf(u[2], rstd, shift), __builtin_fmaf(u[3], rstd, shift));
        }
        tf[st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
    }
}

template <bool PROJ, bool MLP = false, bool STAMPS = false, int TGN = 2>
__global__ __launch_bounds__(256, 2) void fused_qkv_attn_kernel(
    const float* __restrict__ x_in, bf16_t* __restrict__ out, int nwin, float* __restrict__ xio, const BlockTable tbl, int nblk)
{
    // in place (PROJ): every access to the residual stream goes through xio -- with the block loop below a load through a second
    // __restrict__ pointer could be moved above the previous block's stores
    const float* x = PROJ ? xio : x_in;
    static_assert(!MLP || PROJ, "the MLP half follows the proj");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // One of a CU's two resident workgroups runs at wave priority 3 (the high half of nblk carries "shift + 1 | level << 5";
        // the dispatcher fills the CUs once before it doubles up, so workgroups j and j + #CUs share a CU: bit log2(#CUs) of
        // blockIdx).  With equal priorities the two waves of a SIMD contend for every issue slot and drift in and out of phase;
        // a fixed pecking order lets one stream its MFMA segments while the other fills the gaps: 973-987 -> 961-962 us for the
        // six-block launch at 1,920 windows (levels 1 / 2: 965-970 / 938-965; the other bit positions: no effect).
        const int pr = nblk >> 16, pshift = pr & 31, lvl = pr >> 5;
        nblk &= 0xffff;
        if (pshift && ((blockIdx.x >> (pshift - 1)) & 1)) {
            if (lvl == 3) __builtin_amdgcn_s_setprio(3); else if (lvl == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
        }
    }
    unsigned long long ph[NPH32] = {}, tprev = 0, tstart = 0;
    if constexpr (STAMPS) tprev = tstart = stamp_now32();
#define B32_STAMP(K) do { if constexpr (STAMPS) { const unsigned long long t_ = stamp_now32(); ph[K] += t_ - tprev; tprev = t_; } } while (0)

    // nblk consecutive blocks in one launch (whole-block variant): a window never meets another window (same partition in every
    // block, model.py:283-289), so this workgroup carries its two windows through all of them; between blocks x passes through
    // memory as the waves' own stores followed by their own loads (served by L2), the chip-wide load / store bursts of a launch
    // boundary and the boundary itself disappear
    // the residual stream of this wave's 32 tokens in the layout of the proj / FC2 accumulators (lane (g, pl): token 16 tg + pl,
    // channels (n >> 2) * 64 + 16 g + (n & 3) * 4 .. +3 for n < 12).  LayerNorm1 reads it in THIS layout (the qkv weight's K columns
    // are packed to match, packing.pack_qkv_heads, as mlp.0's are for LayerNorm2), so from the second block of a launch on it
    // is the previous block's accumulators: no load, the stream only goes out to memory (for the residual re-read at the proj)
    f32x4 xcar[TGN][12];
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
    // the thread's coordinates are recomputed per block from an opaque copy of threadIdx: as loop invariants hipcc hoists the
    // dozens of per-thread addresses derived from them out of the loop, spills them, and reloads them from scratch inside the
    // head loop -- scratch loads count on vmcnt, which the counted waits below rely on (LDS-DMA ordering)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < 256);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: DMA targets, window, row0, tok0 are SALU
    const int g = lane >> 4, pl = lane & 15;
    // TGN = 2 (default): two windows per workgroup, a wave owns 32 token rows (two tiles of 16); TGN = 1 (small launches): ONE window per
    // workgroup, a wave owns 16 token rows -- twice the workgroups and half the serial work per wave when the launch would not fill the
    // chip anyway (config 4: 540 windows = 270 workgroups on 512 slots; the overlay frame: 240 windows).  Same arithmetic per token.
    const int wi = TGN == 2 ? wave >> 1 : 0;
    const int tok0 = TGN == 2 ? 32 * (wave & 1) : 16 * wave;                // first of this wave's 16 * TGN token rows inside its window
    const int win = TGN == 2 ? blockIdx.x * 2 + wi : blockIdx.x;
    const bool active = win < nwin;
    const int row0 = (active ? win : nwin - 1) * 64 + tok0;
    const BlockPtrs& bp = tbl.b[blk];
    const float* __restrict__ gamma = bp.gamma1; const float* __restrict__ beta = bp.beta1;
    const bf16_t* __restrict__ wh = bp.wh; const float* __restrict__ bh = bp.bh; const float* __restrict__ bias_frag = bp.bias_frag;
    const bf16_t* __restrict__ wproj = bp.wproj; const float* __restrict__ bproj = bp.bproj;
    const MlpArgs ma = bp.ma;
    // Order of the block's prologue: (first block of a launch) the x loads, the longest latency; the barrier that frees the
    // previous block's LDS; the weight DMA of heads 0 and 1; the loads of the small vectors; LayerNorm1 (registers only from the
    // second block on) while all of that is in flight; then the vectors' LDS stores.  Written as load / store / load / store the
    // prologue was three dependent round trips (stamps: 8-10 k cycles per block for 3 k cycles of arithmetic).
    if constexpr (MLP) {
        if (blk == 0) {
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg) {
                const float* xr = x + (size_t)(row0 + 16 * tg + pl) * DIM + g * 16;
#pragma unroll
                for (int n = 0; n < 12; ++n) xcar[tg][n] = *reinterpret_cast<const f32x4*>(xr + (n >> 2) * 64 + (n & 3) * 4);
            }
        }
    }
    if (blk) __syncthreads();          // everyone is done with the previous block's LDS (weight slots, W2 chunk, biases)

    // head h's weight slot by DMA: slot s = u*256 + tid -> k-tile u >> 1, row (u & 1)*32 + (tid >> 3), logical chunk
    // (tid & 7) ^ ((tid >> 4) & 7) (swizzle on the source side)
    // per-lane byte offsets of the weight streams: row (tid >> 3) of a 32-row half tile, 16-byte chunk (tid & 7) ^ swizzle, in a
    // matrix of DIM (qkv, proj, mlp.0) or HID (mlp.2) columns
    const uint32_t voff_d = (uint32_t)(((tid >> 3) * DIM + ((tid & 7) ^ ((tid >> 4) & 7)) * 8) * 2);
    const uint32_t voff_h = (uint32_t)(((tid >> 3) * HID + ((tid & 7) ^ ((tid >> 4) & 7)) * 8) * 2);
    auto dma_w = [&](int h, int buf) {
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(wh);
        char* dst = smem + buf * FW_BYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_piece(r, dst + u * 4096, voff_d, (h * 64 * DIM + (u & 1) * 32 * DIM + (u >> 1) * 64) * 2);
    };
    dma_w(0, 0);
    dma_w(1, 1);

    // fixed trip counts (tid is opaque to the compiler here)
    static_assert(HEADS * 48 == 576 && HID / 4 <= 256, "staging below assumes 576 biases and <= 256 float4 of b1");
    float* qb = reinterpret_cast<float*>(smem + QB_OFF);
    const float q0 = bh[tid], q1 = bh[256 + tid], q2 = bh[512 + (tid & 63)];
    f32x4 b1v = {};
    float c0 = 0.f, c3 = 0.f;
    if constexpr (MLP) {
        b1v = reinterpret_cast<const f32x4*>(ma.b1)[tid < HID / 4 ? tid : 0];
        c0 = bproj[tid < DIM ? tid : 0]; c3 = ma.b2[tid < DIM ? tid : 0];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm1 straight into B fragments.  K-step st of the qkv product contracts over channels
    // 64 (st >> 1) + 16 g + 8 (st & 1) .. +8 = the lane's residual values n = 2 st, 2 st + 1 ----
    bf16x8 tf[TGN][6];
    if constexpr (MLP) {
        // Whole-block form: gamma1 / beta1 are folded into the qkv weight and bias (packing.fold_layernorm), so the B fragments are
        // the plain normalised values: one FMA per value, statistics in ONE sweep (sum and sum of squares, var = E[x^2] - mean^2 in
        // fp32: the stream's |mean| / std stays far below the 2^12 where that form loses the bf16 digits the fragments keep), the
        // four lane groups of a token joined by permlane swaps.  288 VALU instructions per LayerNorm instead of 576 + 24 loads.
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) ln_fragments(xcar[tg], tf[tg]);
        __builtin_amdgcn_sched_barrier(0);
    }
    qb[tid] = q0; qb[256 + tid] = q1;
    if (tid < 64) qb[512 + tid] = q2;
    if constexpr (MLP) {       // mlp.0's bias, b_proj, b2 (LayerNorm2's scale and shift live in mlp.0's weight and bias)
        if (tid < HID / 4) reinterpret_cast<f32x4*>(smem + M_B1_OFF)[tid] = b1v;
        if (tid < DIM) {
            float* vec = reinterpret_cast<float*>(smem + M_VEC_OFF);
            vec[tid] = c0; vec[DIM + tid] = c3;
        }
    }
    if constexpr (!MLP) {
        // the loads (the wave's x rows, gamma, beta) stand together in program order, ahead of the arithmetic: written per row as
        // load-then-use, hipcc under register pressure emits one global round trip per pair of loads
        f32x4 gm[12], bt[12];
        if (!MLP || blk == 0) {
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg) {
                const float* xr = x + (size_t)(row0 + 16 * tg + pl) * DIM + g * 16;
#pragma unroll
                for (int n = 0; n < 12; ++n) xcar[tg][n] = *reinterpret_cast<const f32x4*>(xr + (n >> 2) * 64 + (n & 3) * 4);
            }
        }
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            gm[n] = *reinterpret_cast<const f32x4*>(gamma + (n >> 2) * 64 + g * 16 + (n & 3) * 4);
            bt[n] = *reinterpret_cast<const f32x4*>(beta + (n >> 2) * 64 + g * 16 + (n & 3) * 4);
        }
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) {
            float sum = 0.f;
#pragma unroll
            for (int st = 0; st < 6; ++st)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += xcar[tg][2 * st][e] + xcar[tg][2 * st + 1][e];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / DIM);
            float ss = 0.f;
#pragma unroll
            for (int n = 0; n < 12; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xcar[tg][n][e] - mean; ss += d * d; }
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            const float rstd = rsqrtf(ss * (1.0f / DIM) + 1e-5f);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                uint32_t pk[4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int n = 2 * st + hh;
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (xcar[tg][n][e] - mean) * rstd * gm[n][e] + bt[n][e];
                    pk[2 * hh] = pack_bf16x2(o4[0], o4[1]);
                    pk[2 * hh + 1] = pack_bf16x2(o4[2], o4[3]);
                }
                tf[tg][st] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
            }
        }
    }

    const uint32_t sbase = lds_addr(smem);
    const uint32_t w_off = (uint32_t)swz128(pl, g);
    const uint32_t qb_addr = sbase + QB_OFF + (uint32_t)(4 * g) * 4;
    // K / V tiles of this wave's window: [token][16 channels] bf16, 32-byte rows
    const uint32_t kv_win = sbase + KV_OFF + (uint32_t)wi * 4096;
    const uint32_t kv_wr = (uint32_t)(tok0 + pl) * 32 + g * 8;               // + 16*tg*32; own tokens, channels 4g..
    const uint32_t k_rd = (uint32_t)pl * 32 + g * 8;                          // + 16*kt*32: K[key 16kt+pl][4g..]
    const int trq = pl >> 2, trp = pl & 3;
    const uint32_t v_rd = 2048 + (uint32_t)(4 * g + trq) * 32 + trp * 8;      // + 16*kt*32: V rows 16kt+4g+trq (transposed read)
    bf16_t* sink = reinterpret_cast<bf16_t*>(tup_fa_sink) + lane * 4;

    // PROJ: the attention output of this wave's 32 tokens, all heads: of[tg][h] = O^T tile (channels 4g.., token pl) as bf16x4
    s16x4 of[TGN][PROJ ? HEADS : 1];
    auto dma_wp = [&](int chunk, int buf) {                      // rows 64*chunk .. +63 of the packed proj weight
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(wproj);
        char* dst = smem + buf * FW_BYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_piece(r, dst + u * 4096, voff_d, (chunk * 64 * DIM + (u & 1) * 32 * DIM + (u >> 1) * 64) * 2);
    };

    // MLP weight stream (same slot geometry and thread map as the qkv / proj slots)
    auto dma_w1_piece = [&](int j, int buf, int u) {
        dma_piece(weight_rsrc(ma.w1), smem + buf * FW_BYTES + wave * 1024 + u * 4096, voff_d, (j * 64 * DIM + (u & 1) * 32 * DIM + (u >> 1) * 64) * 2);
    };
    auto dma_w2 = [&](int j) {
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(ma.w2);
        char* dst = smem + M_W2_OFF + wave * 1024;
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_piece(r, dst + u * 4096, voff_h, (j * 64 + ((u >> 1) * 64 + (u & 1) * 32) * HID) * 2);
    };

    // proj / FC2 accumulators = the residual stream.  Whole-block form: the last head requests the stream's re-read (x as the previous
    // block -- or patch_embed -- left it) straight into them, so it lands under that head's softmax and the proj accumulates on top;
    // requested after the proj the 24 loads were a round trip of their own in front of LayerNorm2
    f32x4 acc2[PROJ ? TGN : 1][PROJ ? 12 : 1];
    B32_STAMP(P_LN1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // own pieces of head 0's weights; bias staging visible below
    __syncthreads();
    B32_STAMP(P_SYNC0);
    auto head = [&](const int h) {
        // relative position bias of this wave's (key tile, query tile) pairs, requested BEFORE the DMA below so that the
        // compiler's wait for them is vmcnt(6) (= the DMA pieces), not vmcnt(0)
        f32x4 rb[TGN][4];
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                rb[tg][kt] = *reinterpret_cast<const f32x4*>(bias_frag + ((((size_t)h * 4 + kt) * 4 + (tok0 / 16 + tg)) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);

        // ---- [q; k; v]^T = W_h LN(x)^T: rows = channel within the head (ct 0 = q, 1 = k, 2 = v), columns = tokens ----
        const uint32_t wb = sbase + (uint32_t)((h & 1) * FW_BYTES) + w_off;
        f32x4 acc[TGN][3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const f32x4 bv = __builtin_bit_cast(f32x4, lds_read_b128_asm(qb_addr + (uint32_t)((h * 48 + ct * 16) * 4)));
            acc[0][ct] = bv;           // bias rides in the accumulator (waited for below)
        }
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
            lds_wait<6>();             // the three bias reads have landed
            if constexpr (TGN == 2) {
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) acc[TGN - 1][ct] = acc[0][ct];
            }
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<6>(); }
                else if (step + 1 < 6) { lds_wait<3>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[tg][ct] = mfma16x16x32(wf[cur][ct], tf[tg][step], acc[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        B32_STAMP(P_QKV);
        // ---- q stays (scaled); K and V go to the window's LDS tile ----
        s16x4 qf[TGN];
        const uint32_t kvb = kv_win + (uint32_t)((h & 1) * 8192);
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) {
            qf[tg] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(acc[tg][0][0] * 0.25f, acc[tg][0][1] * 0.25f),
                                                    pack_bf16x2(acc[tg][0][2] * 0.25f, acc[tg][0][3] * 0.25f)});
            lds_write_b64_asm(kvb + kv_wr + tg * 512, u32x2{pack_bf16x2(acc[tg][1][0], acc[tg][1][1]), pack_bf16x2(acc[tg][1][2], acc[tg][1][3])});
            lds_write_b64_asm(kvb + 2048 + kv_wr + tg * 512, u32x2{pack_bf16x2(acc[tg][2][0], acc[tg][2][1]), pack_bf16x2(acc[tg][2][2], acc[tg][2][3])});
        }
        // ONE barrier per head.  Before it every wave waits for its own pieces of the NEXT head's weights (requested a head
        // ago; younger than them: the 2 output stores of head h-1 -- none with PROJ -- and the 8 bias loads above), so
        // passing it means: K / V of this head are written, the next head's weights have landed everywhere, and
        // everyone is done reading this head's weight slot -- which is refilled right away, two heads ahead.
        static_assert(TGN == 2 || (PROJ && MLP), "one window per workgroup is built for the whole-block form only");
        if constexpr (PROJ && TGN == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // 4 bias loads per token tile
        else if constexpr (PROJ) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef TUP_EXP_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        if (h + 2 < HEADS) dma_w(h + 2, h & 1);
        else if constexpr (PROJ) dma_wp(h + 2 - HEADS, h & 1);  // proj chunks 0 and 1 take the place of "heads 12 and 13"
        if constexpr (MLP) {
            if (h == HEADS - 1) {
#pragma unroll
                for (int tg = 0; tg < TGN; ++tg) {
                    const float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM;
#pragma unroll
                    for (int n = 0; n < 12; ++n) acc2[tg][n] = *reinterpret_cast<const f32x4*>(xr + (n >> 2) * 64 + g * 16 + (n & 3) * 4);
                }
            }
        }
        s16x4 kf[4], vf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            kf[kt] = lds_read_b64_asm(kvb + k_rd + kt * 512);
            vf[kt] = lds_read_tr16_asm(kvb + v_rd + kt * 512);
        }
        lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        B32_STAMP(P_HBAR);

        // ---- S^T = K Q^T + bias, softmax over keys, O^T = V^T P^T; both query tiles advance in lockstep (independent chains
        // cover each other's latencies), cross-lane reductions by v_permlane swaps instead of ds_bpermute, exp2 with the
        // log2(e) scale and the row maximum folded into one FMA, P enters the product unnormalised and O is scaled by 1 / sum ----
        f32x4 st[TGN][4];
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[tg][kt] = mfma16x16x16(kf[kt], qf[tg], rb[tg][kt]);          // bias as the accumulator input
        float mx[TGN];
#ifndef TUP_EXP_NOSOFTMAX      // timing experiment (wrong results): the kernel without the softmax's max / exp arithmetic = the most that
                               // hiding it under MFMAs could gain (scripts/ab_block.py, DESIGN 5c: -4.5 %); TUP_EXP_NOBAR likewise
                               // for the per-head barrier (-3.4 %)
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) {
            float m = fmaxf(fmaxf(st[tg][0][0], st[tg][0][1]), fmaxf(st[tg][0][2], st[tg][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) m = fmaxf(fmaxf(m, st[tg][kt][0]), fmaxf(st[tg][kt][1], fmaxf(st[tg][kt][2], st[tg][kt][3])));
            mx[tg] = m;
        }
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) mx[tg] = rows_max(mx[tg]) * 1.4426950408889634f;
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st[tg][kt][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[tg][kt][e], 1.4426950408889634f, -mx[tg]));
#endif
        // The row sums come off the matrix pipe: one more product with an all-ones A operand ("a V whose 16 channels are all 1")
        // leaves sum_k P[q][k] for query column pl in all four of the lane's rows -- the 32 additions per head and lane and the two
        // cross-lane reductions are 4 MFMAs (the kernel is bound by VALU issue, the pipe has the room); the sum is over the bf16
        // values that enter the product, i.e. numerator and denominator see the same rounding.
        const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
        for (int tg = 0; tg < TGN; ++tg) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f}, sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const s16x4 p0 = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(st[tg][2 * kp][0], st[tg][2 * kp][1]), pack_bf16x2(st[tg][2 * kp][2], st[tg][2 * kp][3])});
                const s16x4 p1 = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(st[tg][2 * kp + 1][0], st[tg][2 * kp + 1][1]), pack_bf16x2(st[tg][2 * kp + 1][2], st[tg][2 * kp + 1][3])});
                const bf16x8 pj = join4(p0, p1);
                o = mfma16x16x32(join4(vf[2 * kp], vf[2 * kp + 1]), pj, o);
                sm = mfma16x16x32(ones, pj, sm);
            }
            const float inv = __builtin_amdgcn_rcpf(sm[0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= inv;
            // O^T tile: rows = channel 4g+e, column = query pl  ->  out[row][h*16 + 4g .. +3]
            if constexpr (PROJ) {
                of[tg][h] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])});
            } else {
                bf16_t* op = active ? out + (size_t)(row0 + 16 * tg + pl) * DIM + h * HD + 4 * g : sink;
                *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
        B32_STAMP(P_ATT);
    };
    if constexpr (PROJ) {
#pragma unroll
        for (int h = 0; h < HEADS; ++h) head(h);               // unrolled: of[tg][h] must be a register, not an indexed array
    } else {
#pragma unroll 1
        for (int h = 0; h < HEADS; ++h) head(h);
    }

    if constexpr (PROJ) {
        // ---- x += att W_proj^T + b: K-step p = heads (2p, 2p+1); the packed weight has its columns ordered to match
        // join4's k map, so the fragment addressing is the standard one (k-tile p >> 1, chunk 4 (p & 1) + g) ----
        if constexpr (!MLP) {
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
                for (int n = 0; n < 12; ++n) acc2[tg][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of chunk c
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone's pieces; everyone finished chunk c-1
            if (c == 1) dma_wp(2, 0);
            if constexpr (MLP) {
                if (c == 2) {          // slot 1 (proj chunk 1) is free: mlp.0's first chunk lands under the last proj chunk
#pragma unroll
                    for (int u = 0; u < 6; ++u) dma_w1_piece(0, 1, u);
                }
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
                for (int tg = 0; tg < TGN; ++tg) {
                    const bf16x8 tfp = join4(of[tg][2 * step], of[tg][2 * step + 1]);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc2[tg][4 * c + ct] = mfma16x16x32(wf[cur][ct], tfp, acc2[tg][4 * c + ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        B32_STAMP(P_PROJ);
        if constexpr (!MLP) {
            if (!active) return;
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg) {
                float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM;
#pragma unroll
                for (int n = 0; n < 12; ++n) {
                    const int col = (n >> 2) * 64 + g * 16 + (n & 3) * 4;       // weight rows are permuted per 64-group
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + col);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bproj + col);
                    f32x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = acc2[tg][n][e] + bv[e] + rv[e];
                    *reinterpret_cast<f32x4*>(xr + col) = ov;
                }
            }
        } else {
            // ---- second half of the block.  acc2 <- the new residual stream x + proj + b_proj (inactive waves carry a
            // copy of the last window and take part in every barrier; only their final store is skipped) ----
            // per-channel vector k (0 b_proj, 1 b2) at this lane's channels of accumulator n
            const uint32_t vec_addr = sbase + (uint32_t)(M_VEC_OFF + g * 64);
            auto vec4 = [&](int k, int n) {
                return __builtin_bit_cast(f32x4, lds_read_b128_asm_off(vec_addr, k * (DIM * 4) + ((n >> 2) * 64 + (n & 3) * 4) * 4));
            };
#pragma unroll
            for (int q = 0; q < 3; ++q) {          // + b_proj (the residual is already in the accumulators)
                f32x4 bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = vec4(0, 4 * q + i);
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tg = 0; tg < TGN; ++tg) acc2[tg][4 * q + i] += bv[i];
            }
            // LayerNorm2 from the accumulators: this lane holds 48 of its token's 192 channels, the other three lane
            // groups the rest.  K-step st of FC1 contracts over channels 64*(st>>1) + 16g + 8*(st&1) .. +8 = accumulators
            // n = 2st, 2st+1 (packing.pack_fc1_fused), so the B fragments are packed straight from them.
            // gamma2 / beta2 are folded into mlp.0 (packing.fold_layernorm): plain normalised fragments, as LayerNorm1 above
            bf16x8 tf2[TGN][6];
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg) ln_fragments(acc2[tg], tf2[tg]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {          // FC2 accumulates onto x + b2
                f32x4 bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = vec4(1, 4 * q + i);
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tg = 0; tg < TGN; ++tg) acc2[tg][4 * q + i] += bv[i];
            }

            B32_STAMP(P_LN2);
            // ---- the chunk loop of fused_mlp_v2_kernel (fused_blocks.hip); W1 chunk j lives in slot (j + 1) & 1 ----
            for (int j = 0; j < HID / 64; ++j) {
                // this lane's LDS addresses, rebuilt per chunk from a fresh lane id (see lane_id_fresh: hoisted, they are spilled
                // and their reload waits for the W2 DMA issued just above it)
                const int lf = lane_id_fresh(), g = lf >> 4, pl = lf & 15;
                const uint32_t w_off = (uint32_t)swz128(pl, g);
                const uint32_t w2_off0 = (uint32_t)(M_W2_OFF + swz128(pl, 2 * g)), w2_off1 = (uint32_t)(M_W2_OFF + swz128(pl, 2 * g + 1));
                const uint32_t b1_base = sbase + M_B1_OFF + (uint32_t)(g * 64);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of W1 chunk j have landed
                __syncthreads();                                       // everyone's have; everyone is done with chunk j-1 / the proj
                dma_w2(j);
                const bool more = j + 1 < HID / 64;
                __builtin_amdgcn_sched_barrier(0);
                // every LDS read below = one of five per-chunk base registers + an immediate offset
                const uint32_t wb1 = sbase + (uint32_t)(((j + 1) & 1) * FW_BYTES);
                const uint32_t w1a0 = wb1 + w_off, w1a1 = wb1 + (w_off ^ 64u);
                const uint32_t w2a0 = sbase + w2_off0, w2a1 = sbase + w2_off1, b1j = b1_base + (uint32_t)(j * 256);
                B32_STAMP(P_MTOP);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x4 acc1[TGN][2];
#pragma unroll
                    for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) acc1[tg][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 bb[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        bb[hh] = __builtin_bit_cast(f32x4, lds_read_b128_asm_off(b1j, (2 * s + hh) * 16));
                    // fragments three K-steps ahead: a step is only 4 MFMAs (64 cycles) and an LDS read under load takes
                    // 150-250 (stamps: FC1 ran at 3x its MFMA time with two steps of lookahead)
                    bf16x8 wf[4][2];
                    auto ld1 = [&](int step, int slot) {
                        const int off = (step >> 1) * (64 * 128) + 2 * s * 2048;
                        wf[slot][0] = (step & 1) ? lds_read_b128_asm_off(w1a1, off) : lds_read_b128_asm_off(w1a0, off);
                        wf[slot][1] = (step & 1) ? lds_read_b128_asm_off(w1a1, off + 2048) : lds_read_b128_asm_off(w1a0, off + 2048);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    ld1(0, 0);
                    ld1(1, 1);
                    ld1(2, 2);
#pragma unroll
                    for (int step = 0; step < 6; ++step) {
                        const int cur = step % 4;
                        if (step + 3 < 6) { ld1(step + 3, (step + 3) % 4); lds_wait<6>(); }
                        else if (step + 2 < 6) lds_wait<4>();
                        else if (step + 1 < 6) lds_wait<2>();
                        else lds_wait<0>();
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh)
                                acc1[tg][hh] = mfma16x16x32(wf[cur][hh], tf2[tg][step], step == 0 ? bb[hh] : acc1[tg][hh]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s == 0 && more) { dma_w1_piece(j + 1, j & 1, step); __builtin_amdgcn_sched_barrier(0); }
                    }
                    B32_STAMP(P_FC1);
                    if (s == 0) {
                        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // W2 chunk j landed (younger: the W1 prefetch)
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    B32_STAMP(P_W2BAR);
                    bf16x8 w2f[8];
                    auto w2rd = [&](int n) {
                        const int off = (n >> 2) * (64 * 128) + (n & 3) * 2048;
                        return s ? lds_read_b128_asm_off(w2a1, off) : lds_read_b128_asm_off(w2a0, off);
                    };
#pragma unroll
                    for (int n = 0; n < 8; ++n) w2f[n] = w2rd(n);
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 hfr[TGN];
                    gelu16_fragments<TGN>(acc1, hfr);          // fp16: gelu(x) / 4 (common.h)
                    __builtin_amdgcn_sched_barrier(0);
                    B32_STAMP(P_GELU);
                    lds_wait<4>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int tg = 0; tg < TGN; ++tg) acc2[tg][n] = mfma16x16x32_f16(w2f[n], hfr[tg], acc2[tg][n]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 4; ++n) w2f[n] = w2rd(8 + n);
                    lds_wait<4>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 4; n < 8; ++n)
#pragma unroll
                        for (int tg = 0; tg < TGN; ++tg) acc2[tg][n] = mfma16x16x32_f16(w2f[n], hfr[tg], acc2[tg][n]);
                    __builtin_amdgcn_sched_barrier(0);
                    lds_wait<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int tg = 0; tg < TGN; ++tg) acc2[tg][8 + n] = mfma16x16x32_f16(w2f[n], hfr[tg], acc2[tg][8 + n]);
                    __builtin_amdgcn_sched_barrier(0);
                    B32_STAMP(P_FC2);
                }
            }
            if (active) {
#pragma unroll
                for (int tg = 0; tg < TGN; ++tg) {
                    float* xr = xio + (size_t)(row0 + 16 * tg + pl) * DIM;
#pragma unroll
                    for (int n = 0; n < 12; ++n)
                        *reinterpret_cast<f32x4*>(xr + (n >> 2) * 64 + g * 16 + (n & 3) * 4) = acc2[tg][n];
                }
            }
#pragma unroll
            for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
                for (int n = 0; n < 12; ++n) xcar[tg][n] = acc2[tg][n];        // the next block's LayerNorm1 input
            if constexpr (STAMPS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                B32_STAMP(P_STORE);
                ph[P_TOTAL] = tprev - tstart;
                const int b = blockIdx.x;
                const int rec = b < 4 ? b : (b >= 600 && b < 604 ? 4 + b - 600 : -1);
                if (rec >= 0 && lane == 0 && blk == nblk - 1)
#pragma unroll
                    for (int k = 0; k < NPH32; ++k) tup_b32_stamps[rec][wave][k] = ph[k];
            }
        }
    }
    }   // blk
#undef B32_STAMP
}

}  // namespace

namespace {
BlockTable one_block(const float* g1, const float* b1n, const void* wh, const float* bh, const float* bias_frag, const void* wproj,
                     const float* bproj, const MlpArgs& ma)
{
    BlockTable t{};
    t.b[0] = BlockPtrs{g1, b1n, (const bf16_t*)wh, bh, bias_frag, (const bf16_t*)wproj, bproj, ma};
    return t;
}
int launch_blocks32(float* x, const BlockTable& t, int nblk, int nwin, void* stream)
{
#ifdef TUP_DIAG
    static const bool stamps = getenv("TUP_B32_STAMPS") != nullptr;          // `make diag` only (timing shares)
#else
    constexpr bool stamps = false;
#endif
    // launches of at most 512 windows (the chip's workgroup slots: 256 CUs x 2) run ONE window per workgroup, 16 token rows per wave:
    // twice the workgroups, half the serial work per wave -- the 720p -> 4K overlay frame (240 windows) 1.237 -> 1.166 ms; config 4's
    // 540 windows would be 1.05 rounds that way (2.59 vs 2.56 ms) and stay on the two-window form.  TUP_BLOCK_ONE_WINDOW=0 / 1 forces a form.
    static const int force1 = TUP_ENV_INT("TUP_BLOCK_ONE_WINDOW", -1);
    const bool one_window = force1 >= 0 ? force1 != 0 : nwin <= 512;
    const dim3 grid((nwin + 1) / 2);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (one_window && !stamps) {
        TUP_SET_DYN_LDS((fused_qkv_attn_kernel<true, true, false, 1>), FB_LDS);
        fused_qkv_attn_kernel<true, true, false, 1><<<dim3(nwin), dim3(256), FB_LDS, s>>>(x, nullptr, nwin, x, t, nblk);
#ifdef TUP_DIAG
    } else if (stamps) {
        TUP_SET_DYN_LDS((fused_qkv_attn_kernel<true, true, true>), FB_LDS);
        fused_qkv_attn_kernel<true, true, true><<<grid, dim3(256), FB_LDS, s>>>(x, nullptr, nwin, x, t, nblk);
#endif
    } else {
        TUP_SET_DYN_LDS((fused_qkv_attn_kernel<true, true>), FB_LDS);
        // wave priority of every second resident set (see the kernel's first lines); diagnostic library: TUP_BLOCK_PRIO=0 switches it off
        static const int prio = [] {
            const int forced = TUP_ENV_INT("TUP_BLOCK_PRIO", -1);
            if (forced >= 0) return forced;
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
            if (cus <= 0 || (cus & (cus - 1))) return 0;         // the bit test needs a power of two (MI355X: 256)
            int sh = 0;
            while ((1 << sh) < cus) ++sh;
            return (sh + 1) | (3 << 5);
        }();
        fused_qkv_attn_kernel<true, true><<<grid, dim3(256), FB_LDS, s>>>(x, nullptr, nwin, x, t, nblk | (prio << 16));
    }
    TUP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// att bf16 [M][192] = attention core of qkv(LayerNorm(x)) per 8x8 window (M = 64 * nwin rows in window order):
// x fp32 [M][192]; wh bf16 [12][64][192] = per head the q, k, v weight rows (16 each, natural channel order) + 16 zero
// rows; bh fp32 [12][48] the matching biases; bias_frag fp32 [12][4][4][64][4] from tup_relpos_bias_expand.
extern "C" int tup_fused_qkv_attn_fwd(const float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                                      const float* bias_frag, void* out, int nwin, void* stream)
{
    if (nwin <= 0) return 0;
    TUP_SET_DYN_LDS((fused_qkv_attn_kernel<false>), FA_LDS);
    fused_qkv_attn_kernel<false><<<dim3((nwin + 1) / 2), dim3(256), FA_LDS, reinterpret_cast<hipStream_t>(stream)>>>(
        x, (bf16_t*)out, nwin, nullptr, one_block(gamma, beta, wh, bh, bias_frag, nullptr, nullptr, MlpArgs{}), 1);
    TUP_CHECK_LAUNCH();
    return 0;
}

// The whole attention half of a block, in place: x += proj(attention(qkv(LayerNorm1(x)))) + b_proj (model.py:163-164).
// wproj bf16 [192][192]: attn.proj.weight with rows permuted per 64-group and columns ordered by packing.pack_proj_pairs.
extern "C" int tup_fused_attn_block_fwd(float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                                        const float* bias_frag, const void* wproj, const float* bproj, int nwin, void* stream)
{
    if (nwin <= 0) return 0;
    TUP_SET_DYN_LDS((fused_qkv_attn_kernel<true>), FA_LDS);
    fused_qkv_attn_kernel<true><<<dim3((nwin + 1) / 2), dim3(256), FA_LDS, reinterpret_cast<hipStream_t>(stream)>>>(
        x, nullptr, nwin, x, one_block(gamma, beta, wh, bh, bias_frag, wproj, bproj, MlpArgs{}), 1);
    TUP_CHECK_LAUNCH();
    return 0;
}

// One whole WindowTransformerBlock, in place (model.py:153-172): x += proj(attention(qkv(norm1(x)))); x += mlp(norm2(x)).
// The two LayerNorms' scale and shift arrive FOLDED into the Linear that follows them (packing.fold_layernorm: W diag(gamma),
// b + W beta -- the same function of x, and the kernel normalises with one FMA per value): wh / bh = packing.pack_qkv_heads of the
// folded attn.qkv, w1 / b1 = packing.pack_fc1_fused_q of the folded mlp.0; wproj, bproj, w2 (packing.pack_fc2_h4), b2 as in
// tup_fused_attn_block_fwd / tup_fused_mlp_fwd.
extern "C" int tup_fused_block_fwd(float* x, const void* wh, const float* bh, const float* bias_frag, const void* wproj,
                                   const float* bproj, const void* w1, const float* b1, const void* w2, const float* b2,
                                   int nwin, void* stream)
{
    if (nwin <= 0) return 0;
    const MlpArgs ma{nullptr, nullptr, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2};
    return launch_blocks32(x, one_block(nullptr, nullptr, wh, bh, bias_frag, wproj, bproj, ma), 1, nwin, stream);
}

// nblk (<= 8) consecutive WindowTransformerBlocks, in place, in ONE launch (the loop of model.py:288-289), two waves per window.
// table: HOST array [nblk][9] of device pointers in the argument order of tup_fused_block_fwd after x
// (wh, bh, bias_frag, wproj, bproj, w1, b1, w2, b2), same packing.
extern "C" int tup_fused_blocks32_fwd(float* x, const void* const* table, int nblk, int nwin, void* stream)
{
    if (nwin <= 0 || nblk <= 0) return 0;
    if (nblk > MAX_BLK || table == nullptr) return (int)hipErrorInvalidValue;
    BlockTable t{};
    for (int i = 0; i < nblk; ++i) {
        const void* const* r = table + (size_t)i * 9;
        const MlpArgs ma{nullptr, nullptr, (const bf16_t*)r[5], (const float*)r[6], (const bf16_t*)r[7], (const float*)r[8]};
        t.b[i] = BlockPtrs{nullptr, nullptr, (const bf16_t*)r[0], (const float*)r[1], (const float*)r[2], (const bf16_t*)r[3],
                           (const float*)r[4], ma};
    }
    return launch_blocks32(x, t, nblk, nwin, stream);
}

#ifdef TUP_DIAG
// Timing experiments only (`make diag`): per-phase cycle sums of the last TUP_B32_STAMPS=1 launch, [8 workgroups][4 waves][16 phases].
extern "C" int tup_debug_block32_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tup_b32_stamps), sizeof(unsigned long long) * 8 * 4 * NPH32);
}
#endif
