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

// PROJ = true additionally runs attn.proj + the residual add (model.py:131,164) on the attention output while it is still
// in registers: the 12 per-head O^T tiles ARE the token fragments of that GEMM (two heads = one K-step of 32, weight
// columns pre-permuted to match), W_proj streams through the two weight slots in 3 chunks, and x is updated in place.
template <bool PROJ>
__global__ __launch_bounds__(256, 2) void fused_qkv_attn_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const bf16_t* __restrict__ wh, const float* __restrict__ bh, const float* __restrict__ bias_frag,
    bf16_t* __restrict__ out, int nwin, const bf16_t* __restrict__ wproj, const float* __restrict__ bproj, float* __restrict__ xio)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    const int wi = wave >> 1, hf = wave & 1;
    const int win = blockIdx.x * 2 + wi;
    const bool active = win < nwin;
    const int row0 = (active ? win : nwin - 1) * 64 + 32 * hf;          // first of this wave's 32 token rows

    float* qb = reinterpret_cast<float*>(smem + QB_OFF);
    for (int i = tid; i < HEADS * 48; i += 256) qb[i] = bh[i];

    // head h's weight slot by DMA: slot s = u*256 + tid -> k-tile u >> 1, row (u & 1)*32 + (tid >> 3), logical chunk
    // (tid & 7) ^ ((tid >> 4) & 7) (swizzle on the source side)
    const bf16_t* w_thr = wh + (size_t)(tid >> 3) * DIM + ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    auto dma_w = [&](int h, int buf) {
        char* dst = smem + buf * FW_BYTES + wave * 1024;
        const bf16_t* src = w_thr + (size_t)h * 64 * DIM;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * DIM + (u >> 1) * 64),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };
    dma_w(0, 0);
    dma_w(1, 1);

    // ---- LayerNorm1 straight into B fragments: token 16tg+pl, channels 32*st + 8g .. +8 ----
    bf16x8 tf[2][6];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        const float* xr = x + (size_t)(row0 + 16 * tg + pl) * DIM + 8 * g;
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
    // K / V tiles of this wave's window: [token][16 channels] bf16, 32-byte rows
    const uint32_t kv_win = sbase + KV_OFF + (uint32_t)wi * 4096;
    const uint32_t kv_wr = (uint32_t)(32 * hf + pl) * 32 + g * 8;            // + 16*tg*32; own tokens, channels 4g..
    const uint32_t k_rd = (uint32_t)pl * 32 + g * 8;                          // + 16*kt*32: K[key 16kt+pl][4g..]
    const int trq = pl >> 2, trp = pl & 3;
    const uint32_t v_rd = 2048 + (uint32_t)(4 * g + trq) * 32 + trp * 8;      // + 16*kt*32: V rows 16kt+4g+trq (transposed read)
    bf16_t* sink = reinterpret_cast<bf16_t*>(tup_fa_sink) + lane * 4;

    // PROJ: the attention output of this wave's 32 tokens, all heads: of[tg][h] = O^T tile (channels 4g.., token pl) as bf16x4
    s16x4 of[2][PROJ ? HEADS : 1];
    const bf16_t* wp_thr = PROJ ? wproj + (size_t)(tid >> 3) * DIM + ((tid & 7) ^ ((tid >> 4) & 7)) * 8 : nullptr;
    auto dma_wp = [&](int chunk, int buf) {                      // rows 64*chunk .. +63 of the packed proj weight
        char* dst = smem + buf * FW_BYTES + wave * 1024;
        const bf16_t* src = wp_thr + (size_t)chunk * 64 * DIM;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u & 1) * 32 * DIM + (u >> 1) * 64),
                                             (__attribute__((address_space(3))) void*)(dst + u * 4096), 16, 0, 0);
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // own pieces of head 0's weights; bias staging visible below
    __syncthreads();
    auto head = [&](const int h) {
        // relative position bias of this wave's (key tile, query tile) pairs, requested BEFORE the DMA below so that the
        // compiler's wait for them is vmcnt(6) (= the DMA pieces), not vmcnt(0)
        f32x4 rb[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                rb[tg][kt] = *reinterpret_cast<const f32x4*>(bias_frag + ((((size_t)h * 4 + kt) * 4 + (2 * hf + tg)) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);

        // ---- [q; k; v]^T = W_h LN(x)^T: rows = channel within the head (ct 0 = q, 1 = k, 2 = v), columns = tokens ----
        const uint32_t wb = sbase + (uint32_t)((h & 1) * FW_BYTES) + w_off;
        f32x4 acc[2][3];
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
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[1][ct] = acc[0][ct];
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int cur = step % 3;
                if (step + 2 < 6) { ld(step + 2, (step + 2) % 3); lds_wait<6>(); }
                else if (step + 1 < 6) { lds_wait<3>(); }
                else { lds_wait<0>(); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[tg][ct] = mfma16x16x32(wf[cur][ct], tf[tg][step], acc[tg][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- q stays (scaled); K and V go to the window's LDS tile ----
        s16x4 qf[2];
        const uint32_t kvb = kv_win + (uint32_t)((h & 1) * 8192);
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            qf[tg] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(acc[tg][0][0] * 0.25f, acc[tg][0][1] * 0.25f),
                                                    pack_bf16x2(acc[tg][0][2] * 0.25f, acc[tg][0][3] * 0.25f)});
            lds_write_b64_asm(kvb + kv_wr + tg * 512, u32x2{pack_bf16x2(acc[tg][1][0], acc[tg][1][1]), pack_bf16x2(acc[tg][1][2], acc[tg][1][3])});
            lds_write_b64_asm(kvb + 2048 + kv_wr + tg * 512, u32x2{pack_bf16x2(acc[tg][2][0], acc[tg][2][1]), pack_bf16x2(acc[tg][2][2], acc[tg][2][3])});
        }
        // ONE barrier per head.  Before it every wave waits for its own pieces of the NEXT head's weights (requested a head
        // ago; younger than them: the 2 output stores of head h-1 -- none with PROJ -- and the 8 bias loads above), so
        // passing it means: K / V of this head are written, the next head's weights have landed everywhere, and
        // everyone is done reading this head's weight slot -- which is refilled right away, two heads ahead.
        if constexpr (PROJ) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (h + 2 < HEADS) dma_w(h + 2, h & 1);
        else if constexpr (PROJ) dma_wp(h + 2 - HEADS, h & 1);  // proj chunks 0 and 1 take the place of "heads 12 and 13"
        s16x4 kf[4], vf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            kf[kt] = lds_read_b64_asm(kvb + k_rd + kt * 512);
            vf[kt] = lds_read_tr16_asm(kvb + v_rd + kt * 512);
        }
        lds_wait<0>();
        __builtin_amdgcn_sched_barrier(0);

        // ---- S^T = K Q^T + bias, softmax over keys, O^T = V^T P^T ----
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            f32x4 st[4];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                st[kt] = mfma16x16x16(kf[kt], qf[tg], rb[tg][kt]);          // bias as the accumulator input
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, st[kt][e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) { st[kt][e] = __expf(st[kt][e] - mx); sum += st[kt][e]; }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                s16x4 pp[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 pv = st[2 * kp + hh];
                    // normalise before the bf16 rounding of P (softmax output is what the reference multiplies by v)
                    pp[hh] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(pv[0] * inv, pv[1] * inv), pack_bf16x2(pv[2] * inv, pv[3] * inv)});
                }
                o = mfma16x16x32(join4(vf[2 * kp], vf[2 * kp + 1]), join4(pp[0], pp[1]), o);
            }
            // O^T tile: rows = channel 4g+e, column = query pl  ->  out[row][h*16 + 4g .. +3]
            if constexpr (PROJ) {
                of[tg][h] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])});
            } else {
                bf16_t* op = active ? out + (size_t)(row0 + 16 * tg + pl) * DIM + h * HD + 4 * g : sink;
                *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
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
        f32x4 acc2[2][12];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int n = 0; n < 12; ++n) acc2[tg][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of chunk c
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone's pieces; everyone finished chunk c-1
            if (c == 1) dma_wp(2, 0);
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
                for (int tg = 0; tg < 2; ++tg) {
                    const bf16x8 tfp = join4(of[tg][2 * step], of[tg][2 * step + 1]);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc2[tg][4 * c + ct] = mfma16x16x32(wf[cur][ct], tfp, acc2[tg][4 * c + ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!active) return;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
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
    }
}

}  // namespace

// att bf16 [M][192] = attention core of qkv(LayerNorm(x)) per 8x8 window (M = 64 * nwin rows in window order):
// x fp32 [M][192]; wh bf16 [12][64][192] = per head the q, k, v weight rows (16 each, natural channel order) + 16 zero
// rows; bh fp32 [12][48] the matching biases; bias_frag fp32 [12][4][4][64][4] from tup_relpos_bias_expand.
extern "C" int tup_fused_qkv_attn_fwd(const float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                                      const float* bias_frag, void* out, int nwin, void* stream)
{
    if (nwin <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FA_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    fused_qkv_attn_kernel<false><<<dim3((nwin + 1) / 2), dim3(256), FA_LDS, reinterpret_cast<hipStream_t>(stream)>>>(
        x, gamma, beta, (const bf16_t*)wh, bh, bias_frag, (bf16_t*)out, nwin, nullptr, nullptr, nullptr);
    TUP_CHECK_LAUNCH();
    return 0;
}

// The whole attention half of a block, in place: x += proj(attention(qkv(LayerNorm1(x)))) + b_proj (model.py:163-164).
// wproj bf16 [192][192]: attn.proj.weight with rows permuted per 64-group and columns ordered by packing.pack_proj_pairs.
extern "C" int tup_fused_attn_block_fwd(float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                                        const float* bias_frag, const void* wproj, const float* bproj, int nwin, void* stream)
{
    if (nwin <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FA_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    fused_qkv_attn_kernel<true><<<dim3((nwin + 1) / 2), dim3(256), FA_LDS, reinterpret_cast<hipStream_t>(stream)>>>(
        x, gamma, beta, (const bf16_t*)wh, bh, bias_frag, nullptr, nwin, (const bf16_t*)wproj, bproj, x);
    TUP_CHECK_LAUNCH();
    return 0;
}
