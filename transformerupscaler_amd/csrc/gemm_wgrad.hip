// Weight-gradient GEMMs of the token path (gfx950): out[i][j] += sum_m P[m][i] * Q[m][j].
//
// The reduction runs over token rows m, which is the slow axis of both operands in memory, so both
// MFMA fragments need a transpose: tiles are staged row-major into swizzled LDS and read back with
// ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major), which is
// exactly the A / B fragment of v_mfma_f32_16x16x16_bf16.  M is split over gridDim.z; partial
// results are added to the fp32 output with global float atomics (64-B segments).
//
// Autograd call sites replaced (reference train.py:138 backward through):
//   nn.Linear weights     model.py:79,81,146-151   P = grad_out [M][N], Q = layer input [M][K]
//   patch_embed weight    model.py:215,268          P = grad tokens, Q = 8x8 patches of feat (gather)
//   patch_unembed weight  model.py:225,302          P = tokens, Q = 8x8 patches of grad map (gather)
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

enum { OP_BF16 = 0, OP_F32 = 1, OP_PATCH = 2 };

struct WgradParams {
    const void* P; int ldp;      // [M][NI] bf16 or fp32
    const void* Q; int ldq;      // [M][NJ] bf16 / fp32, or NHWC bf16 map for OP_PATCH
    float* out; int ldo;         // [NI][NJ] fp32, accumulated
    float* colsum_out;           // optional [NI] fp32, accumulated: column sums of P (the layer's bias gradient)
    int M, NI, NJ, mchunk;
    int H, W, Ht, Wt_, nWx, nWy, reflect;   // OP_PATCH geometry (token rows in window layout)
    int linear;                             // OP_PATCH: token rows are a plain [B][Ht][Wt] grid (ResidualTransformer)
    int xcd_remap;
};

TUP_DEVICE s16x4 lds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <int PMODE, int QMODE>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 64 * 128];   // [buf][P|Q][64 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in launch order, and
    // the gridDim.x * gridDim.y blocks of one M slice all read the same rows of P and Q: in launch order those blocks
    // land on all 8 XCDs and every L2 fetches the slice for itself.  Renumber so that XCD k works through a contiguous
    // range of logical blocks (same M slice = same XCD).
    const int T = gridDim.x * gridDim.y * gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int Lp = L;
    if (p.xcd_remap) {
        const int xcd = L & 7, slot = L >> 3, base = T >> 3, rem = T & 7;
        Lp = xcd * base + min(xcd, rem) + slot;
    }
    const int bx = Lp % gridDim.x, by = (Lp / gridDim.x) % gridDim.y, bz = Lp / (gridDim.x * gridDim.y);
    const int i0 = bx * 64, j0 = by * 64;
    const int mbeg = bz * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    if (mbeg >= mend) return;
    const int nsteps = (mend - mbeg + 63) / 64;

    // staging: thread -> chunk (tid&7) of rows (tid>>3) and (tid>>3)+32
    const int chunk = tid & 7;
    // two register sets: tile s+2 is requested while tile s is multiplied and tile s+1 waits in the other set -- with one set
    // a step (8 MFMAs per wave) was shorter than the global round trip it had to cover
    u32x4 pregs[2][(PMODE == OP_F32) ? 4 : 2], qregs[2][(QMODE == OP_F32) ? 4 : 2];
    unsigned okm[2] = {0u, 0u};              // validity of a set's four loads: bit 2u = P row u, bit 2u+1 = Q row u

    // Loads are unconditional (clamped address, value zeroed afterwards): a predicated load compiles to a branch around it,
    // and with branches in the stream hipcc falls back to s_waitcnt vmcnt(0) everywhere -- which makes the wait for the
    // older register set also a wait for the set requested a moment ago.
    // (the zeroing happens at the LDS store, a full step later: a select right behind the load would wait for it at once)
    auto load_plain = [&](const void* base, int ld, int col0, int m, bool f32, u32x4* r) -> bool {
        const bool ok = m < mend;
        const int mc = ok ? m : mend - 1;
        if (!f32) {
            r[0] = *reinterpret_cast<const u32x4*>((const bf16_t*)base + (size_t)mc * ld + col0 + chunk * 8);
        } else {
            const float* s = (const float*)base + (size_t)mc * ld + col0 + chunk * 8;
            r[0] = *reinterpret_cast<const u32x4*>(s);
            r[1] = *reinterpret_cast<const u32x4*>(s + 4);
        }
        return ok;
    };
    auto load_patch = [&](int m, u32x4* r) -> bool {
        // column block j0 = patch pixel (i, j); 64 channels
        bool ok = m < mend;
        int b, ty, tx;
        if (p.linear) {
            const int per = p.Ht * p.Wt_;
            b = m / per;
            const int rem = m - b * per;
            ty = rem / p.Wt_; tx = rem - ty * p.Wt_;
        } else {
            const int tok = m & 63;
            int win = m >> 6;
            const int wx = win % p.nWx; win /= p.nWx;
            const int wy = win % p.nWy;
            b = win / p.nWy;
            ty = wy * 8 + (tok >> 3); tx = wx * 8 + (tok & 7);
        }
        ok = ok && ty < p.Ht && tx < p.Wt_;
        const int pix = j0 >> 6;
        int py = ty * 8 + (pix >> 3), px = tx * 8 + (pix & 7);
        if (p.reflect) {
            if (py >= p.H) py = 2 * p.H - 2 - py;
            if (px >= p.W) px = 2 * p.W - 2 - px;
        } else {
            ok = ok && py < p.H && px < p.W;
        }
        const size_t off = ok ? (((size_t)b * p.H + py) * p.W + px) * 64 : 0;           // pixel 0 of the map when masked
        r[0] = *reinterpret_cast<const u32x4*>((const bf16_t*)p.Q + off + chunk * 8);
        return ok;
    };
    auto load_stage = [&](int s, u32x4* preg, u32x4* qreg, unsigned& ok) {
        ok = 0u;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = mbeg + s * 64 + (tid >> 3) + 32 * u;
            constexpr int PS = (PMODE == OP_F32) ? 2 : 1, QS = (QMODE == OP_F32) ? 2 : 1;
            if (load_plain(p.P, p.ldp, i0, m, PMODE == OP_F32, &preg[u * PS])) ok |= 1u << (2 * u);
            bool qok;
            if constexpr (QMODE == OP_PATCH) qok = load_patch(m, &qreg[u * QS]);
            else qok = load_plain(p.Q, p.ldq, j0, m, QMODE == OP_F32, &qreg[u * QS]);
            if (qok) ok |= 2u << (2 * u);
        }
    };
    auto cvt = [&](const u32x4* r, bool f32) -> u32x4 {
        if (!f32) return r[0];
        const f32x4 lo = __builtin_bit_cast(f32x4, r[0]), hi = __builtin_bit_cast(f32x4, r[1]);
        return u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
    };
    auto store_stage = [&](int buf, const u32x4* preg, const u32x4* qreg, unsigned ok) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (tid >> 3) + 32 * u;
            constexpr int PS = (PMODE == OP_F32) ? 2 : 1, QS = (QMODE == OP_F32) ? 2 : 1;
            char* base = smem + buf * (2 * 64 * 128);
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(base + swz128(row, chunk)) = ((ok >> (2 * u)) & 1u) ? cvt(&preg[u * PS], PMODE == OP_F32) : z;
            *reinterpret_cast<u32x4*>(base + 64 * 128 + swz128(row, chunk)) = ((ok >> (2 * u + 1)) & 1u) ? cvt(&qreg[u * QS], QMODE == OP_F32) : z;
        }
    };

    f32x4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of P: the workgroups of the first column block also multiply their P fragments by a
    // ones operand (one more MFMA per K-step there) instead of a separate pass over P (tup_colsum)
    const bool want_cs = p.colsum_out != nullptr && by == 0;
    f32x4 accs = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // transposed-read addressing: lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3
    const int trq = l16 >> 2, trp = l16 & 3;

    load_stage(0, pregs[0], qregs[0], okm[0]);
    load_stage(min(1, nsteps - 1), pregs[1], qregs[1], okm[1]);
    store_stage(0, pregs[0], qregs[0], okm[0]);
    __syncthreads();
    // step s: LDS buffer s & 1 holds tile s; register set (s + 1) & 1 holds tile s + 1; set s & 1 is free
    auto step = [&](const int s, auto setc) {
        constexpr int FREE = decltype(setc)::value;          // = s & 1
        const int buf = s & 1;
        // unconditional (the last two steps re-request the last tile): a load under a branch makes hipcc assume the
        // smaller in-flight count at the join and wait for the new loads as well
        load_stage(min(s + 2, nsteps - 1), pregs[FREE], qregs[FREE], okm[FREE]);
        const char* pb = smem + buf * (2 * 64 * 128);
        const char* qb = pb + 64 * 128;
        // two K-steps of 32 rows on v_mfma_f32_16x16x32_bf16 (the 16x16x16 form runs at half its rate on gfx950):
        // lane group g contracts rows 8g .. 8g+7, delivered by two transposed reads of 4 rows each
        auto join = [](s16x4 lo, s16x4 hi) {
            const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
            return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = 32 * ks + 8 * g + trq;
            const int pcol = 16 * wave + 4 * trp;
            const bf16x8 af = join(lds_read_tr16(pb + swz128(row, pcol >> 3) + (pcol & 7) * 2),
                                   lds_read_tr16(pb + swz128(row + 4, pcol >> 3) + (pcol & 7) * 2));
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int qcol = 16 * jt + 4 * trp;
                const bf16x8 bfr = join(lds_read_tr16(qb + swz128(row, qcol >> 3) + (qcol & 7) * 2),
                                        lds_read_tr16(qb + swz128(row + 4, qcol >> 3) + (qcol & 7) * 2));
                acc[jt] = mfma16x16x32(af, bfr, acc[jt]);
            }
            if (want_cs) accs = mfma16x16x32(af, ones, accs);
        }
        if (s + 1 < nsteps) store_stage(buf ^ 1, pregs[FREE ^ 1], qregs[FREE ^ 1], okm[FREE ^ 1]);
        __syncthreads();
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>{});
    }

    // D[row = i 4g+e][col = j l16]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#ifndef TUP_EXP_NOATOMIC      // timing experiment (wrong results)
            atomicAdd(p.out + (size_t)(i0 + 16 * wave + 4 * g + e) * p.ldo + j0 + 16 * jt + l16, acc[jt][e]);
#else
            if (acc[jt][e] == 123.456f) p.out[0] = 1.f;
#endif
    if (want_cs && l16 == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.colsum_out + i0 + 16 * wave + 4 * g + e, accs[e]);
    }
}

// column sums: out[n] += sum_m G[m][n].  16-byte loads: LPR lanes cover a 64-column stripe of one row, the
// block's other lanes take further rows; grid.x = column stripes, grid.y splits M.
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ G, int ld, float* __restrict__ out, int M, int N, int mchunk,
                                                     const unsigned char* __restrict__ rowmask)
{
    constexpr int EPL = F32 ? 4 : 8;            // elements per lane per load
    constexpr int LPR = 64 / EPL;               // lanes per row stripe
    constexpr int RPB = 256 / LPR;              // rows per block iteration
    const int lc = threadIdx.x % LPR, lr = threadIdx.x / LPR;
    const int n = blockIdx.x * 64 + lc * EPL;
    const int mbeg = blockIdx.y * mchunk, mend = min(M, mbeg + mchunk);
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    int m = mbeg + lr;
    if constexpr (!F32) if (rowmask == nullptr) {
        // the 472 MB bias-gradient sums of the 64-channel maps: four rows in flight per lane (one dependent load per iteration ran the
        // stream at 3.9 TB/s)
        const bf16_t* gp = (const bf16_t*)G;
        for (; m + 3 * RPB < mend; m += 4 * RPB) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(gp + (size_t)(m + u * RPB) * ld + n);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] += __builtin_bit_cast(float, v[u][q] << 16);
                    acc[2 * q + 1] += __builtin_bit_cast(float, v[u][q] & 0xffff0000u);
                }
        }
    }
    for (; m < mend; m += RPB) {
        if (rowmask && !rowmask[m]) continue;
        if constexpr (F32) {
            const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)G + (size_t)m * ld + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        } else {
            const u32x4 v = *reinterpret_cast<const u32x4*>((const bf16_t*)G + (size_t)m * ld + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[2 * q] += __builtin_bit_cast(float, v[q] << 16);
                acc[2 * q + 1] += __builtin_bit_cast(float, v[q] & 0xffff0000u);
            }
        }
    }
    __shared__ float red[RPB][64 + 1];
#pragma unroll
    for (int e = 0; e < EPL; ++e) red[lr][lc * EPL + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int r = 0; r < RPB; ++r) s += red[r][threadIdx.x];
        atomicAdd(out + blockIdx.x * 64 + threadIdx.x, s);
    }
}

// ------------------------------------------------------------------------------------------------
// Wide-tile form for the two patch weights (NI = 192 or 128 token features x NJ = 4096 patch elements, M = every token).
// gemm_wgrad_kernel above gives a workgroup a 64 x 64 output tile, so the fp32 token operand is fetched 64 times and the 472 MB
// map three times: 4.5 GB through L2 per launch for 0.52 GB of operands, and the launch runs at that L2 rate (340 us).  Here
// ONE workgroup per CU owns a 192 x (64 JP) tile -- all token features x JP patch pixels: the map is read once, the tokens
// 4096 / (64 JP) times.  64-row stages of both operands go to LDS by DMA (global_load_lds; swizzle, patch gather, reflect /
// zero padding and the ragged end of the M slice on the source side: no staging registers, which is what the 192-row form of
// round 2 died of) into a ring of NST stages with one raw barrier per stage; NW = 4 or 8 waves, each (384 / NW) x (32 JP) of the
// tile: per 32 rows of M a wave reads 24 / NW + 2 JP transposed fragments (asm, hand-counted lgkmcnt) for 48 JP / NW MFMAs.  The token operand must
// be bf16 in memory (DMA does not convert).
// Measured and NOT used for the Linear layers of the blocks (192 x {192, 576, 768}, JP = 3): the loop runs at 1.7 us per 48 KB
// stage and CU (7 TB/s from cache-resident operands), but every workgroup ends with its whole tile as float atomics -- 256
// workgroups x 36,864 = 9.4 M per launch whatever the layer, and the L2's atomic units take about one per clock and channel:
// 36 us on top of a 7-26 us loop (43-63 us against 22-47 us for the 64 x 64 tiles, whose atomics hide behind the other
// resident workgroups).  One image per XCD (no line migrating between L2s) + a reduce launch measured the same.  With M
// slices of 7,680 rows (patch weights) the same epilogue is 10 % of the launch.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned int tup_wg_zero_line[4] = {0u, 0u, 0u, 0u};

template <int JP, int NST, bool PATCH, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gemm_wgrad_wide_kernel(const WgradParams p)
{
    constexpr int IP = 3;                              // 64-column panels of P per workgroup
    constexpr int STAGE = (IP + JP) * 8192;            // a panel stage = 64 rows x 128 B
    constexpr int RU = 8 / NW;                         // 32-row (NW = 4) or 64-row (NW = 8) groups of a stage per DMA instruction
    constexpr int PER = (IP + JP) * RU;                // DMA instructions per thread and stage
    constexpr int IT = 24 / NW;                        // 16-row tiles of the wave's i range: NW / 2 waves along i, two along j
    constexpr int JT = 2 * JP;                         // 16-column tiles of the wave's j range
    static_assert((NST - 2) * PER <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int wi = wave % (NW / 2), wj = wave / (NW / 2);

    const int nib = p.NI / 192, nij = nib * (p.NJ / (64 * JP));
    const int T = gridDim.x, L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3, base = T >> 3, rem = T & 7;
    const int Lp = xcd * base + min(xcd, rem) + slot;          // XCD k works through a contiguous range: same M slice = same L2
    const int ij = Lp % nij, mz = Lp / nij;
    const int i0 = (ij % nib) * 192, j0 = (ij / nib) * (64 * JP);
    const int mbeg = mz * p.mchunk, mend = min(p.M, mbeg + p.mchunk);
    if (mbeg >= mend) return;
    const int nsteps = (mend - mbeg + 63) / 64;

    // DMA bookkeeping: a thread moves chunk slot tid & 7 of rows (tid >> 3) and (tid >> 3) + 32 of every panel; the slot holds
    // logical chunk dc (swz128's XOR, applied on the source side)
    const int dr = tid >> 3, dc = (tid & 7) ^ ((tid >> 4) & 7);
    const char* psrc = (const char*)p.P + ((size_t)i0 + dc * 8) * 2;
    const char* qsrc = (const char*)p.Q + (PATCH ? (size_t)dc * 16 : ((size_t)j0 + dc * 8) * 2);
    auto dma_stage = [&](int s, int buf) {
        char* dst = dsm + buf * STAGE + wave * 1024;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int m = mbeg + s * 64 + dr + 32 * u;
            const bool ok = s < nsteps && m < mend;
            const char* prow = psrc + (size_t)m * p.ldp * 2;
#pragma unroll
            for (int pn = 0; pn < IP; ++pn)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? prow + pn * 128 : (const char*)tup_wg_zero_line),
                                                 (__attribute__((address_space(3))) void*)(dst + pn * 8192 + u * 4096), 16, 0, 0);
            if constexpr (!PATCH) {
                const char* qrow = qsrc + (size_t)m * p.ldq * 2;
#pragma unroll
                for (int pn = 0; pn < JP; ++pn)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? qrow + pn * 128 : (const char*)tup_wg_zero_line),
                                                     (__attribute__((address_space(3))) void*)(dst + (IP + pn) * 8192 + u * 4096), 16, 0, 0);
            } else {
                // panel pn of the tile = patch pixel (j0 >> 6) + pn, 64 channels (see load_patch of the narrow kernel)
                int b, ty, tx;
                if (p.linear) {
                    const int per = p.Ht * p.Wt_;
                    b = m / per;
                    const int r2 = m - b * per;
                    ty = r2 / p.Wt_; tx = r2 - ty * p.Wt_;
                } else {
                    const int tok = m & 63;
                    int win = m >> 6;
                    const int wx = win % p.nWx; win /= p.nWx;
                    const int wy = win % p.nWy;
                    b = win / p.nWy;
                    ty = wy * 8 + (tok >> 3); tx = wx * 8 + (tok & 7);
                }
                const bool tok_ok = ok && ty < p.Ht && tx < p.Wt_;
#pragma unroll
                for (int pn = 0; pn < JP; ++pn) {
                    const int pix = (j0 >> 6) + pn;
                    int py = ty * 8 + (pix >> 3), px = tx * 8 + (pix & 7);
                    bool pok = tok_ok;
                    if (p.reflect) {
                        if (py >= p.H) py = 2 * p.H - 2 - py;
                        if (px >= p.W) px = 2 * p.W - 2 - px;
                    } else {
                        pok = pok && py < p.H && px < p.W;
                    }
                    const char* src = pok ? qsrc + (((size_t)b * p.H + py) * p.W + px) * 128 : (const char*)tup_wg_zero_line;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + (IP + pn) * 8192 + u * 4096), 16, 0, 0);
                }
            }
        }
    };

    f32x4 acc[IT][JT];
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_cs = p.colsum_out != nullptr && j0 == 0 && wj == 0;
    f32x4 accs[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) accs[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // transposed-read addressing (as above): lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of a tile.
    // The reads are asm statements (ds_read_b64_tr_b16, hand-counted lgkmcnt): behind a compiler-visible LDS read hipcc puts
    // s_waitcnt vmcnt(0), i.e. a wait for the stages just requested.
    const int trq = l16 >> 2, trp = l16 & 3;
    auto frag_off = [&](int panel0, int col, int h) -> uint32_t {          // col = first column of the 16-wide tile in its panel group
        const int c = col + 4 * trp, pn = panel0 + (c >> 6), pc = c & 63;
        return (uint32_t)(pn * 8192 + (pc & 7) * 2 + swz128(8 * g + trq + 4 * h, pc >> 3));
    };
    uint32_t aoff[IT][2], boff[JT][2];
#pragma unroll
    for (int it = 0; it < IT; ++it) { aoff[it][0] = frag_off(0, 16 * IT * wi + 16 * it, 0); aoff[it][1] = frag_off(0, 16 * IT * wi + 16 * it, 1); }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) { boff[jt][0] = frag_off(IP, 32 * JP * wj + 16 * jt, 0); boff[jt][1] = frag_off(IP, 32 * JP * wj + 16 * jt, 1); }
    auto rd = [](uint32_t addr, auto ksc) {
        u32x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(decltype(ksc)::value * 4096));
        return v;
    };
    auto join = [](u32x2 lo, u32x2 hi) { return __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]}); };
    const uint32_t lds0 = lds_addr(dsm);

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) dma_stage(s, s);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        // own pieces of stage s, then everyone's; everyone is done with stage s - 1 (every LDS read of it was waited for).  A raw
        // barrier: __syncthreads() carries a fence, which hipcc turns into vmcnt(0) -- a wait for the stages still in flight
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 2) * PER) : "memory");
        dma_stage(s + NST - 1, buf == 0 ? NST - 1 : buf - 1);                         // into the buffer stage s - 1 used (past the end: zero lines)
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t sb = lds0 + (uint32_t)(buf * STAGE);
        auto kstep = [&](auto ksc) {
            u32x2 ar[IT][2], br[2][2];
#pragma unroll
            for (int it = 0; it < IT; ++it) { ar[it][0] = rd(sb + aoff[it][0], ksc); ar[it][1] = rd(sb + aoff[it][1], ksc); }
            br[0][0] = rd(sb + boff[0][0], ksc); br[0][1] = rd(sb + boff[0][1], ksc);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt + 1 < JT) {
                    br[(jt + 1) & 1][0] = rd(sb + boff[jt + 1][0], ksc); br[(jt + 1) & 1][1] = rd(sb + boff[jt + 1][1], ksc);
                    lds_wait<2>();
                } else {
                    lds_wait<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 bfr = join(br[jt & 1][0], br[jt & 1][1]);
#pragma unroll
                for (int it = 0; it < IT; ++it) acc[it][jt] = mfma16x16x32(join(ar[it][0], ar[it][1]), bfr, acc[it][jt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (want_cs) {
#pragma unroll
                for (int it = 0; it < IT; ++it) accs[it] = mfma16x16x32(join(ar[it][0], ar[it][1]), ones, accs[it]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-line requests past the end

    // D[row = i 4g+e][col = j l16]
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                atomicAdd(p.out + (size_t)(i0 + 16 * IT * wi + 16 * it + 4 * g + e) * p.ldo + j0 + 32 * JP * wj + 16 * jt + l16, acc[it][jt][e]);
    if (want_cs && l16 == 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(p.colsum_out + i0 + 16 * IT * wi + 16 * it + 4 * g + e, accs[it][e]);
    }
}

template <int JP, int NST, bool PATCH, int NW>
int launch_wide(WgradParams p, hipStream_t s)
{
    if (p.M <= 0) return 0;
    if (p.NI % 192 || p.NJ % (64 * JP)) return (int)hipErrorInvalidValue;
    constexpr size_t lds = (size_t)NST * (3 + JP) * 8192;
    TUP_SET_DYN_LDS((gemm_wgrad_wide_kernel<JP, NST, PATCH, NW>), lds);
    const int nij = (p.NI / 192) * (p.NJ / (64 * JP));
    // one workgroup per CU (the LDS ring); M slices of whole 64-row stages
    int msplit = 256 / nij;
    if (msplit < 1) msplit = 1;
    p.mchunk = (((p.M + msplit - 1) / msplit) + 63) / 64 * 64;
    msplit = (p.M + p.mchunk - 1) / p.mchunk;
    gemm_wgrad_wide_kernel<JP, NST, PATCH, NW><<<dim3(nij * msplit), dim3(64 * NW), lds, s>>>(p);
    TUP_CHECK_LAUNCH();
    return 0;
}

template <int PMODE, int QMODE>
int launch(WgradParams p, hipStream_t s)
{
    if (p.M <= 0) return 0;
    if (p.NI % 64 || p.NJ % 64) return (int)hipErrorInvalidValue;
    const int blocks_ij = (p.NI / 64) * (p.NJ / 64);
    // Five workgroups per CU are resident (32 KB of LDS each, 52 registers): any grid up to 1,280 workgroups is one round; the
    // split is rounded DOWN so that the M slices never come out shorter than the target (more slices = more fp32 atomics on the
    // same 64 x 64 output tiles; the kernel is bound by L2 reads of its operands)
    int msplit = 1024 / blocks_ij;
    const int maxsplit = (p.M + 63) / 64;
    if (msplit > maxsplit) msplit = maxsplit;
    if (msplit < 1) msplit = 1;
    p.mchunk = (((p.M + msplit - 1) / msplit) + 63) / 64 * 64;
    msplit = (p.M + p.mchunk - 1) / p.mchunk;
    static const int remap = TUP_ENV_FLAG("TUP_WGRAD_NO_XCD") ? 0 : 1;            // A/B switch
    p.xcd_remap = remap;
    gemm_wgrad_kernel<PMODE, QMODE><<<dim3(p.NI / 64, p.NJ / 64, msplit), dim3(256), 0, s>>>(p);
    TUP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// out[NI][NJ] (fp32, ldo) += P^T Q and, when colsum_out != NULL, colsum_out[NI] += column sums of P (weight and bias
// gradient of a Linear layer in one pass over grad_out).  p_dtype / q_dtype: 0 = bf16, 1 = fp32.  The caller zeroes both.
extern "C" int tup_gemm_wgrad_bias(const void* P, int p_dtype, int ldp, const void* Q, int q_dtype, int ldq,
                                   float* out, int ldo, float* colsum_out, int M, int NI, int NJ, void* stream);

// out[NI][NJ] (fp32, ldo) += P^T Q.  p_dtype / q_dtype: 0 = bf16, 1 = fp32.  The caller zeroes `out`.
extern "C" int tup_gemm_wgrad(const void* P, int p_dtype, int ldp, const void* Q, int q_dtype, int ldq,
                              float* out, int ldo, int M, int NI, int NJ, void* stream)
{
    return tup_gemm_wgrad_bias(P, p_dtype, ldp, Q, q_dtype, ldq, out, ldo, nullptr, M, NI, NJ, stream);
}

extern "C" int tup_gemm_wgrad_bias(const void* P, int p_dtype, int ldp, const void* Q, int q_dtype, int ldq,
                                   float* out, int ldo, float* colsum_out, int M, int NI, int NJ, void* stream)
{
    WgradParams p{};
    p.P = P; p.ldp = ldp; p.Q = Q; p.ldq = ldq; p.out = out; p.ldo = ldo; p.M = M; p.NI = NI; p.NJ = NJ;
    p.colsum_out = colsum_out;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p_dtype == 0 && q_dtype == 0) return launch<OP_BF16, OP_BF16>(p, s);
    if (p_dtype == 1 && q_dtype == 0) return launch<OP_F32, OP_BF16>(p, s);
    if (p_dtype == 0 && q_dtype == 1) return launch<OP_BF16, OP_F32>(p, s);
    if (p_dtype == 1 && q_dtype == 1) return launch<OP_F32, OP_F32>(p, s);
    return (int)hipErrorInvalidValue;
}

// out[192][4096] += P^T patches(map): P fp32 [M][192] token rows (window layout), map NHWC bf16
// [B][H][W][64]; column (i*8+j)*64 + c.  reflect = 1 for patch_embed's weight (reflect-padded
// forward input), 0 for patch_unembed's weight (cropped output -> zero outside).
extern "C" int tup_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W,
                               int reflect, void* stream)
{
    WgradParams p{};
    p.H = H; p.W = W; p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8; p.reflect = reflect;
    p.P = P; p.ldp = 192; p.Q = map; p.out = out; p.ldo = 4096;
    p.M = B * p.nWy * p.nWx * 64; p.NI = 192; p.NJ = 4096;
    return launch<OP_F32, OP_PATCH>(p, reinterpret_cast<hipStream_t>(stream));
}

// The same from bf16 token rows (P bf16 [M][192]: the rounding the fp32 form applies on load, done by the caller) on the wide-tile
// kernel: one pass over the map.
extern "C" int tup_patch_wgrad_bf16(const void* P, const void* map, float* out, int B, int H, int W, int reflect, void* stream)
{
    WgradParams p{};
    p.H = H; p.W = W; p.Ht = (H + 7) / 8; p.Wt_ = (W + 7) / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8; p.reflect = reflect;
    p.P = P; p.ldp = 192; p.Q = map; p.out = out; p.ldo = 4096;
    p.M = B * p.nWy * p.nWx * 64; p.NI = 192; p.NJ = 4096;
    if ((long long)B * H * W * 128 >= (1LL << 31)) return (int)hipErrorInvalidValue;          // 32-bit pixel offsets in the gather
    static const int form = TUP_ENV_INT("TUP_PATCH_WGRAD_FORM", 0);          // tuning knob (diagnostic build)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // 4 x 720p, incl. the zero fill and the caller's bf16 cast (64 x 64 tiles: 350-390 us): 192 x 256 tile / 2 stages / 8 waves
    // 177-182 us; 192 x 128 / 3 stages / 8 waves 194-214; 192 x 256 / 2 / 4 waves 260-280; 192 x 128 / 3 / 4 waves 285-300 -- one wave
    // per SIMD hides neither its gather arithmetic nor its LDS waits
#ifdef TUP_DIAG          // the other forms exist in `make diag` only
    if (form == 1) return launch_wide<2, 3, true, 4>(p, st);
    if (form == 2) return launch_wide<4, 2, true, 4>(p, st);
    if (form == 3) return launch_wide<2, 3, true, 8>(p, st);
#endif
    (void)form;
    return launch_wide<4, 2, true, 8>(p, st);
}

// ResidualTransformer's patch_embed / patch_unembed weights: out[128][4096] += P^T patches(map), P fp32 [B*T][128]
// on the plain token grid (H, W multiples of 8).
extern "C" int tup_rt_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W, void* stream)
{
    if (H % 8 || W % 8) return (int)hipErrorInvalidValue;
    WgradParams p{};
    p.H = H; p.W = W; p.Ht = H / 8; p.Wt_ = W / 8; p.linear = 1; p.reflect = 0;
    p.P = P; p.ldp = 128; p.Q = map; p.out = out; p.ldo = 4096;
    p.M = B * p.Ht * p.Wt_; p.NI = 128; p.NJ = 4096;
    return launch<OP_F32, OP_PATCH>(p, reinterpret_cast<hipStream_t>(stream));
}

// WindowTransformer's patch_embed / patch_unembed weights: out[NI][4096] += P^T patches(map), P fp32 [M][NI] token rows in
// window layout over the floor(H/8) x floor(W/8) token grid (no reflect padding: pixels beyond the grid do not exist).
extern "C" int tup_wt_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W, int NI, void* stream)
{
    if (NI % 64 || H < 8 || W < 8) return (int)hipErrorInvalidValue;
    WgradParams p{};
    p.H = H; p.W = W; p.Ht = H / 8; p.Wt_ = W / 8;
    p.nWy = (p.Ht + 7) / 8; p.nWx = (p.Wt_ + 7) / 8; p.reflect = 0;
    p.P = P; p.ldp = NI; p.Q = map; p.out = out; p.ldo = 4096;
    p.M = B * p.nWy * p.nWx * 64; p.NI = NI; p.NJ = 4096;
    return launch<OP_F32, OP_PATCH>(p, reinterpret_cast<hipStream_t>(stream));
}

// out[N] += column sums of G [M][N] (bias gradients); rowmask (uint8 [M], may be NULL) selects the rows
// that count (patch_embed's bias does not reach the zero-padded tokens).
extern "C" int tup_colsum(const void* G, int dtype, int ld, float* out, int M, int N, const void* rowmask, void* stream)
{
    if (M <= 0 || N <= 0) return 0;
    if (N % 64 != 0 || ld % 8 != 0) return (int)hipErrorInvalidValue;
    // every workgroup ends with 64 float atomics on the same addresses (they serialise in L2): a few hundred workgroups, not thousands
    // (the 64-column map sums: 2,048 workgroups 106 us, 512 95 us; three column stripes of token rows are flat from 512 to 2,048)
    static const int forced = TUP_ENV_INT("TUP_COLSUM_BLOCKS", 0);
    const int target = forced > 0 ? forced : (N == 64 ? 512 : 2048);
    int msplit = target / (N / 64);
    if (msplit < 1) msplit = 1;
    int mchunk = (M + msplit - 1) / msplit;
    if (mchunk < 256) mchunk = 256;
    msplit = (M + mchunk - 1) / mchunk;
    dim3 grid(N / 64, msplit);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 1) colsum_kernel<true><<<grid, dim3(256), 0, s>>>(G, ld, out, M, N, mchunk, (const unsigned char*)rowmask);
    else colsum_kernel<false><<<grid, dim3(256), 0, s>>>(G, ld, out, M, N, mchunk, (const unsigned char*)rowmask);
    TUP_CHECK_LAUNCH();
    return 0;
}
