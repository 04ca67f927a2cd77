// Backward of LayerNorm and of the window-attention core (gfx950).  Replaces what autograd runs for
// reference models/FastTransformer/model.py:114-130 (attention) and :142,144,163,169 (LayerNorm) under
// train.py:138.
//
// Attention backward: one wave per (window, head), persistent over windows so that the per-head
// dS sum (the relative-position-bias gradient) accumulates in registers.  Nothing N x N is stored
// by the forward: P is recomputed from q, k, the dense bias and the row's log-sum-exp, which the
// training forward saves (64 floats per window and head); rowsum(P * dP) is dO . O, from the saved
// attention output.  With both row statistics known up front the 64 x 64 tile is swept ONCE, in the
//   N-layout (rows = query, cols = key):  dV^T = dO^T P,  dK^T = Q^T dS,  dBias += dS
// and dQ^T = K^T dS^T takes its B operand from the same dS tiles, written to the wave's LDS as they are
// produced and read back transposed (ds_read_b64_tr_b16).  (The first form swept the tile twice -- a
// T-layout pass for the statistics and dQ, then the N-layout pass -- with two exp and two dropout hashes
// per score and 22 dependent cross-lane shuffles per window: 208 us at 960 windows, one wave per SIMD.)
#include "common.h"

namespace {

// two 4-element fragments -> one 16x16x32 operand (x32 runs at twice the rate of the x16 MFMA on gfx950); both operands
// of a product use the same k map, so any two 16-wide slices of the contracted axis can be paired
TUP_DEVICE bf16x8 join4(s16x4 lo, s16x4 hi) {
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}


constexpr int HD = 16, NTOK = 64;      // heads / width are template parameters (12 x 16 = 192, or 8 x 16 = 128)

template <int NQ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const bf16_t* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ gres,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
    bf16_t* __restrict__ gdrop, uint32_t drop_thresh, float drop_inv_keep, uint32_t drop_seed)
{
    constexpr int LD = NQ * 64;              // 192 (FastTransformer) or 128 (ResidualTransformer)
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    f32x4 gm[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) gm[q] = *reinterpret_cast<const f32x4*>(gamma + q * 64 + sub * 4);
    float dg[NQ * 4], db[NQ * 4];
#pragma unroll
    for (int i = 0; i < NQ * 4; ++i) { dg[i] = 0.f; db[i] = 0.f; }

    for (int row0 = blockIdx.x * 16; row0 < M; row0 += gridDim.x * 16) {
        const int row = row0 + slot;
        const bool ok = row < M;
        const int r = ok ? row : 0;
        const float mu = mean[r], rs = rstd[r];
        // every load of the row stands ahead of the arithmetic (x, gy and the residual-path gradient that is only added at the
        // end): the kernel is a 165 MB stream per launch, its speed is the number of loads a wave keeps in flight
        f32x4 xq[NQ], rq[NQ];
        u32x2 gq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = q * 64 + sub * 4;
            xq[q] = *reinterpret_cast<const f32x4*>(x + (size_t)r * LD + c);
            gq[q] = *reinterpret_cast<const u32x2*>(gy + (size_t)r * LD + c);
            rq[q] = gres ? *reinterpret_cast<const f32x4*>(gres + (size_t)r * LD + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float xh[NQ * 4], gg[NQ * 4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 xv = xq[q];
            const u32x2 gv = gq[q];
            const float gyv[4] = {__builtin_bit_cast(float, gv[0] << 16), __builtin_bit_cast(float, gv[0] & 0xffff0000u),
                                  __builtin_bit_cast(float, gv[1] << 16), __builtin_bit_cast(float, gv[1] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = q * 4 + e;
                xh[i] = (xv[e] - mu) * rs;
                gg[i] = gyv[e] * gm[q][e];
                s1 += gg[i];
                s2 += gg[i] * xh[i];
                if (ok) { dg[i] += gyv[e] * xh[i]; db[i] += gyv[e]; }
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 *= (1.0f / LD); s2 *= (1.0f / LD);
        if (ok) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * 64 + sub * 4;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gg[q * 4 + e] - s1 - xh[q * 4 + e] * s2);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rq[q][e];
                *reinterpret_cast<f32x4*>(dx + (size_t)row * LD + c) = o;
                if (gdrop) {       // the gradient's next stop is the Dropout behind the previous Linear (model.py:82,132,150): dx * mask / keep
                    const uint32_t e0 = (uint32_t)row * LD + c;
                    *reinterpret_cast<u32x2*>(gdrop + (size_t)row * LD + c) = u32x2{
                        pack_bf16x2(o[0] * drop_scale(drop_seed, e0, drop_thresh, drop_inv_keep), o[1] * drop_scale(drop_seed, e0 + 1, drop_thresh, drop_inv_keep)),
                        pack_bf16x2(o[2] * drop_scale(drop_seed, e0 + 2, drop_thresh, drop_inv_keep), o[3] * drop_scale(drop_seed, e0 + 3, drop_thresh, drop_inv_keep))};
                }
            }
        }
    }
    // reduce the 16 row slots of the block, then one atomic per column
    __shared__ float red[2][16][LD];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][slot][q * 64 + sub * 4 + e] = dg[q * 4 + e];
            red[1][slot][q * 64 + sub * 4 + e] = db[q * 4 + e];
        }
    __syncthreads();
    if (threadIdx.x < LD) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) { a += red[0][s][threadIdx.x]; b += red[1][s][threadIdx.x]; }
        atomicAdd(dgamma + threadIdx.x, a);
        atomicAdd(dbeta + threadIdx.x, b);
    }
}

// N-layout copy of the dense bias: fragN[h][qt][kt][lane][e] = bias(query 16qt+4g+e, key 16kt+l16)
template <int HEADS>
__global__ void relpos_expand_n_kernel(const float* __restrict__ table, float* __restrict__ frag)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HEADS * 16 * 256) return;
    const int e = idx & 3, lane = (idx >> 2) & 63, kt = (idx >> 8) & 3, qt = (idx >> 10) & 3, h = idx >> 12;
    const int qi = 16 * qt + 4 * (lane >> 4) + e, kj = 16 * kt + (lane & 15);
    const int rel = ((qi >> 3) - (kj >> 3) + 7) * 15 + ((qi & 7) - (kj & 7) + 7);
    frag[idx] = table[rel * HEADS + h];
}

// dtable[rel][h] = sum over (query, key) pairs with that relative offset of the dense N-layout gradient
// (dfrag[h][qt][kt][lane][e] = d bias(query 16qt + 4(lane >> 4) + e, key 16kt + (lane & 15)), the layout of relpos_expand_n_kernel)
template <int HEADS>
__global__ void relpos_reduce_kernel(const float* __restrict__ dfrag, float* __restrict__ dtable)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 225 * HEADS) return;
    const int h = idx % HEADS, rel = idx / HEADS;
    const int dy = rel / 15 - 7, dx = rel % 15 - 7;      // query - key offsets
    float s = 0.f;
    for (int ky = 0; ky < 8; ++ky) {
        const int qy = ky + dy;
        if (qy < 0 || qy > 7) continue;
        for (int kx = 0; kx < 8; ++kx) {
            const int qx = kx + dx;
            if (qx < 0 || qx > 7) continue;
            const int qi = qy * 8 + qx, kj = ky * 8 + kx;
            const int lane = ((qi >> 2) & 3) * 16 + (kj & 15);
            s += dfrag[((((size_t)h * 4 + (qi >> 4)) * 4 + (kj >> 4)) * 64 + lane) * 4 + (qi & 3)];
        }
    }
    dtable[idx] = s;      // idx = rel*HEADS + h
}

// The LDS tiles are private to a wave and waves of a block run different trip counts, so the
// write->read hand-off is ordered per wave (DS ops of one wave execute in order), not by s_barrier.
TUP_DEVICE void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

TUP_DEVICE s16x4 to_bf16x4(const f32x4 v) {
    const u32x2 p = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    return __builtin_bit_cast(s16x4, p);
}

TUP_DEVICE float dot_bf16x4(const s16x4 a, const s16x4 b) {
    const u32x2 x = __builtin_bit_cast(u32x2, a), y = __builtin_bit_cast(u32x2, b);
    float r = __builtin_bit_cast(float, x[0] << 16) * __builtin_bit_cast(float, y[0] << 16);
    r = __builtin_fmaf(__builtin_bit_cast(float, x[0] & 0xffff0000u), __builtin_bit_cast(float, y[0] & 0xffff0000u), r);
    r = __builtin_fmaf(__builtin_bit_cast(float, x[1] << 16), __builtin_bit_cast(float, y[1] << 16), r);
    return __builtin_fmaf(__builtin_bit_cast(float, x[1] & 0xffff0000u), __builtin_bit_cast(float, y[1] & 0xffff0000u), r);
}

template <int HEADS>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ gout, const bf16_t* __restrict__ att, const float* __restrict__ lse,
    const float* __restrict__ bias_n, bf16_t* __restrict__ gqkv, float* __restrict__ dbias_part, int nwin, int nslots,
    uint32_t drop_thresh, float drop_inv_keep, uint32_t drop_seed)
{
    constexpr int DIM = HEADS * HD;
    typedef short v4i16 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) bf16_t lds[4][3][NTOK * HD];     // per wave: K, Q, dO as [tok][hd]
    __shared__ __attribute__((aligned(16))) bf16_t dsl[4][8][256];           // per wave: dS^T tiles of a query-tile pair, [hh][kt][key 16][query 16]
    __shared__ __attribute__((aligned(16))) float dl[4][NTOK];               // per wave: rowsum(P * dP) = dO . O per query
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    bf16_t* kl = lds[wave][0];
    bf16_t* ql = lds[wave][1];
    bf16_t* dol = lds[wave][2];
    // One wave per SIMD (512 registers): a wave = one (slot, head); the head's dense bias (the same for every window of the
    // persistent wave) and the running dS sum stay in registers.  Measured alternatives at 960 windows, dropout 0.1 (this form:
    // 142 us for the three launches of the backward): two waves per SIMD by letting hipcc spill 71 registers 198 us; two waves
    // per SIMD with the bias and the dS sum in LDS (ds_add_f32 from the four waves of a head-workgroup) 340-410 us.
    const int gw = blockIdx.x * 4 + wave;
    const int h = gw % HEADS, slot = gw / HEADS;
    f32x4 dbacc[4][4], bn[4][4];                 // [qt][kt], N layout
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            dbacc[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bn[qt][kt] = *reinterpret_cast<const f32x4*>(bias_n + ((((size_t)h * 4 + qt) * 4 + kt) * 64 + lane) * 4);
        }

    // Everything a window needs from memory: the K, Q, dO rows of this lane's token (for the LDS tiles the transposed fragments are
    // gathered from), the row fragments [tok 16t+p][d 4g..4g+3] of q, k, v, dO and O, and the log-sum-exp of the lane's 16 query rows.
    // The NEXT window's set is requested before the current window's arithmetic starts: with one wave per SIMD nothing else covers the
    // round trip (26 k cycles per window for ~8 k cycles of issue before this).
    struct WinRegs { u32x4 rows[6]; s16x4 qf[4], kf[4], vf[4], dof[4], of[4]; f32x4 ls[4]; };
    auto load_window = [&](int win) {
        WinRegs r;
        const bf16_t* base = qkv + (size_t)win * NTOK * (3 * DIM) + h * HD;
        const bf16_t* gbase = gout + (size_t)win * NTOK * DIM + h * HD;
        const bf16_t* abase = att + (size_t)win * NTOK * DIM + h * HD;
        const bf16_t* rr = base + (size_t)lane * (3 * DIM);
        const bf16_t* gr = gbase + (size_t)lane * DIM;
        r.rows[0] = *reinterpret_cast<const u32x4*>(rr);           r.rows[1] = *reinterpret_cast<const u32x4*>(rr + 8);
        r.rows[2] = *reinterpret_cast<const u32x4*>(rr + DIM);     r.rows[3] = *reinterpret_cast<const u32x4*>(rr + DIM + 8);
        r.rows[4] = *reinterpret_cast<const u32x4*>(gr);           r.rows[5] = *reinterpret_cast<const u32x4*>(gr + 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t roff = (size_t)(16 * t + p) * (3 * DIM) + 4 * g;
            r.qf[t] = *reinterpret_cast<const s16x4*>(base + roff);
            r.kf[t] = *reinterpret_cast<const s16x4*>(base + roff + DIM);
            r.vf[t] = *reinterpret_cast<const s16x4*>(base + roff + 2 * DIM);
            r.dof[t] = *reinterpret_cast<const s16x4*>(gbase + (size_t)(16 * t + p) * DIM + 4 * g);
            r.of[t] = *reinterpret_cast<const s16x4*>(abase + (size_t)(16 * t + p) * DIM + 4 * g);
            r.ls[t] = *reinterpret_cast<const f32x4*>(lse + ((size_t)win * HEADS + h) * NTOK + 16 * t + 4 * g);
        }
        return r;
    };
    WinRegs nx;
    if (slot < nwin) nx = load_window(slot);
    for (int win = slot; win < nwin; win += nslots) {
        const WinRegs cur = nx;
        const uint32_t pair = (uint32_t)win * HEADS + h;        // dropout element index = (pair*64 + query)*64 + key
        // stage K, Q, dO rows (lane = token) for the transposed (gather) fragments
        *reinterpret_cast<u32x4*>(ql + lane * HD) = cur.rows[0]; *reinterpret_cast<u32x4*>(ql + lane * HD + 8) = cur.rows[1];
        *reinterpret_cast<u32x4*>(kl + lane * HD) = cur.rows[2]; *reinterpret_cast<u32x4*>(kl + lane * HD + 8) = cur.rows[3];
        *reinterpret_cast<u32x4*>(dol + lane * HD) = cur.rows[4]; *reinterpret_cast<u32x4*>(dol + lane * HD + 8) = cur.rows[5];
        s16x4 qf[4], kf[4], vf[4], dof[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            qf[t] = cur.qf[t]; kf[t] = cur.kf[t]; vf[t] = cur.vf[t]; dof[t] = cur.dof[t];
            // rowsum(P * dP) of query 16t+p = dO . O (O = D(P) V, the saved attention output: holds with dropout, whose mask sits
            // inside both factors); the four lane groups hold four channels each
            const float d = rows_sum(dot_bf16x4(dof[t], cur.of[t]));
            if (g == 0) dl[wave][16 * t + p] = d;
        }
        if (win + nslots < nwin) nx = load_window(win + nslots);
        wave_lds_sync();
        // column-fragments [d p][tok 16t+4g+j] (A operands of the transposed products)
        s16x4 kT[4], qT[4], doT[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x4 a, b, c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tok = 16 * t + 4 * g + j;
                a[j] = kl[tok * HD + p]; b[j] = ql[tok * HD + p]; c[j] = dol[tok * HD + p];
            }
            kT[t] = __builtin_bit_cast(s16x4, a); qT[t] = __builtin_bit_cast(s16x4, b); doT[t] = __builtin_bit_cast(s16x4, c);
        }

        // ---------------- N-layout sweep: rows = query 16qt+4g+e, cols = key 16kt+p ----------------
        f32x4 dvT[4], dkT[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { dvT[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dkT[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {            // dV / dK contract over queries: two query tiles per MFMA
            f32x4 l2[2], dd[2];                     // log2(e) * log-sum-exp and dO . O of this lane's four query rows
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int qt = 2 * qp + hh;
                l2[hh] = cur.ls[qt] * 1.4426950408889634f;
                dd[hh] = *reinterpret_cast<const f32x4*>(&dl[wave][16 * qt + 4 * g]);
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s16x4 prb[2], dsb[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int qt = 2 * qp + hh;
                    const f32x4 bf = bn[qt][kt];
                    const f32x4 s = mfma16x16x16(qf[qt], kf[kt], f32x4{0.f, 0.f, 0.f, 0.f});
                    const f32x4 dp = mfma16x16x16(dof[qt], vf[kt], f32x4{0.f, 0.f, 0.f, 0.f});
                    f32x4 pr, ds;
                    float dm[4] = {1.f, 1.f, 1.f, 1.f};          // O = D(P) V with D = mask/keep: dP = mask/keep * (dO V^T)
                    if (drop_thresh)       // this lane: queries 4g .. 4g+3 of one key; the key's pair partner is the neighbouring lane
                        drop_pair4_rows(drop_seed, (pair * 64u + 16u * qt + 4u * g) * 64u + 16u * kt + p, 64u, drop_thresh, drop_inv_keep, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pr[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(__builtin_fmaf(s[e], 0.25f, bf[e]), 1.4426950408889634f, -l2[hh][e]));
                        ds[e] = pr[e] * (dp[e] * dm[e] - dd[hh][e]);
                        pr[e] *= dm[e];                    // dV uses the dropped probabilities
                        dbacc[qt][kt][e] += ds[e];
                    }
                    prb[hh] = to_bf16x4(pr);
                    dsb[hh] = to_bf16x4(ds);
                    // the same tile as [key p][query 4g .. 4g+3]: 32-byte rows, read back transposed below
                    *reinterpret_cast<s16x4*>(&dsl[wave][hh * 4 + kt][p * 16 + 4 * g]) = dsb[hh];
                }
                dvT[kt] = mfma16x16x32(join4(doT[2 * qp], doT[2 * qp + 1]), join4(prb[0], prb[1]), dvT[kt]);   // dV^T += dO^T P
                dkT[kt] = mfma16x16x32(join4(qT[2 * qp], qT[2 * qp + 1]), join4(dsb[0], dsb[1]), dkT[kt]);     // dK^T += Q^T dS
            }
            wave_lds_sync();
            // dQ^T[d][query] = K^T[d][key] dS^T[key][query], two key tiles per MFMA; the B fragment of lane (g, p) = dS[query p][keys
            // 4g .. 4g+3] = column p of rows 4g .. 4g+3 of the stored [key][query] tile: the transposing LDS read
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int qt = 2 * qp + hh;
                f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    s16x4 t[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        t[u] = __builtin_bit_cast(s16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4i16*)&dsl[wave][hh * 4 + 2 * kp + u][(4 * g + (p >> 2)) * 16 + (p & 3) * 4]));
                    dq = mfma16x16x32(join4(kT[2 * kp], kT[2 * kp + 1]), join4(t[0], t[1]), dq);
                }
                bf16_t* o = gqkv + ((size_t)win * NTOK + 16 * qt + p) * (3 * DIM) + h * HD + 4 * g;
                *reinterpret_cast<u32x2*>(o) = u32x2{pack_bf16x2(dq[0] * 0.25f, dq[1] * 0.25f), pack_bf16x2(dq[2] * 0.25f, dq[3] * 0.25f)};
            }
            wave_lds_sync();      // the dS tiles are overwritten by the next pair
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16_t* o = gqkv + ((size_t)win * NTOK + 16 * kt + p) * (3 * DIM) + h * HD + 4 * g;
            *reinterpret_cast<u32x2*>(o + DIM) = u32x2{pack_bf16x2(dkT[kt][0] * 0.25f, dkT[kt][1] * 0.25f),
                                                      pack_bf16x2(dkT[kt][2] * 0.25f, dkT[kt][3] * 0.25f)};
            *reinterpret_cast<u32x2*>(o + 2 * DIM) = u32x2{pack_bf16x2(dvT[kt][0], dvT[kt][1]), pack_bf16x2(dvT[kt][2], dvT[kt][3])};
        }
        wave_lds_sync();      // LDS tiles are overwritten by the next window
    }

    // the dS sum goes to this wave's (workgroup's) OWN slice of the scratch buffer with plain stores ([part][h][qt][kt][lane][4]); the
    // parts are summed by dbias_sum_kernel.  (Float atomics straight into the dense gradient -- every wave of a head adding to the
    // same 4,096 addresses, 128-way -- were 49 of the kernel's 245 us.)
    float* part = dbias_part + ((size_t)slot * HEADS + h) * 4096;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            *reinterpret_cast<f32x4*>(part + (((size_t)qt * 4 + kt) * 64 + lane) * 4) = dbacc[qt][kt];
}

// gd = g * mask/keep (element index m*192 + n): gradient through proj_drop / the MLP's Dropout
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ gin, bf16_t* __restrict__ gout, size_t n4,
                                                          uint32_t thresh, float inv_keep, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(gin + 4 * i);
        const uint32_t e0 = (uint32_t)(4 * i);
        *reinterpret_cast<u32x2*>(gout + 4 * i) = u32x2{
            pack_bf16x2(v[0] * drop_scale(seed, e0, thresh, inv_keep), v[1] * drop_scale(seed, e0 + 1, thresh, inv_keep)),
            pack_bf16x2(v[2] * drop_scale(seed, e0 + 2, thresh, inv_keep), v[3] * drop_scale(seed, e0 + 3, thresh, inv_keep))};
    }
}

}  // namespace

// gout bf16 [M][192] = gin fp32 [M][192] * dropout mask / (1 - p): backward of nn.Dropout after proj / mlp.2
// (model.py:82,132,150) with the mask the forward epilogue used (same seed).
extern "C" int tup_dropout_bwd(const float* gin, void* gout, long long n, float drop_p, unsigned int drop_seed, void* stream)
{
    if (n <= 0) return 0;
    if (n % 4 != 0 || drop_p <= 0.f || drop_p >= 1.f) return (int)hipErrorInvalidValue;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dropout_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        gin, (bf16_t*)gout, (size_t)(n / 4), (uint32_t)((double)drop_p * 4294967296.0), 1.0f / (1.0f - drop_p), drop_seed);
    TUP_CHECK_LAUNCH();
    return 0;
}

namespace {
// 16 rows per block and sweep.  Every workgroup ends with 2 x 192 float atomics onto the SAME dgamma / dbeta addresses, and those
// serialise in L2: at 61,440 rows 1,280 workgroups took 95 us per call, 256 take 57 us (same box, incl. the two memsets) -- the grid
// is one workgroup per CU, every workgroup the same number of sweeps.  TUP_LN_BWD_BLOCKS overrides (timing experiments).
int ln_bwd_blocks(int M)
{
    static const int cap = TUP_ENV_INT("TUP_LN_BWD_BLOCKS", 256);
    const int groups = (M + 15) / 16;
    const int sweeps = (groups + cap - 1) / cap;
    return (groups + sweeps - 1) / sweeps;
}
}  // namespace

// dx = LN'(gy) [+ gres]; dgamma/dbeta (fp32 [192]) are accumulated (caller zeroes them).  gdrop (optional, bf16 [M][192]) also
// receives dx * dropout mask / (1 - drop_p) for the site keyed by drop_seed (= tup_dropout_bwd of dx, which every LayerNorm backward
// of a block is followed by: the gradient enters the Dropout behind attn.proj / mlp.2 next, model.py:82,132,150).
extern "C" int tup_layernorm_bwd(const void* gy, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* gres, float* dx, float* dgamma, float* dbeta,
                                 int M, void* gdrop, float drop_p, unsigned int drop_seed, void* stream)
{
    if (M <= 0) return 0;
    if (gdrop && (drop_p <= 0.f || drop_p >= 1.f)) return (int)hipErrorInvalidValue;
    const int blocks = ln_bwd_blocks(M);
    layernorm_bwd_kernel<3><<<dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)gy, x, mean, rstd, gamma, gres, dx, dgamma, dbeta, M, (bf16_t*)gdrop,
        gdrop ? (uint32_t)((double)drop_p * 4294967296.0) : 0u, gdrop ? 1.0f / (1.0f - drop_p) : 1.f, drop_seed);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Same for the 128-wide rows of ResidualTransformer (model.py:17-18,28,31 of that plugin).
extern "C" int tup_layernorm128_bwd(const void* gy, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, const float* gres, float* dx, float* dgamma, float* dbeta,
                                    int M, void* gdrop, float drop_p, unsigned int drop_seed, void* stream)
{
    if (M <= 0) return 0;
    if (gdrop && (drop_p <= 0.f || drop_p >= 1.f)) return (int)hipErrorInvalidValue;
    const int blocks = ln_bwd_blocks(M);
    layernorm_bwd_kernel<2><<<dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)gy, x, mean, rstd, gamma, gres, dx, dgamma, dbeta, M, (bf16_t*)gdrop,
        gdrop ? (uint32_t)((double)drop_p * 4294967296.0) : 0u, gdrop ? 1.0f / (1.0f - drop_p) : 1.f, drop_seed);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Dense bias in the N-layout fragment order (backward only): fp32 [12][4 qt][4 kt][64][4].
extern "C" int tup_relpos_bias_expand_n(const float* table, float* frag, void* stream)
{
    relpos_expand_n_kernel<12><<<dim3(12 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(table, frag);
    TUP_CHECK_LAUNCH();
    return 0;
}

namespace {
// dbias_n[i] = sum over slots of part[slot][i], i < n4 float4s: 64 float4 columns x 4 slot groups per workgroup, LDS-combined
__global__ __launch_bounds__(256) void dbias_sum_kernel(const float* __restrict__ part, float* __restrict__ dbias_n, int n4, int nslots)
{
    __shared__ f32x4 red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < n4)
        for (int s = grp; s < nslots; s += 4) acc += reinterpret_cast<const f32x4*>(part)[(size_t)s * n4 + col];
    red[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && col < n4)
        reinterpret_cast<f32x4*>(dbias_n)[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Persistent waves per head: one wave per (slot, head), one wave per SIMD -- the grid is sized to ONE resident set of the chip
// (256 CUs x 4 SIMDs = 1,024 waves: 85 slots x 12 heads).  128 slots were 1.5 resident sets, i.e. two rounds with half of the chip
// idle in the second: 186 -> 154 us.  TUP_ATTN_BWD_SLOTS overrides (timing experiments).
inline int attn_bwd_slots(int nwin, int heads)
{
    static const int forced = TUP_ENV_INT("TUP_ATTN_BWD_SLOTS", 0);
    const int cap = forced > 0 ? forced : 1024 / heads;
    return nwin < cap ? nwin : cap;
}

template <int HEADS>
int launch_attn_bwd(const void* qkv, const void* gout, const void* att, const float* lse, const float* bias_n, void* gqkv,
                    float* dbias_n, float* scratch, int nwin, float drop_p, unsigned int drop_seed, hipStream_t s)
{
    if (scratch == nullptr || att == nullptr || lse == nullptr) return (int)hipErrorInvalidValue;
    uint32_t thresh; float inv_keep;
    drop_pair_params(drop_p, thresh, inv_keep);
    const int nslots = attn_bwd_slots(nwin, HEADS);
    const int nwaves = nslots * HEADS;            // multiple of 4 because HEADS is
    window_attn_bwd_kernel<HEADS><<<dim3(nwaves / 4), dim3(256), 0, s>>>(
        (const bf16_t*)qkv, (const bf16_t*)gout, (const bf16_t*)att, lse, bias_n, (bf16_t*)gqkv, scratch, nwin, nslots,
        thresh, inv_keep, drop_seed);
    TUP_CHECK_LAUNCH();
    const int n4 = HEADS * 1024;
    dbias_sum_kernel<<<dim3(n4 / 64), dim3(256), 0, s>>>(scratch, dbias_n, n4, nslots);
    TUP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// The same three entry points for `heads` = 8 (WindowTransformer, width 128) or 12: table / dtable fp32 [225][heads],
// frag / dbias_n fp32 [heads][4][4][64][4], qkv / gqkv bf16 [nwin][64][48*heads], gout bf16 [nwin][64][16*heads].
extern "C" int tup_relpos_bias_expand_n_h(const float* table, float* frag, int heads, void* stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (heads == 12) relpos_expand_n_kernel<12><<<dim3(12 * 16), dim3(256), 0, s>>>(table, frag);
    else if (heads == 8) relpos_expand_n_kernel<8><<<dim3(8 * 16), dim3(256), 0, s>>>(table, frag);
    else return (int)hipErrorInvalidValue;
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_window_attn_bwd_h(const void* qkv, const void* gout, const void* att, const float* lse, const float* bias_n,
                                     void* gqkv, float* dbias_n, float* scratch, int nwin, int heads, float drop_p,
                                     unsigned int drop_seed, void* stream)
{
    if (nwin <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f) return (int)hipErrorInvalidValue;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (heads == 12) return launch_attn_bwd<12>(qkv, gout, att, lse, bias_n, gqkv, dbias_n, scratch, nwin, drop_p, drop_seed, s);
    if (heads == 8) return launch_attn_bwd<8>(qkv, gout, att, lse, bias_n, gqkv, dbias_n, scratch, nwin, drop_p, drop_seed, s);
    return (int)hipErrorInvalidValue;
}

extern "C" int tup_relpos_bias_reduce_h(const float* dbias_n, float* dtable, int heads, void* stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (heads == 12) relpos_reduce_kernel<12><<<dim3((225 * 12 + 255) / 256), dim3(256), 0, s>>>(dbias_n, dtable);
    else if (heads == 8) relpos_reduce_kernel<8><<<dim3((225 * 8 + 255) / 256), dim3(256), 0, s>>>(dbias_n, dtable);
    else return (int)hipErrorInvalidValue;
    TUP_CHECK_LAUNCH();
    return 0;
}

// qkv bf16 [nwin][64][576], gout bf16 [nwin][64][192] (grad of the attention output before proj), att bf16 [nwin][64][192] (that
// output) and lse fp32 [nwin][12][64] (log-sum-exp of every score row), both as tup_window_attn_fwd left them -> gqkv bf16
// [nwin][64][576]; dbias_n fp32 [12][4][4][64][4] (dense bias gradient in the layout of tup_relpos_bias_expand_n) overwritten;
// scratch fp32 [tup_window_attn_bwd_scratch(nwin, 12)] (per-wave partial sums of the bias gradient, no initialisation needed).
extern "C" int tup_window_attn_bwd(const void* qkv, const void* gout, const void* att, const float* lse, const float* bias_n,
                                   void* gqkv, float* dbias_n, float* scratch, int nwin, float drop_p, unsigned int drop_seed,
                                   void* stream)
{
    if (nwin <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f) return (int)hipErrorInvalidValue;
    return launch_attn_bwd<12>(qkv, gout, att, lse, bias_n, gqkv, dbias_n, scratch, nwin, drop_p, drop_seed, reinterpret_cast<hipStream_t>(stream));
}

// Floats of scratch tup_window_attn_bwd(_h) needs for nwin windows and `heads` heads.
extern "C" long long tup_window_attn_bwd_scratch(int nwin, int heads)
{
    return nwin <= 0 ? 0 : (long long)attn_bwd_slots(nwin, heads) * heads * 4096;
}

// dense N-layout bias gradient (tup_window_attn_bwd) -> relative_position_bias_table gradient fp32 [225][12] (overwritten).
extern "C" int tup_relpos_bias_reduce(const float* dbias_n, float* dtable, void* stream)
{
    relpos_reduce_kernel<12><<<dim3((225 * 12 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(dbias_n, dtable);
    TUP_CHECK_LAUNCH();
    return 0;
}
