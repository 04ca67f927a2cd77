// Thin (3-channel side) convolutions and the output tail of the FastTransformer path (gfx950).
//
//   tup_conv3x3_c3_fwd      conv1 3->64 +bias +ReLU       models/FastTransformer/model.py:202-203,251
//   tup_conv3x3_planar_fwd  final_upscale 3->3*r*r + PixelShuffle(r)  utils.py:62-63,74-75,83-84 (n_feats=3)
//                           final_upscale_conv 3->3, fused "+ upscaled_input" and clamp
//                                                         model.py:212,317,320,327
//   tup_resize_aa_fwd       antialiased bilinear Resize + clamp   model.py:323-327, train.py:127-130
//
// These are HBM-bound: conv1 writes 128 B per pixel and reads 12 B; the planar kernels move a
// handful of fp32 planes.  conv1 still uses MFMA (im2col K = 27 padded to 32, one K-step) so
// the VALU never limits the store stream.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 32, HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int LW = 36;                                  // padded LDS row pitch (elements)
constexpr int PLANE = HALO_H * LW;                      // 360
constexpr int ZERO_BASE = 3 * PLANE;                    // a zero plane for the K padding (k >= 27)

// Store one pixel group's packed outputs with full-line coalescing.  The MFMA layout leaves lane (g, p) with the 32 B
// (chunks 2g, 2g+1) of pixel p, so "store chunk 2g, then chunk 2g+1" makes each store instruction write 16-B pieces at a
// 32-B stride.  Lanes p and p^8 swap one chunk (DPP row_ror:8) so that the first instruction covers pixels 0-7 of the
// group completely (8 x 128 B contiguous) and the second pixels 8-15.  row = out + first pixel of the group; npix =
// pixels of the group inside the map.
TUP_DEVICE void store_group_coalesced(bf16_t* row, int g, int p, const uint32_t (&pk)[8], int npix)
{
    const bool hi = p >= 8;
    uint32_t keep[4], send[4], recv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { keep[q] = hi ? pk[4 + q] : pk[q]; send[q] = hi ? pk[q] : pk[4 + q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) recv[q] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send[q], 0x128, 0xf, 0xf, false);   // row_ror:8
    const int px = p & 7, chunk = 2 * g + (hi ? 1 : 0);
    bf16_t* o1 = row + px * 64 + chunk * 8;
    if (px < npix) *reinterpret_cast<u32x4*>(o1) = hi ? u32x4{recv[0], recv[1], recv[2], recv[3]} : u32x4{keep[0], keep[1], keep[2], keep[3]};
    if (px + 8 < npix) *reinterpret_cast<u32x4*>(o1 + 8 * 64) = hi ? u32x4{keep[0], keep[1], keep[2], keep[3]} : u32x4{recv[0], recv[1], recv[2], recv[3]};
}

__global__ __launch_bounds__(256) void conv3x3_c3_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_mask, const bf16_t* __restrict__ out_mask,
    bf16_t* __restrict__ out, int H, int W, int relu, int tilesX, int tilesY)
{
    __shared__ __attribute__((aligned(16))) bf16_t lds[4 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int ty0 = ty * TH, tx0 = tx * TW;

    for (int idx = tid; idx < 4 * PLANE; idx += 256) {
        const int c = idx / PLANE, rem = idx - c * PLANE;
        const int yy = rem / LW, xx = rem - yy * LW;
        const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
        float v = 0.f;
        if (c < 3 && xx < HALO_W && iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const size_t gi = (((size_t)b * 3 + c) * H + iy) * W + ix;
            v = x[gi];
            if (in_mask && !(in_mask[gi] > 0.f)) v = 0.f;      // ReLU backward fused into the load
        }
        lds[idx] = f32_to_bf16(v);
    }

    // loop-invariant weight fragments: A[row = cout n_local][k = 8g + j]
    bf16x8 wf[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const bf16x8*>(wp + (ct * 16 + p) * 32 + 8 * g);

    f32x4 bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = bias ? *reinterpret_cast<const f32x4*>(bias + g * 16 + ct * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    // im2col offsets of this lane group's 8 k values: k = tap*3 + c
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int tap = k / 3, c = k - tap * 3;
        const int dy = tap / 3, dx = tap - dy * 3;
        koff[j] = (k < 27) ? (c * PLANE + dy * LW + dx) : ZERO_BASE;
    }
    __syncthreads();

#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int row = 2 * wave + (pg >> 1), x0 = (pg & 1) * 16;
        const int pb = row * LW + x0 + p;
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = lds[koff[j] + pb];
        const int oy = ty0 + row, ox = tx0 + x0 + p;
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma16x16x32(wf[ct], pf, f32x4{0.f, 0.f, 0.f, 0.f});
        if (oy >= H || ox >= W) continue;
        bf16_t* o = out + (((size_t)b * H + oy) * W + ox) * 64 + g * 16;
        uint32_t pk[8], mw[8];
        if (out_mask) {
            const bf16_t* mp = out_mask + (((size_t)b * H + oy) * W + ox) * 64 + g * 16;
            const u32x4 m0 = *reinterpret_cast<const u32x4*>(mp), m1 = *reinterpret_cast<const u32x4*>(mp + 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) { mw[q] = m0[q]; mw[4 + q] = m1[q]; }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[ct][e] + bv[ct][e];
                if (relu) v[e] = fmaxf(v[e], 0.f);
                if (out_mask) {
                    const int wi = (ct * 4 + e) >> 1;
                    const float mv = __builtin_bit_cast(float, (e & 1) ? (mw[wi] & 0xffff0000u) : (mw[wi] << 16));
                    if (!(mv > 0.f)) v[e] = 0.f;
                }
            }
            pk[ct * 2 + 0] = pack_bf16x2(v[0], v[1]);
            pk[ct * 2 + 1] = pack_bf16x2(v[2], v[3]);
        }
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
}

#ifndef TUP_CONV1_OCC
#define TUP_CONV1_OCC 4
#endif
// Persistent form of the kernel above: a workgroup walks tiles blockIdx.x, + gridDim.x, ... .  conv1 is a 943 MB store
// stream (128 B written per pixel against 12 B read) and the one-tile kernel pays a global round trip (halo loads ->
// LDS) in front of every 32 KB of stores: 2.7 TB/s written.  Here the halo values are requested into registers TWO
// tiles ahead (vA / vB alternate; LDS double-buffered, one barrier per tile), weights / bias / im2col offsets are set
// up once per workgroup and the (element -> source offset) map of the staging once per thread: 3.2 TB/s (353 -> 295 us
// at 8 x 720p).  What is left is the interaction of the scattered fp32 halo reads with the store stream: without the
// reads the same kernel stores at 4.8 TB/s (timing ablation), and a dedicated loader wave did not beat this form.
__global__ __launch_bounds__(256, TUP_CONV1_OCC) void conv3x3_c3_persistent_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_mask, const bf16_t* __restrict__ out_mask,
    bf16_t* __restrict__ out, int H, int W, int relu, int tilesX, int tilesY, int ntiles, int xcd_bands)
{
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][4 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;

    // staging map: element idx = tid + k*256 of the 3 haloed planes -> offset from the tile's origin pixel
    constexpr int NST = (3 * PLANE + 255) / 256;         // 5
    int goff[NST], dyx[NST];                               // dyx = (yy << 8) | xx, or -1 for elements never loaded
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int idx = tid + k * 256;
        const int c = idx / PLANE, rem = idx - c * PLANE;
        const int yy = rem / LW, xx = rem - yy * LW;
        const bool used = idx < 3 * PLANE && xx < HALO_W;
        goff[k] = (c * H + (yy - 1)) * W + (xx - 1);
        dyx[k] = used ? ((yy << 8) | xx) : -1;
    }
    for (int i = tid; i < 2 * 4 * PLANE; i += 256) (&lds[0][0])[i] = f32_to_bf16(0.f);      // zero plane + row padding, once

    bf16x8 wf[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const bf16x8*>(wp + (ct * 16 + p) * 32 + 8 * g);
    f32x4 bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = bias ? *reinterpret_cast<const f32x4*>(bias + g * 16 + ct * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int tap = k / 3, c = k - tap * 3;
        const int dy = tap / 3, dx = tap - dy * 3;
        koff[j] = (k < 27) ? (c * PLANE + dy * LW + dx) : ZERO_BASE;
    }

    auto load_tile = [&](int t, float (&v)[NST]) {
        const int tx = t % tilesX, r1 = t / tilesX;
        const int ty = r1 % tilesY, b = r1 / tilesY;
        const int ty0 = ty * TH, tx0 = tx * TW;
        const size_t origin = ((size_t)b * 3 * H + ty0) * W + tx0;
        const bool interior = ty0 >= 1 && ty0 + TH + 1 <= H && tx0 >= 1 && tx0 + TW + 1 <= W;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            v[k] = 0.f;
            bool ok = dyx[k] >= 0;
            if (ok && !interior) {
                const int iy = ty0 - 1 + (dyx[k] >> 8), ix = tx0 - 1 + (dyx[k] & 255);
                ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            }
            if (ok) {
                const size_t gi = origin + goff[k];      // goff may be negative: size_t wrap-around is the subtraction
                v[k] = x[gi];
                if (in_mask && !(in_mask[gi] > 0.f)) v[k] = 0.f;      // ReLU backward fused into the load
            }
        }
    };

    float vA[NST], vB[NST];
    // tiles of this workgroup: XCD-contiguous bands (blockIdx & 7 = XCD): a 34-float halo row straddles two or three 128-B lines
    // that the tiles left and right of it read too -- with neighbouring tiles on one XCD its L2 serves them (round-robin tiles
    // fetched 269 MB for the 88 MB input)
    int G = gridDim.x, t = blockIdx.x;
    if ((gridDim.x & 7) == 0 && xcd_bands) {
        const int band = (ntiles + 7) >> 3, start = (blockIdx.x & 7) * band;
        ntiles = min(ntiles, start + band);
        t = start + (blockIdx.x >> 3);
        G = gridDim.x >> 3;
    }
    if (t < ntiles) load_tile(t, vA);
    if (t + G < ntiles) load_tile(t + G, vB);
    __syncthreads();                                        // zero fill done before the first staging stores
    auto process = [&](int t, float (&v)[NST], int buf) {
        bf16_t* L = lds[buf];
#pragma unroll
        for (int k = 0; k < NST; ++k)
            if (dyx[k] >= 0) L[tid + k * 256] = f32_to_bf16(v[k]);
        __syncthreads();
        const int tx = t % tilesX, r1 = t / tilesX;
        const int ty = r1 % tilesY, b = r1 / tilesY;
        const int ty0 = ty * TH, tx0 = tx * TW;
        if (t + 2 * G < ntiles) load_tile(t + 2 * G, v);        // in flight under two tiles' MFMAs and stores
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            const int row = 2 * wave + (pg >> 1), x0 = (pg & 1) * 16;
            const int pb = row * LW + x0 + p;
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = L[koff[j] + pb];
            const int oy = ty0 + row, ox = tx0 + x0 + p;
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma16x16x32(wf[ct], pf, bv[ct]);
            if (oy >= H) continue;                  // wave-uniform (a pixel group is one image row)
            const bool ok = ox < W;
            uint32_t pk[8], mw[8];
            if (out_mask && ok) {
                const bf16_t* mp = out_mask + (((size_t)b * H + oy) * W + ox) * 64 + g * 16;
                const u32x4 m0 = *reinterpret_cast<const u32x4*>(mp), m1 = *reinterpret_cast<const u32x4*>(mp + 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) { mw[q] = m0[q]; mw[4 + q] = m1[q]; }
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vv[e] = acc[ct][e];                          // bias rode in the accumulator
                    if (relu) vv[e] = fmaxf(vv[e], 0.f);
                    if (out_mask && ok) {
                        const int wi = (ct * 4 + e) >> 1;
                        const float mv = __builtin_bit_cast(float, (e & 1) ? (mw[wi] & 0xffff0000u) : (mw[wi] << 16));
                        if (!(mv > 0.f)) vv[e] = 0.f;
                    }
                }
                pk[ct * 2 + 0] = pack_bf16x2(vv[0], vv[1]);
                pk[ct * 2 + 1] = pack_bf16x2(vv[2], vv[3]);
            }
            store_group_coalesced(out + (((size_t)b * H + oy) * W + tx0 + x0) * 64, g, p, pk, W - (tx0 + x0));
        }
    };
    while (t < ntiles) {
        process(t, vA, 0);
        t += G;
        if (t >= ntiles) break;
        process(t, vB, 1);
        t += G;
    }
}

// Direct fp32 3x3 conv on planar [B][3][H][W] input, COUT = 3*r*r outputs written through
// PixelShuffle(r) to [B][3][H*r][W*r]; optional "+ add" (same shape as out) and clamp to [0,1].
// weights: [COUT][28] fp32 (27 taps in (cin, ky, kx) order + 1 pad), bias [COUT].
__global__ __launch_bounds__(256) void conv3x3_planar_kernel(
    const float* __restrict__ x, const float* __restrict__ w28, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ out, int H, int W, int r, int clamp01)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [COUT][28] + [COUT]
    const int cout = 3 * r * r;
    for (int i = threadIdx.x; i < cout * 28; i += 256) wl[i] = w28[i];
    for (int i = threadIdx.x; i < cout; i += 256) wl[cout * 28 + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iy = oy + dy - 1, ix = ox + dx - 1;
                in[c * 9 + dy * 3 + dx] = (iy >= 0 && iy < H && ix >= 0 && ix < W)
                                              ? x[(((size_t)b * 3 + c) * H + iy) * W + ix] : 0.f;
            }
    const int Hr = H * r, Wr = W * r, rr = r * r;
    for (int co = 0; co < cout; ++co) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(wl + co * 28);
        float s = wl[cout * 28 + co];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const f32x4 wq = wv[q];
            s = fmaf(wq[0], in[q * 4 + 0], s);
            s = fmaf(wq[1], in[q * 4 + 1], s);
            s = fmaf(wq[2], in[q * 4 + 2], s);
            if (q < 6) s = fmaf(wq[3], in[q * 4 + 3], s);
        }
        const int c = co / rr, sp = co - c * rr;
        const int si = sp / r, sj = sp - si * r;
        const size_t oi = (((size_t)b * 3 + c) * Hr + (oy * r + si)) * Wr + (ox * r + sj);
        if (add) s += add[oi];
        if (clamp01 == 2) out[(size_t)gridDim.z * 3 * Hr * Wr + oi] = fminf(fmaxf(s, 0.f), 1.f);      // [unclamped | clamped]
        else if (clamp01) s = fminf(fmaxf(s, 0.f), 1.f);
        out[oi] = s;
    }
}

// The r = 1, 3 -> 3 case (final_upscale_conv at HR and its input gradient, model.py:317): FOUR adjacent pixels per thread.  A row of
// a thread's 3x6 input window is one aligned 16-byte load + two edge scalars, so the four pixels cost 27 load instructions instead
// of 108 and the 81 weights sit in registers.  W % 4 == 0.
__global__ __launch_bounds__(256) void conv3x3_planar_r1x4_kernel(
    const float* __restrict__ x, const float* __restrict__ w28, const float* __restrict__ bias,
    const float* __restrict__ add, float* __restrict__ out, int H, int W, int clamp01)
{
    __shared__ __attribute__((aligned(16))) float wl[3 * 28 + 4];
    if (threadIdx.x < 84) wl[threadIdx.x] = w28[threadIdx.x];
    if (threadIdx.x < 3) wl[84 + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int ox = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    float wr[3][27];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const f32x4 wq = *reinterpret_cast<const f32x4*>(wl + co * 28 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e < 27) wr[co][q * 4 + e] = wq[e];
        }
    f32x4 acc[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) acc[co] = f32x4{wl[84 + co], wl[84 + co], wl[84 + co], wl[84 + co]};
    // all 27 loads of the thread are requested before the first use
    f32x4 m[3][3];
    float e0[3][3], e5[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy + dy - 1;
            const bool in = iy >= 0 && iy < H;                 // wave-uniform (a wave is one output row)
            const float* row = x + (((size_t)b * 3 + c) * H + (in ? iy : oy)) * W;
            m[c][dy] = *reinterpret_cast<const f32x4*>(row + ox);
            e0[c][dy] = row[ox > 0 ? ox - 1 : ox];
            e5[c][dy] = row[ox + 4 < W ? ox + 4 : ox];
            if (!in) { m[c][dy] = f32x4{0.f, 0.f, 0.f, 0.f}; e0[c][dy] = 0.f; e5[c][dy] = 0.f; }
            if (ox == 0) e0[c][dy] = 0.f;
            if (ox + 4 >= W) e5[c][dy] = 0.f;
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float v[6] = {e0[c][dy], m[c][dy][0], m[c][dy][1], m[c][dy][2], m[c][dy][3], e5[c][dy]};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const float wk = wr[co][c * 9 + dy * 3 + dx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(wk, v[e + dx], acc[co][e]);
                }
        }
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        const size_t oi = (((size_t)b * 3 + co) * H + oy) * W + ox;
        f32x4 s = acc[co];
        if (add) s += *reinterpret_cast<const f32x4*>(add + oi);
        if (clamp01 == 2) {          // training: [unclamped | clamped] in one pass (the clamp's backward gate and the model output)
            f32x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = fminf(fmaxf(s[e], 0.f), 1.f);
            *reinterpret_cast<f32x4*>(out + (size_t)gridDim.z * 3 * H * W + oi) = c;
        } else if (clamp01) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = fminf(fmaxf(s[e], 0.f), 1.f);
        }
        *reinterpret_cast<f32x4*>(out + oi) = s;
    }
}

// Separable antialiased-bilinear resize evaluated as one 2-D gather per output pixel.
// Tap tables (ymin/ysize/yw[Ho][KY], xmin/xsize/xw[Wo][KX]) are built on the host exactly as
// aten's _compute_indices_weights_aa does (float32); see transformerupscaler_amd/resize_taps.py.
__global__ __launch_bounds__(256) void resize_aa_kernel(
    const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ ymin,
    const int* __restrict__ ysize, const float* __restrict__ yw, int KY, const int* __restrict__ xmin,
    const int* __restrict__ xsize, const float* __restrict__ xw, int KX, int Hi, int Wi, int Ho, int Wo,
    int clamp01, size_t both_off)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int plane = blockIdx.z;
    if (ox >= Wo || oy >= Ho) return;
    const float* src = in + (size_t)plane * Hi * Wi;
    const int y0 = ymin[oy], ny = ysize[oy], x0 = xmin[ox], nx = xsize[ox];
    float acc = 0.f;
    if (nx <= 8) {
        // the column weights live in registers for all rows and the taps of a row are requested together (the rolled
        // double loop re-read a weight and paid a load round trip per tap: 317 us for 3 x 4 planes of 1440x2560 -> 1080x1920)
        float wx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wx[j] = j < nx ? xw[ox * KX + j] : 0.f;
        for (int i = 0; i < ny; ++i) {
            const float* rowp = src + (size_t)(y0 + i) * Wi + x0;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = j < nx ? rowp[j] : 0.f;
            float h = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) h = fmaf(wx[j], v[j], h);
            acc = fmaf(yw[oy * KY + i], h, acc);
        }
    } else {
        for (int i = 0; i < ny; ++i) {
            const float* rowp = src + (size_t)(y0 + i) * Wi + x0;
            float h = 0.f;
            for (int j = 0; j < nx; ++j) h = fmaf(xw[ox * KX + j], rowp[j], h);
            acc = fmaf(yw[oy * KY + i], h, acc);
        }
    }
    if (clamp01 == 2) out[both_off + ((size_t)plane * Ho + oy) * Wo + ox] = fminf(fmaxf(acc, 0.f), 1.f);
    else if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
    out[((size_t)plane * Ho + oy) * Wo + ox] = acc;
}

// Separable form: a workgroup owns RS_TR output rows x 256 output columns, a thread one column.  The horizontal taps are applied
// once per (input row of the tile, column) into LDS (read back by the same thread only: per-thread indexed storage, no barrier),
// the vertical taps read it: ~(rows/ratio + KY) * nx + ny loads per RS_TR outputs instead of ny * 8 per output, and the four waves
// store 1 KB of an output row together.  Same fmaf chains per axis as the 2-D gather above, but the horizontal sum is formed before
// the vertical one for every input row (the gather does the same) -- identical results.  RS_MAXR bounds the input-row span.
constexpr int RS_TR = 16, RS_MAXR = 32;
__global__ __launch_bounds__(256) void resize_aa_sep_kernel(
    const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ ymin,
    const int* __restrict__ ysize, const float* __restrict__ yw, int KY, const int* __restrict__ xmin,
    const int* __restrict__ xsize, const float* __restrict__ xw, int KX, int Hi, int Wi, int Ho, int Wo,
    int clamp01, size_t both_off)
{
    __shared__ float hbuf[RS_MAXR][256];
    const int col = threadIdx.x;
    const int ox = blockIdx.x * 256 + col, oxc = min(ox, Wo - 1);
    const int oy0 = blockIdx.y * RS_TR, oy1 = min(oy0 + RS_TR, Ho) - 1;
    const int plane = blockIdx.z;
    const float* src = in + (size_t)plane * Hi * Wi;
    const int r0 = ymin[oy0], r1 = ymin[oy1] + ysize[oy1] - 1;          // monotone tables: input rows of the tile
    const int x0 = xmin[oxc], nx = xsize[oxc];
    float wx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wx[j] = j < nx ? xw[oxc * KX + j] : 0.f;
    for (int r = r0; r <= r1; r += 4) {          // four input rows per trip, all their loads in flight together
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* rowp = src + (size_t)min(r + u, r1) * Wi + x0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = j < nx ? rowp[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float h = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) h = fmaf(wx[j], v[u][j], h);
            if (r + u <= r1) hbuf[r + u - r0][col] = h;
        }
    }
    if (ox >= Wo) return;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const int y0 = ymin[oy] - r0, ny = ysize[oy];
        float acc = 0.f;
        for (int i = 0; i < ny; ++i) acc = fmaf(yw[oy * KY + i], hbuf[y0 + i][col], acc);
        if (clamp01 == 2) out[both_off + ((size_t)plane * Ho + oy) * Wo + ox] = fminf(fmaxf(acc, 0.f), 1.f);
        else if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
        out[((size_t)plane * Ho + oy) * Wo + ox] = acc;
    }
}

__global__ __launch_bounds__(256) void clamp01_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = fminf(fmaxf(in[i], 0.f), 1.f);
}

}  // namespace

// x: [B][3][H][W] fp32 (NCHW, as the reference module receives it); wp: [64][32] bf16 packed
// (row n_local = ct*16+4g+e holds channel g*16+ct*4+e; k = tap*3 + cin, zero padded 27..31);
// bias: [64] fp32 in channel order (or NULL); out: [B][H][W][64] bf16.
// The same kernel is the input-gradient conv of the 64->3 convs (up1_conv, decoder_conv2) with the
// transposed/flipped weight packed as wp; for that use in_mask (fp32, shape of x: input *= in_mask > 0,
// the ReLU backward of up1_conv) and out_mask (bf16 NHWC, shape of out: out *= out_mask > 0).
extern "C" int tup_conv3x3_c3_fwd(const float* x, const void* wp, const float* bias, const float* in_mask,
                                  const void* out_mask, void* out, int B, int H, int W, int relu, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long nblk = (long long)tilesX * tilesY * B;
    if (nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    static const bool one_tile = TUP_ENV_FLAG("TUP_CONV1_ONE_TILE");      // A/B switch: the non-persistent kernel
    if (!one_tile && (long long)B * 3 * H * W < (1LL << 31)) {
        // three workgroups are resident per CU (140 registers): grids of 3, 6 or 12 per CU run the same 282-285 us at 8 x 720p, 2 or 4
        // per CU (a partial round) 330 us
        static const int per_cu = TUP_ENV_INT("TUP_CONV1_WG_PER_CU", 2 * TUP_CONV1_OCC);
        const unsigned grid = (unsigned)(nblk < 256 * per_cu ? nblk : 256 * per_cu);
        static const int xcd_bands = TUP_ENV_FLAG("TUP_CONV1_NO_XCD_BANDS") ? 0 : 1;
        conv3x3_c3_persistent_kernel<<<dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            x, (const bf16_t*)wp, bias, in_mask, (const bf16_t*)out_mask, (bf16_t*)out, H, W, relu, tilesX, tilesY, (int)nblk, xcd_bands);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    conv3x3_c3_kernel<<<dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        x, (const bf16_t*)wp, bias, in_mask, (const bf16_t*)out_mask, (bf16_t*)out, H, W, relu, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_conv3x3_planar_fwd(const float* x, const float* w28, const float* bias, const float* add,
                                      float* out, int B, int H, int W, int r, int clamp01, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (r < 1 || r > 6 || B > 65535) return (int)hipErrorInvalidValue;
    const int cout = 3 * r * r;
    static const bool one_px = TUP_ENV_FLAG("TUP_PLANAR_ONE_PIXEL");           // A/B switch
    if (r == 1 && W % 4 == 0 && !one_px) {
        conv3x3_planar_r1x4_kernel<<<dim3((W / 4 + 63) / 64, (H + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            x, w28, bias, add, out, H, W, clamp01);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    conv3x3_planar_kernel<<<grid, dim3(256), (size_t)cout * 29 * sizeof(float),
                            reinterpret_cast<hipStream_t>(stream)>>>(x, w28, bias, add, out, H, W, r, clamp01);
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_resize_aa_fwd(const float* in, float* out, const int* ymin, const int* ysize, const float* yw,
                                 int KY, const int* xmin, const int* xsize, const float* xw, int KX, int planes,
                                 int Hi, int Wi, int Ho, int Wo, int clamp01, void* stream)
{
    if (planes <= 0) return 0;
    if (planes > 65535) return (int)hipErrorInvalidValue;
    // separable kernel when every column has <= 8 taps and RS_TR output rows never span more than RS_MAXR input rows
    // (RS_TR / ratio + KY + 1 is an upper bound of the span)
    static const bool gather = TUP_ENV_FLAG("TUP_RESIZE_GATHER");             // A/B switch
    if (!gather && KX <= 8 && (long long)RS_TR * Hi / Ho + KY + 2 <= RS_MAXR) {
        resize_aa_sep_kernel<<<dim3((Wo + 255) / 256, (Ho + RS_TR - 1) / RS_TR, planes), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            in, out, ymin, ysize, yw, KY, xmin, xsize, xw, KX, Hi, Wi, Ho, Wo, clamp01, (size_t)planes * Ho * Wo);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, planes);
    resize_aa_kernel<<<grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        in, out, ymin, ysize, yw, KY, xmin, xsize, xw, KX, Hi, Wi, Ho, Wo, clamp01, (size_t)planes * Ho * Wo);
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_clamp01_fwd(const float* in, float* out, long long n, void* stream)
{
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    clamp01_kernel<<<dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(in, out, (size_t)n);
    TUP_CHECK_LAUNCH();
    return 0;
}
