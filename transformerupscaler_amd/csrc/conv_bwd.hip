// Backward-only kernels of the convolutional side of the FastTransformer path (gfx950): weight /
// bias gradients of every 3x3 conv, the input gradient of the planar up-convs, the backward of the
// antialiased resize + clamp, and the gradient merge at `feat`.  They replace what torch autograd
// runs under reference train.py:138 for model.py:251-265 and :308-327.
// (Input gradients of the 64-channel convs reuse the forward implicit-GEMM kernels with
// transposed / flipped weights: conv3x3_c64.hip, conv_thin.hip.)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 32, HALO_W = TW + 2, HALO_H = TH + 2, NPIX_HALO = HALO_H * HALO_W;
constexpr int X_TILE_BYTES = NPIX_HALO * 128;

TUP_DEVICE s16x4 lds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// xr > 1: x is [H*xr][W*xr][64] and the tile reads its (xsi, xsj) space-to-depth plane (stride-xr conv input)
TUP_DEVICE void stage_x_halo(char* lds, const bf16_t* xb, int H, int W, int ty0, int tx0, int tid, int xr = 1, int xsi = 0, int xsj = 0) {
    for (int idx = tid; idx < NPIX_HALO * 8; idx += 256) {
        const int q = idx >> 3, c = idx & 7;
        const int yy = q / HALO_W, xx = q - yy * HALO_W;
        const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
            v = *reinterpret_cast<const u32x4*>(xb + ((size_t)(iy * xr + xsi) * (W * xr) + (ix * xr + xsj)) * 64 + c * 8);
        *reinterpret_cast<u32x4*>(lds + swz128(q, c)) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dW[co][tap][ci] += sum_pixels G[p][co] * X[p + tap][ci]   (64 x 64 channels, MFMA 16x16x16,
// both operands transposed out of LDS with ds_read_b64_tr_b16).  Persistent over pixel tiles: the
// whole 64 x 576 partial result lives in registers (144 per lane) and is flushed once.
// G may be the sub-pixel plane `sp` of a pixel-shuffled gradient [B][H*gr][W*gr][64].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv3x3_wgrad_c64_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gmap, float* __restrict__ dwp, float* __restrict__ dbias,
    int B, int H, int W, int gr, int sp, int tilesX, int tilesY, int xr, int xsp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* x_lds = smem;
    char* g_lds = smem + X_TILE_BYTES;          // [256 pixels][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int trq = l16 >> 2, trp = l16 & 3;
    const int si = sp / gr, sj = sp - si * gr;
    const int Hg = H * gr, Wg = W * gr;
    const int cit = wave & 3, coh = wave >> 2;

    // Work split (8 waves): wave w owns the input-channel tile cit = w & 3, the cout pair coh = w >> 2 and all nine
    // taps, so one K-step is 2 G fragments + 9 X fragments for 18 MFMAs.  (With "wave = cout tile" every wave re-read
    // all 36 X fragments: 37 transposed reads per 36 MFMAs.)
    f32x4 acc[9][2];                 // [tap][cout tile of the pair]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    // Register double-buffering of the tile operands: the X halo (340 pixels x 8 chunks -> 6 pieces per thread) and
    // the G tile (256 x 8 -> 4 pieces) of tile k+1 are requested before the K loop of tile k and written to LDS after
    // it, so the global-load round trip hides under the MFMAs (it used to sit between the two barriers of every tile).
    constexpr int XP = (NPIX_HALO * 8 + 511) / 512;
    u32x4 xpre[XP], gpre[4];
    const int xsi = xsp / xr, xsj = xsp % xr;
    // A piece's offset from its tile's first halo pixel does not depend on the tile (xrel / grel, set once): an interior tile costs
    // one 64-bit add per load; the general form (clipping at the map's edges) is ~28 vector instructions per load and was a third
    // of the kernel's vector work.
    const bool last_ok = tid + (XP - 1) * 512 < NPIX_HALO * 8;
    auto fetch_tile = [&](int tile) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY;
        const int b = t / tilesY;
        const int ty0 = ty * TH, tx0 = tx * TW;
        const bf16_t* xb = x + (size_t)b * H * W * 64 * xr * xr;
        const bf16_t* gb0 = gmap + (size_t)b * Hg * Wg * 64;
        if (ty0 >= 1 && ty0 + TH + 1 <= H && tx0 >= 1 && tx0 + TW + 1 <= W) {          // interior tile
            const bf16_t* xt = xb + ((size_t)((ty0 - 1) * xr + xsi) * (W * xr) + ((tx0 - 1) * xr + xsj)) * 64;
            const bf16_t* gt = gb0 + ((size_t)(ty0 * gr + si) * Wg + (tx0 * gr + sj)) * 64;
            // (rebuilt from the thread index per tile -- six instructions per load; as loop invariants the ten offsets cost the
            // K loop its registers: 256 + scratch)
            const int tq = (int)opaque_copy((uint32_t)tid);
#pragma unroll
            for (int u = 0; u < XP; ++u) {
                const int idx = min(tq + u * 512, NPIX_HALO * 8 - 1);
                const int q = idx >> 3, c = idx & 7;
                const int yy = (q * 241) >> 13, xx = q - yy * HALO_W;            // q / 34 for q < 340
                xpre[u] = u32x4{0u, 0u, 0u, 0u};
                if (u + 1 < XP || last_ok) xpre[u] = *reinterpret_cast<const u32x4*>(xt + ((yy * xr) * (W * xr) + xx * xr) * 64 + c * 8);
            }
            const bf16_t* gl = gt + (((tq >> 8) * gr) * Wg + ((tq >> 3) & 31) * gr) * 64 + (tq & 7) * 8;
#pragma unroll
            for (int u = 0; u < 4; ++u) gpre[u] = *reinterpret_cast<const u32x4*>(gl + (size_t)u * (2 * gr * Wg * 64));
            return;
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 512;
            const int q = idx >> 3, c = idx & 7;
            const int yy = q / HALO_W, xx = q - yy * HALO_W;
            const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
            xpre[u] = u32x4{0u, 0u, 0u, 0u};
            if (idx < NPIX_HALO * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                xpre[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)(iy * xr + xsi) * (W * xr) + (ix * xr + xsj)) * 64 + c * 8);
        }
        const bf16_t* gb = gmap + (size_t)b * Hg * Wg * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * 512;
            const int pix = idx >> 3, c = idx & 7;
            const int oy = ty0 + (pix >> 5), ox = tx0 + (pix & 31);
            gpre[u] = u32x4{0u, 0u, 0u, 0u};
            if (oy < H && ox < W)
                gpre[u] = *reinterpret_cast<const u32x4*>(gb + ((size_t)(oy * gr + si) * Wg + (ox * gr + sj)) * 64 + c * 8);
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 512;
            if (idx < NPIX_HALO * 8) *reinterpret_cast<u32x4*>(x_lds + swz128(idx >> 3, idx & 7)) = xpre[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * 512;
            *reinterpret_cast<u32x4*>(g_lds + swz128(idx >> 3, idx & 7)) = gpre[u];
        }
    };

    // lane constants of the fragment reads (see the K loop): G rows are 32 pixels, so their swizzle phase does not depend on the row
    int gbase[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gcol = 16 * (2 * coh + c) + 4 * trp;
            gbase[c][h] = swz128(8 * g + trq + 4 * h, gcol >> 3) + (gcol & 7) * 2;
        }
    // X: pixel q = hy * 34 + 8g + trq + dx + 4h, logical chunk xc, byte (xcol & 7) * 2 inside it
    const int xcol = 16 * cit + 4 * trp, xc = xcol >> 3;
    int xs[3][2], xb0[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q0 = 8 * g + trq + dx + 4 * h;
            xs[dx][h] = q0 >> 1;
            xb0[dx][h] = q0 * 128 + (xcol & 7) * 2;
        }
    const int ntiles = tilesX * tilesY * B;
    if ((int)blockIdx.x < ntiles) fetch_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        commit_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch_tile(tile + gridDim.x);
        // One K-step = one tile row of 32 pixels on v_mfma_f32_16x16x32_bf16 (the 16x16x16 form runs at half its
        // rate on gfx950): lane group g contracts pixels 8g .. 8g+7, delivered by two transposed reads (4 pixels each).
        // The loop runs over the ten HALO rows: the three dx fragments of halo row hy serve the taps (dy, dx) of output rows
        // hy - dy, so every X fragment is read once per tile (30 instead of 72) against a window of three rows' G fragments.
        // Addresses: the swizzle phase of pixel q is (q >> 1) & 7 and a halo row is 34 pixels = 17 pairs, so row hy shifts the
        // phase of row 0 by hy; what is left per read is add / and / xor / shift-add on a lane constant, the row's byte offset
        // is an immediate.  (The row loop used to rebuild every address from q: 926 vector instructions per tile and wave for
        // 144 MFMAs -- the kernel ran on the vector port, profiles/r04_pmc_mfma_train.json: 5.4 per MFMA, matrix pipe 30 % busy.)
        auto join = [](s16x4 lo, s16x4 hi) {
            const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
            return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
        };
        bf16x8 gf[3][2];                 // G fragments of output rows hy, hy - 1, hy - 2 (slot = row % 3)
        // the phases pass through an empty asm per tile: left visible, hipcc hoists all 60 read addresses out of the tile loop
        // (256 registers + scratch)
        uint32_t xst[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h) xst[dx][h] = opaque_copy((uint32_t)xs[dx][h]);
#pragma unroll
        for (int hy = 0; hy < HALO_H; ++hy) {
            if (hy < TH) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    gf[hy % 3][c] = join(lds_read_tr16(g_lds + gbase[c][0] + hy * 4096), lds_read_tr16(g_lds + gbase[c][1] + hy * 4096));
                if (cit == 0) {    // bias gradient: this lane holds G[8 pixels][co = 16*(2*coh + c) + l16]
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[c] += bf16_to_f32(gf[hy % 3][c][j]);
                }
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                s16x4 part[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ph = (int)((xst[dx][h] + hy) & 7u);
                    part[h] = lds_read_tr16(x_lds + xb0[dx][h] + ((xc ^ ph) << 4) + hy * (HALO_W * 128));
                }
                const bf16x8 bfr = join(part[0], part[1]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int ry = hy - dy;
                    if (ry >= 0 && ry < TH) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[dy * 3 + dx][c] = mfma16x16x32(gf[ry % 3][c], bfr, acc[dy * 3 + dx][c]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // flush: D[row = co 4g+e][col = ci l16]; packed layout [co][tap][ci] keeps 16 lanes on one 64-B segment
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#ifndef TUP_EXP_NOATOMIC      // timing experiment (wrong results): the kernel without its 9.4 M float atomics
                atomicAdd(dwp + ((size_t)(16 * (2 * coh + c) + 4 * g + e) * 9 + tap) * 64 + 16 * cit + l16, acc[tap][c][e]);
#else
                if (acc[tap][c][e] == 123.456f) dwp[0] = 1.f;
#endif
    if (dbias && cit == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v = bsum[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) atomicAdd(dbias + 16 * (2 * coh + c) + l16, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// thin convs (cout = 3: up1_conv, decoder_conv2): G planar fp32 [B][3][H][W], X NHWC bf16.
// dwp[co][tap][ci] += ..., dbias[co] += sum G.  The 64-cout kernel's scheme with the 3 real couts padded to one 16-row tile:
// wave = input-channel tile, all nine taps, one K-step per tile row on v_mfma_f32_16x16x32_bf16, the loop over the ten halo
// rows (each X fragment read once, a window of three rows' G fragments), read addresses from lane constants.  (Until round 4
// the four waves split the taps on the 16x16x16 form and rebuilt every swizzled address from the pixel index: 740 vector
// instructions per tile and wave for 144 half-rate MFMAs, 2.0 TB/s.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_thin_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gpl, float* __restrict__ dwp, float* __restrict__ dbias,
    int B, int H, int W, int tilesX, int tilesY)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* x_lds = smem;
    bf16_t* g_lds = reinterpret_cast<bf16_t*>(smem + X_TILE_BYTES);     // [256 pixels][16 couts] (3 real)
    const int tid = threadIdx.x, lane = tid & 63, cit = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int trq = l16 >> 2, trp = l16 & 3;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < 256 * 16; i += 256) g_lds[i] = f32_to_bf16(0.f);       // columns 3..15 stay zero
    // register double-buffering of the tile operands (X halo: 11 pieces per thread, G: this thread's pixel): the loads
    // of tile k+1 fly during the K loop of tile k -- this kernel is a 64-channel streaming read with little MFMA work
    constexpr int XP = (NPIX_HALO * 8 + 255) / 256;
    u32x4 xpre[XP];
    float gpre[3];
    const bool last_ok = tid + (XP - 1) * 256 < NPIX_HALO * 8;
    auto fetch_tile = [&](int tile) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY;
        const int b = t / tilesY;
        const int ty0 = ty * TH, tx0 = tx * TW;
        const bf16_t* xb = x + (size_t)b * H * W * 64;
        if (ty0 >= 1 && ty0 + TH + 1 <= H && tx0 >= 1 && tx0 + TW + 1 <= W) {          // interior tile: offsets from the thread index
            const bf16_t* xt = xb + ((size_t)(ty0 - 1) * W + (tx0 - 1)) * 64;
            const int tq = (int)opaque_copy((uint32_t)tid);                             // (not hoisted: see the 64-cout kernel)
#pragma unroll
            for (int u = 0; u < XP; ++u) {
                const int idx = min(tq + u * 256, NPIX_HALO * 8 - 1);
                const int q = idx >> 3, c = idx & 7;
                const int yy = (q * 241) >> 13, xx = q - yy * HALO_W;                   // q / 34 for q < 340
                xpre[u] = u32x4{0u, 0u, 0u, 0u};
                if (u + 1 < XP || last_ok) xpre[u] = *reinterpret_cast<const u32x4*>(xt + (yy * W + xx) * 64 + c * 8);
            }
            const float* gp0 = gpl + ((size_t)b * 3 * H + ty0 + (tq >> 5)) * W + tx0 + (tq & 31);
#pragma unroll
            for (int co = 0; co < 3; ++co) gpre[co] = gp0[(size_t)co * H * W];
            return;
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 256;
            const int q = idx >> 3, c = idx & 7;
            const int yy = q / HALO_W, xx = q - yy * HALO_W;
            const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
            xpre[u] = u32x4{0u, 0u, 0u, 0u};
            if (idx < NPIX_HALO * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                xpre[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * W + ix) * 64 + c * 8);
        }
        const int oy = ty0 + (tid >> 5), ox = tx0 + (tid & 31);
        const bool ok = oy < H && ox < W;
#pragma unroll
        for (int co = 0; co < 3; ++co) gpre[co] = ok ? gpl[(((size_t)b * 3 + co) * H + oy) * W + ox] : 0.f;
    };
    // lane constants of the fragment reads.  G rows are 32 bytes (16 couts), not swizzled: pixel ry * 32 + 8g + trq (+ 4)
    const int gb0 = (8 * g + trq) * 32 + trp * 8;
    // X: pixel q = hy * 34 + 8g + trq + dx + 4h; the swizzle phase of row hy is row 0's + hy (34 pixels = 17 pairs)
    const int xcol = 16 * cit + 4 * trp, xc = xcol >> 3;
    int xs[3][2], xb0[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q0 = 8 * g + trq + dx + 4 * h;
            xs[dx][h] = q0 >> 1;
            xb0[dx][h] = q0 * 128 + (xcol & 7) * 2;
        }
    auto join = [](s16x4 lo, s16x4 hi) {
        const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
        return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
    };
    const int ntiles = tilesX * tilesY * B;
    if ((int)blockIdx.x < ntiles) fetch_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 256;
            if (idx < NPIX_HALO * 8) *reinterpret_cast<u32x4*>(x_lds + swz128(idx >> 3, idx & 7)) = xpre[u];
        }
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            bsum[co] += gpre[co];
            g_lds[tid * 16 + co] = f32_to_bf16(gpre[co]);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch_tile(tile + gridDim.x);
        uint32_t xst[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h) xst[dx][h] = opaque_copy((uint32_t)xs[dx][h]);
        bf16x8 gf[3];                    // G fragments of output rows hy, hy - 1, hy - 2 (slot = row % 3)
#pragma unroll
        for (int hy = 0; hy < HALO_H; ++hy) {
            if (hy < TH)
                gf[hy % 3] = join(lds_read_tr16(reinterpret_cast<const char*>(g_lds) + gb0 + hy * 1024),
                                  lds_read_tr16(reinterpret_cast<const char*>(g_lds) + gb0 + 128 + hy * 1024));
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                s16x4 part[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ph = (int)((xst[dx][h] + hy) & 7u);
                    part[h] = lds_read_tr16(x_lds + xb0[dx][h] + ((xc ^ ph) << 4) + hy * (HALO_W * 128));
                }
                const bf16x8 bfr = join(part[0], part[1]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int ry = hy - dy;
                    if (ry >= 0 && ry < TH) acc[dy * 3 + dx] = mfma16x16x32(gf[ry % 3], bfr, acc[dy * 3 + dx]);
                }
            }
        }
        __syncthreads();
    }
    // D[row = co 4g+e][col = ci l16]: only rows 0..2 (g == 0, e < 3) are real
    if (g == 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int e = 0; e < 3; ++e) atomicAdd(dwp + ((size_t)e * 9 + tap) * 64 + 16 * cit + l16, acc[tap][e]);
    }
    if (dbias) {
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            float v = bsum[co];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) atomicAdd(dbias + co, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv1 (3 -> 64): X planar fp32 [B][3][H][W], G NHWC bf16 [B][H][W][64].
// dw[co][ci][ky][kx] += (PyTorch layout), dbias[co] +=.  thread = (co, tap group).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_wgrad_c3_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ gmap, float* __restrict__ dw, float* __restrict__ dbias,
    int B, int H, int W, int tilesX, int tilesY)
{
    __shared__ float x_lds[3][HALO_H][HALO_W + 2];
    __shared__ __attribute__((aligned(16))) bf16_t g_lds[256 * 64];
    const int tid = threadIdx.x, co = tid & 63, tg = tid >> 6;
    const int tap0 = (tg == 0) ? 0 : 1 + 2 * tg, ntap = (tg == 0) ? 3 : 2;
    float acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[a][c] = 0.f;
    float bsum = 0.f;
    const int ntiles = tilesX * tilesY * B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY;
        const int b = t / tilesY;
        const int ty0 = ty * TH, tx0 = tx * TW;
        for (int idx = tid; idx < 3 * NPIX_HALO; idx += 256) {
            const int c = idx / NPIX_HALO, q = idx - c * NPIX_HALO;
            const int yy = q / HALO_W, xx = q - yy * HALO_W;
            const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
            x_lds[c][yy][xx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)b * 3 + c) * H + iy) * W + ix] : 0.f;
        }
        for (int idx = tid; idx < 256 * 8; idx += 256) {
            const int pix = idx >> 3, c = idx & 7;
            const int oy = ty0 + (pix >> 5), ox = tx0 + (pix & 31);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (oy < H && ox < W) v = *reinterpret_cast<const u32x4*>(gmap + (((size_t)b * H + oy) * W + ox) * 64 + c * 8);
            *reinterpret_cast<u32x4*>(g_lds + pix * 64 + c * 8) = v;
        }
        __syncthreads();
        for (int pix = 0; pix < 256; ++pix) {
            const float gv = bf16_to_f32(g_lds[pix * 64 + co]);
            const int py = pix >> 5, px = pix & 31;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (a < ntap) {
                    const int tap = tap0 + a;
                    const int yy = py + tap / 3, xx = px + tap % 3;
                    acc[a][0] = fmaf(gv, x_lds[0][yy][xx], acc[a][0]);
                    acc[a][1] = fmaf(gv, x_lds[1][yy][xx], acc[a][1]);
                    acc[a][2] = fmaf(gv, x_lds[2][yy][xx], acc[a][2]);
                }
            }
            if (tg == 0) bsum += gv;
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (a < ntap)
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(dw + ((size_t)co * 3 + c) * 9 + tap0 + a, acc[a][c]);
    if (dbias && tg == 0) atomicAdd(dbias + co, bsum);
}

// ------------------------------------------------------------------------------------------------
// planar convs (final_upscale 3 -> 3*r*r + PixelShuffle, final_upscale_conv 3 -> 3):
//   wgrad: dw[co][ci][ky][kx] += sum G_pre[co][p] X[ci][p+tap], dbias[co] += sum G_pre[co][p],
//          G_pre[c*r*r + sp][y][x] = G[c][y*r+si][x*r+sj]   (G planar fp32 [B][3][H*r][W*r]).
// One thread per output element (co, ci, tap) / bias, looping over the pixels of an LDS tile.
// ------------------------------------------------------------------------------------------------
// v2: a thread owns one sub-pixel phase (si, sj) and walks LR pixels, holding the phase's 3 x 27 partial weight
// gradients (+ 3 bias sums) in registers for its whole persistent run: 30 LDS reads per 81 FMAs.  (v1 gave each THREAD
// one (cout, tap) output and looped it over the tile's pixels -- 84 of 256 threads busy at r = 1, two LDS reads per FMA;
// 0.7 ms per call.)  The per-thread sums are combined through LDS float atomics once at the end.
// RT / THT > 0: compile-time r and rows per tile -> the tile's operands are fetched into registers one tile AHEAD (all loads of a
// tile in flight under the previous tile's arithmetic); RT = 0: generic r, staged in place (a load -> LDS-store loop with
// run-time bounds is not unrolled and pays a global round trip per iteration: 20 us per 1,024-pixel tile, the whole 0.36 ms).
template <int RT, int THT>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_planar_kernel(
    const float* __restrict__ x, const float* __restrict__ gpl, float* __restrict__ dw, float* __restrict__ dbias,
    int B, int H, int W, int r_, int th_, int tilesX, int tilesY)
{
    const int r = RT > 0 ? RT : r_, th = RT > 0 ? THT : th_;
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int cout = 3 * r * r, rr = r * r;
    float* x_lds = fl;                               // [3][th+2][HALO_W]
    float* g_lds = fl + 3 * (th + 2) * HALO_W;       // [cout][th*32]
    float* red = g_lds + cout * th * 32;             // [rr][84]
    const int npix = th * 32;
    const int tid = threadIdx.x;
    const int nslot = 256 / rr;
    const int ph = tid % rr, slot = tid / rr;
    const bool worker = slot < nslot;
    float acc[3][27], bsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[c][k] = 0.f;
    for (int i = tid; i < rr * 84; i += 256) red[i] = 0.f;
    const int Hr = H * r, Wr = W * r;
    const int ntiles = tilesX * tilesY * B;
    constexpr int XRm = RT > 0 ? (THT + 2 + 7) / 8 : 1, GRm = RT > 0 ? THT / 8 : 1, COm = RT > 0 ? 3 * RT * RT : 1;
    float xv[3][XRm][2], gv[COm][GRm];
    auto fetch_regs = [&](int tile_) {
        if constexpr (RT > 0) {
            int t_ = tile_;
            const int tx_ = t_ % tilesX; t_ /= tilesX;
            const int ty_ = t_ % tilesY, b_ = t_ / tilesY;
            const int y0_ = ty_ * THT, x0_ = tx_ * TW;
            const int lx_ = tid & 31, ly_ = tid >> 5;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* xc = x + ((size_t)b_ * 3 + c) * H * W;
#pragma unroll
                for (int k = 0; k < XRm; ++k) {
                    const int iy = y0_ - 1 + ly_ + 8 * k, ix = x0_ - 1 + lx_;
                    const bool rowok = iy >= 0 && iy < H && ly_ + 8 * k < THT + 2;
                    xv[c][k][0] = (rowok && ix >= 0 && ix < W) ? xc[(size_t)iy * W + ix] : 0.f;
                    xv[c][k][1] = (rowok && lx_ < HALO_W - 32 && ix + 32 < W) ? xc[(size_t)iy * W + ix + 32] : 0.f;
                }
            }
#pragma unroll
            for (int co = 0; co < COm; ++co) {
                const int c = co / (RT * RT), sp = co % (RT * RT), si = sp / RT, sj = sp % RT;
                const float* gc = gpl + ((size_t)b_ * 3 + c) * Hr * Wr;
#pragma unroll
                for (int k = 0; k < GRm; ++k) {
                    const int oy = y0_ + ly_ + 8 * k, ox = x0_ + lx_;
                    gv[co][k] = (oy < H && ox < W) ? gc[(size_t)(oy * RT + si) * Wr + (ox * RT + sj)] : 0.f;
                }
            }
        }
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY;
        const int b = t / tilesY;
        const int ty0 = ty * th, tx0 = tx * TW;
        const int lx = tid & 31, ly = tid >> 5;                       // 8 rows of 32 columns per pass
        if constexpr (RT > 0) {
            constexpr int XR = (THT + 2 + 7) / 8, GR = THT / 8, CO = 3 * RT * RT;
            if (tile == (int)blockIdx.x) fetch_regs(tile);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < XR; ++k) {
                    const int yy = ly + 8 * k;
                    if (yy < THT + 2) {
                        x_lds[(c * (THT + 2) + yy) * HALO_W + lx] = xv[c][k][0];
                        if (lx < HALO_W - 32) x_lds[(c * (THT + 2) + yy) * HALO_W + lx + 32] = xv[c][k][1];
                    }
                }
#pragma unroll
            for (int co = 0; co < CO; ++co)
#pragma unroll
                for (int k = 0; k < GR; ++k) g_lds[co * npix + (ly + 8 * k) * 32 + lx] = gv[co][k];
        } else {
            for (int c = 0; c < 3; ++c) {
                const float* xc = x + ((size_t)b * 3 + c) * H * W;
                float* xl = x_lds + c * (th + 2) * HALO_W;
                for (int yy = ly; yy < th + 2; yy += 8) {
                    const int iy = ty0 - 1 + yy;
                    const bool rowok = iy >= 0 && iy < H;
                    const int ix = tx0 - 1 + lx;
                    xl[yy * HALO_W + lx] = (rowok && ix >= 0 && ix < W) ? xc[(size_t)iy * W + ix] : 0.f;
                    if (lx < HALO_W - 32) {
                        const int ix2 = ix + 32;
                        xl[yy * HALO_W + lx + 32] = (rowok && ix2 < W) ? xc[(size_t)iy * W + ix2] : 0.f;
                    }
                }
            }
            for (int c = 0; c < 3; ++c)
                for (int si = 0; si < r; ++si)
                    for (int sj = 0; sj < r; ++sj) {
                        const int co = c * rr + si * r + sj;
                        const float* gc = gpl + ((size_t)b * 3 + c) * Hr * Wr;
                        for (int py = ly; py < th; py += 8) {
                            const int oy = ty0 + py, ox = tx0 + lx;
                            g_lds[co * npix + py * 32 + lx] = (oy < H && ox < W) ? gc[(size_t)(oy * r + si) * Wr + (ox * r + sj)] : 0.f;
                        }
                    }
        }
        __syncthreads();
        if constexpr (RT > 0) { if (tile + (int)gridDim.x < ntiles) fetch_regs(tile + gridDim.x); }
        if (worker)
            for (int pix = slot; pix < npix; pix += nslot) {
                const float g0 = g_lds[(0 * rr + ph) * npix + pix], g1 = g_lds[(1 * rr + ph) * npix + pix],
                            g2 = g_lds[(2 * rr + ph) * npix + pix];
                bsum[0] += g0; bsum[1] += g1; bsum[2] += g2;
                const float* xp = x_lds + (pix >> 5) * HALO_W + (pix & 31);
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const float v = xp[(ci * (th + 2) + tap / 3) * HALO_W + tap % 3];
                        const int k = ci * 9 + tap;
                        acc[0][k] = fmaf(g0, v, acc[0][k]); acc[1][k] = fmaf(g1, v, acc[1][k]); acc[2][k] = fmaf(g2, v, acc[2][k]);
                    }
            }
        __syncthreads();
    }
    if (rr == 1 || rr == 4) {
        // lanes of one phase are rr apart: butterfly over the wave first, then ONE LDS atomic per wave and value (256 threads
        // adding to the same 84 LDS words serialised 64-fold per wave: this tail was most of the kernel's 0.39 ms)
        const int lane = tid & 63;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 28; ++k) {
                float v = (k < 27) ? acc[c][k < 27 ? k : 0] : bsum[c];
                if constexpr (RT > 0) {            // compile-time butterfly: the 84 chains overlap instead of serialising
#pragma unroll
                    for (int o = 32; o >= RT * RT; o >>= 1) v += __shfl_xor(v, o);
                } else {
                    for (int o = 32; o >= rr; o >>= 1) v += __shfl_xor(v, o);
                }
                if (lane < rr) atomicAdd(red + ph * 84 + c * 28 + k, v);
            }
        }
    } else if (worker) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 27; ++k) atomicAdd(red + ph * 84 + c * 28 + k, acc[c][k]);
            atomicAdd(red + ph * 84 + c * 28 + 27, bsum[c]);
        }
    }
    __syncthreads();
    for (int o = tid; o < rr * 84; o += 256) {
        const int p2 = o / 84, rem = o - p2 * 84;
        const int c = rem / 28, k = rem - c * 28;
        const int co = c * rr + p2;
        if (k < 27) atomicAdd(dw + co * 27 + k, red[o]);
        else if (dbias) atomicAdd(dbias + co, red[o]);
    }
}

// dgrad of a planar up-conv: gx[ci][y][x] = sum_{co,ky,kx} W[co][ci][ky][kx] * G_pre[co][y+1-ky][x+1-kx]
__global__ __launch_bounds__(256) void conv3x3_dgrad_planar_kernel(
    const float* __restrict__ gpl, const float* __restrict__ w, float* __restrict__ gx, int H, int W, int r)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [cout][27] original layout (co, ci, ky, kx)
    const int cout = 3 * r * r, rr = r * r;
    for (int i = threadIdx.x; i < cout * 27; i += 256) wl[i] = w[i];
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    const int Hr = H * r, Wr = W * r;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int co = 0; co < cout; ++co) {
        const int c = co / rr, s = co - c * rr;
        const int si = s / r, sj = s - si * r;
        const float* gp = gpl + ((size_t)b * 3 + c) * Hr * Wr;
        const float* wc = wl + co * 27;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = oy + 1 - ky;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = ox + 1 - kx;
                if (xx < 0 || xx >= W) continue;
                const float gv = gp[(size_t)(yy * r + si) * Wr + (xx * r + sj)];
                a0 = fmaf(wc[ky * 3 + kx], gv, a0);
                a1 = fmaf(wc[9 + ky * 3 + kx], gv, a1);
                a2 = fmaf(wc[18 + ky * 3 + kx], gv, a2);
            }
        }
    }
    const size_t o = ((size_t)b * 3 * H + oy) * W + ox;
    gx[o] = a0; gx[o + (size_t)H * W] = a1; gx[o + 2 * (size_t)H * W] = a2;
}

// The r = 2 case of the above (the last final_upscale stage at every scale but 3; model.py:316): FOUR adjacent LR pixels per thread.
// The 6 HR rows x 12 HR columns the four pixels gather from are four aligned 16-byte loads per row (72 loads per thread instead
// of 432 two-element-strided ones); every HR value feeds up to three of the four outputs, all with wave-uniform weights (scalar
// loads from the weight tensor: the indices are compile-time constants).  W % 4 == 0.
// (bounded to two waves per SIMD: unbounded it took 305 registers = ONE wave per SIMD for a streaming kernel, 218 us; with the bound 255
// registers and 8 spilled ones, 142 us)
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_planar_r2x4_kernel(
    const float* __restrict__ gpl, const float* __restrict__ w, float* __restrict__ gx, int H, int W)
{
    const int ox = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    const int Hr = 2 * H, Wr = 2 * W;
    f32x4 acc[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) acc[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool lft = ox > 0, rgt = ox + 4 < W;                   // the first / last 16-byte piece of a row is inside the image
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* gp = gpl + ((size_t)b * 3 + c) * Hr * Wr;
        f32x4 v[6][4];
#pragma unroll
        for (int ry = 0; ry < 6; ++ry) {
            const int Y = 2 * (oy - 1) + ry;
            const bool in = Y >= 0 && Y < Hr;                    // wave-uniform (a wave is one LR row)
            const float* row = gp + (size_t)(in ? Y : 0) * Wr + 2 * ox;
            v[ry][0] = *reinterpret_cast<const f32x4*>(row + (lft ? -4 : 0));
            v[ry][1] = *reinterpret_cast<const f32x4*>(row);
            v[ry][2] = *reinterpret_cast<const f32x4*>(row + 4);
            v[ry][3] = *reinterpret_cast<const f32x4*>(row + (rgt ? 8 : 4));
            if (!in) { v[ry][1] = f32x4{0.f, 0.f, 0.f, 0.f}; v[ry][2] = v[ry][1]; }
            if (!in || !lft) v[ry][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!in || !rgt) v[ry][3] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ry = 0; ry < 6; ++ry) {
            const int si = ry & 1, ky = 2 - (ry >> 1);            // HR row 2 (oy - 1 + ry / 2) + si  <->  tap row ky
#pragma unroll
            for (int i = 2; i < 14; ++i) {                        // HR column 2 ox - 4 + i = 2 (ox + xr) + sj
                const int sj = i & 1, xr = (i >> 1) - 2;
                const float g = v[ry][i >> 2][i & 3];
                const float* wc = w + (c * 4 + si * 2 + sj) * 27 + ky * 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kx = e + 1 - xr;                    // output ox + e reads LR column ox + e + 1 - kx
                    if (kx < 0 || kx > 2) continue;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) acc[ci][e] = fmaf(wc[ci * 9 + kx], g, acc[ci][e]);
                }
            }
        }
    }
    const size_t o = ((size_t)b * 3 * H + oy) * W + ox;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) *reinterpret_cast<f32x4*>(gx + o + (size_t)ci * H * W) = acc[ci];
}

// Fused L1 loss (autograd.l1_loss(..., fuse_into_model_backward=True)): the gradient that enters the clamp is not read from memory but
// formed here, sign(clamp(pre) - target) * scale with scale = d loss / numel on the device (nn.L1Loss backward, train.py:132,138).
TUP_DEVICE float l1_grad(float pre, float target, float s) {
    const float d = fminf(fmaxf(pre, 0.f), 1.f) - target;
    return d > 0.f ? s : (d < 0.f ? -s : 0.f);
}
// l1_scale[1] != 0 ("plain"): `pre` is the loss input itself (the Resize output of train.py:127-130, nothing clamps it): no clamp, no gate
TUP_DEVICE float l1_grad_plain(float o, float target, float s) {
    const float d = o - target;
    return d > 0.f ? s : (d < 0.f ? -s : 0.f);
}

// ------------------------------------------------------------------------------------------------
// backward of (antialiased resize -> clamp): gin[y][x] = sum over the output pixels whose taps cover
// (y, x) of wy*wx*gout, gout masked by 0 <= pre <= 1 when `pre` (the pre-clamp output) is given.
// oy0/oyn, ox0/oxn: for every input row / column the contiguous range of outputs that reference it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_aa_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ pre, float* __restrict__ gin,
    const int* __restrict__ ymin, const float* __restrict__ yw, int KY, const int* __restrict__ xmin,
    const float* __restrict__ xw, int KX, const int* __restrict__ oy0, const int* __restrict__ oyn,
    const int* __restrict__ ox0, const int* __restrict__ oxn, int Hi, int Wi, int Ho, int Wo, const float* __restrict__ l1_scale)
{
    const float l1s = l1_scale ? l1_scale[0] : 0.f;          // fused L1: `gout` is the loss target
    const bool plain = l1_scale && l1_scale[1] != 0.f;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int plane = blockIdx.z;
    if (x >= Wi || y >= Hi) return;
    const float* go = gout + (size_t)plane * Ho * Wo;
    const float* pr = pre ? pre + (size_t)plane * Ho * Wo : nullptr;
    float acc = 0.f;
    const int ya = oy0[y], yn = oyn[y], xa = ox0[x], xn = oxn[x];
    if (xn <= 4) {
        // column weights once per thread, a row's taps requested together (as in resize_aa_kernel)
        float wx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int ox = min(xa + j, Wo - 1); wx[j] = j < xn ? xw[ox * KX + (x - xmin[ox])] : 0.f; }
        for (int i = 0; i < yn; ++i) {
            const int oy = ya + i;
            const float wyv = yw[oy * KY + (y - ymin[oy])];
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = (size_t)oy * Wo + min(xa + j, Wo - 1);
                gv[j] = j < xn ? go[o] : 0.f;
                if (pr) {
                    const float pv = pr[o];
                    if (l1_scale) gv[j] = plain ? l1_grad_plain(pv, gv[j], l1s) : l1_grad(pv, gv[j], l1s);
                    if ((!plain && !(pv >= 0.f && pv <= 1.f)) || j >= xn) gv[j] = 0.f;
                }
            }
            float h = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) h = fmaf(wx[j], gv[j], h);
            acc = fmaf(wyv, h, acc);
        }
    } else {
        for (int i = 0; i < yn; ++i) {
            const int oy = ya + i;
            const float wyv = yw[oy * KY + (y - ymin[oy])];
            float h = 0.f;
            for (int j = 0; j < xn; ++j) {
                const int ox = xa + j;
                float gv = go[(size_t)oy * Wo + ox];
                if (pr) {
                    const float pv = pr[(size_t)oy * Wo + ox];
                    if (l1_scale) gv = plain ? l1_grad_plain(pv, gv, l1s) : l1_grad(pv, gv, l1s);
                    if (!plain && !(pv >= 0.f && pv <= 1.f)) gv = 0.f;
                }
                h = fmaf(xw[ox * KX + (x - xmin[ox])], gv, h);
            }
            acc = fmaf(wyv, h, acc);
        }
    }
    gin[((size_t)plane * Hi + y) * Wi + x] = acc;
}

// Separable form of the above: a workgroup owns RB_TR input rows x 256 input columns, a thread one column.  Horizontal pass: for
// every output row the tile's input rows reference, t[oy] = sum over the column's outputs of wx * (gated) gout -- once per
// (output row, column), into LDS (read back by the same thread only); vertical pass: gin[y] = sum_oy wy * t[oy].
constexpr int RB_TR = 16, RB_MAXR = 32;
__global__ __launch_bounds__(256) void resize_aa_bwd_sep_kernel(
    const float* __restrict__ gout, const float* __restrict__ pre, float* __restrict__ gin,
    const int* __restrict__ ymin, const float* __restrict__ yw, int KY, const int* __restrict__ xmin,
    const float* __restrict__ xw, int KX, const int* __restrict__ oy0, const int* __restrict__ oyn,
    const int* __restrict__ ox0, const int* __restrict__ oxn, int Hi, int Wi, int Ho, int Wo, const float* __restrict__ l1_scale)
{
    __shared__ float tbuf[RB_MAXR][256];
    const float l1s = l1_scale ? l1_scale[0] : 0.f;          // fused L1: `gout` is the loss target
    const bool plain = l1_scale && l1_scale[1] != 0.f;
    const int col = threadIdx.x;
    const int x = blockIdx.x * 256 + col, xc = min(x, Wi - 1);
    const int ya = blockIdx.y * RB_TR, yb = min(ya + RB_TR, Hi) - 1;
    const int plane = blockIdx.z;
    const float* go = gout + (size_t)plane * Ho * Wo;
    const float* pr = pre ? pre + (size_t)plane * Ho * Wo : nullptr;
    // output rows referenced by input rows ya..yb: the per-row ranges are monotone in y
    int o0 = Ho, o1 = -1;
    for (int y = ya; y <= yb; ++y) {
        const int a = oy0[y], n = oyn[y];
        if (n > 0) { o0 = min(o0, a); o1 = max(o1, a + n - 1); }
    }
    const int xa = ox0[xc], xn = oxn[xc];
    constexpr int XN = 6;                        // outputs referencing one input column (host checks the bound)
    float wx[XN];
#pragma unroll
    for (int j = 0; j < XN; ++j) { const int ox = min(xa + j, Wo - 1); wx[j] = j < xn ? xw[ox * KX + (xc - xmin[ox])] : 0.f; }
    for (int oy = o0; oy <= o1; oy += 4) {       // four output rows per trip, all their loads in flight together
        float gv[4][XN], pv[4][XN];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < XN; ++j) {
                const size_t o = (size_t)min(oy + u, o1) * Wo + min(xa + j, Wo - 1);
                gv[u][j] = j < xn ? go[o] : 0.f;
                pv[u][j] = (pr && j < xn) ? pr[o] : 0.5f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float h = 0.f;
#pragma unroll
            for (int j = 0; j < XN; ++j) {
                const float gg = !l1_scale ? gv[u][j] : (plain ? l1_grad_plain(pv[u][j], gv[u][j], l1s) : l1_grad(pv[u][j], gv[u][j], l1s));
                h = fmaf(wx[j], (plain || (pv[u][j] >= 0.f && pv[u][j] <= 1.f)) ? gg : 0.f, h);          // (wx[j] = 0 beyond the column's outputs)
            }
            if (oy + u <= o1) tbuf[oy + u - o0][col] = h;
        }
    }
    if (x >= Wi) return;
    for (int y = ya; y <= yb; ++y) {
        const int a = oy0[y], n = oyn[y];
        float acc = 0.f;
        for (int i = 0; i < n; ++i) {
            const int oy = a + i;
            acc = fmaf(yw[oy * KY + (y - ymin[oy])], tbuf[oy - o0][col], acc);
        }
        gin[((size_t)plane * Hi + y) * Wi + x] = acc;
    }
}

// gin = gout * (0 <= pre <= 1) [* (relu_src > 0)]
__global__ __launch_bounds__(256) void mask_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ pre,
                                                       const float* __restrict__ relu_src, float* __restrict__ gin, size_t n,
                                                       const float* __restrict__ l1_scale)
{
    const float l1s = l1_scale ? l1_scale[0] : 0.f;          // fused L1: `gout` is the loss target, `pre` is required
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        float g = gout[i];
        if (pre) { const float p = pre[i]; if (l1_scale) g = l1_grad(p, g, l1s); if (!(p >= 0.f && p <= 1.f)) g = 0.f; }
        if (relu_src && !(relu_src[i] > 0.f)) g = 0.f;
        gin[i] = g;
    }
}

// out = (a + b + fold(gpe)) * (feat > 0): merges the three gradient paths into `feat` (skip add,
// up-branch, patch_embed) and applies conv2's ReLU backward.  gpe is the reflect-PADDED map
// [B][Hp][Wp][64]; rows/cols >= H/W fold back onto 2H-2-y / 2W-2-x (F.pad reflect backward).
__global__ __launch_bounds__(256) void feat_grad_combine_kernel(
    const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const bf16_t* __restrict__ gpe,
    const bf16_t* __restrict__ feat, bf16_t* __restrict__ out, int B, int H, int W, int Hp, int Wp)
{
    const size_t total = (size_t)B * H * W * 8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = idx & 7;
        size_t pix = idx >> 3;
        const int x = pix % W; pix /= W;
        const int y = pix % H;
        const int bb = pix / H;
        const size_t off = (((size_t)bb * H + y) * W + x) * 64 + c * 8;
        float v[8];
        auto addv = [&](const bf16_t* p, bool first) {
            const u32x4 w4 = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __builtin_bit_cast(float, w4[q] << 16), hi = __builtin_bit_cast(float, w4[q] & 0xffff0000u);
                if (first) { v[2 * q] = lo; v[2 * q + 1] = hi; } else { v[2 * q] += lo; v[2 * q + 1] += hi; }
            }
        };
        addv(a + off, true);
        if (b) addv(b + off, false);
        const int y2 = 2 * H - 2 - y, x2 = 2 * W - 2 - x;
        const bool fy = (y2 >= H && y2 < Hp), fx = (x2 >= W && x2 < Wp);
        auto gpe_at = [&](int yy, int xx) { return gpe + (((size_t)bb * Hp + yy) * Wp + xx) * 64 + c * 8; };
        addv(gpe_at(y, x), false);
        if (fy) addv(gpe_at(y2, x), false);
        if (fx) addv(gpe_at(y, x2), false);
        if (fy && fx) addv(gpe_at(y2, x2), false);
        const u32x4 f4 = *reinterpret_cast<const u32x4*>(feat + off);
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float flo = __builtin_bit_cast(float, f4[q] << 16), fhi = __builtin_bit_cast(float, f4[q] & 0xffff0000u);
            pk[q] = pack_bf16x2(flo > 0.f ? v[2 * q] : 0.f, fhi > 0.f ? v[2 * q + 1] : 0.f);
        }
        *reinterpret_cast<u32x4*>(out + off) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    }
}

int persistent_grid(long long ntiles, int per_cu) {
    long long g = 256LL * per_cu;
    return (int)(ntiles < g ? ntiles : g);
}

}  // namespace

static int conv_c64_wgrad_launch(const void* x, const void* gmap, float* dwp, float* dbias,
                                 int B, int H, int W, int gr, int sp, int xr, int xsp, void* stream)
{
    if (B <= 0) return 0;
    if (gr < 1 || sp < 0 || sp >= gr * gr || xr < 1 || xsp < 0 || xsp >= xr * xr) return (int)hipErrorInvalidValue;
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long nt = (long long)tilesX * tilesY * B;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = X_TILE_BYTES + 256 * 128;
    TUP_SET_DYN_LDS((conv3x3_wgrad_c64_kernel), lds);
    static const int wg_per_cu = TUP_ENV_INT("TUP_WGRAD64_WG_PER_CU", 1);            // tuning knob (diagnostic build)
    conv3x3_wgrad_c64_kernel<<<dim3(persistent_grid(nt, wg_per_cu)), dim3(512), lds, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)x, (const bf16_t*)gmap, dwp, dbias, B, H, W, gr, sp, tilesX, tilesY, xr, xsp);
    TUP_CHECK_LAUNCH();
    return 0;
}

// dwp fp32 [64][9][64] (co, tap, ci) +=, dbias fp32 [64] += (or NULL).  x NHWC bf16 [B][H][W][64];
// gmap NHWC bf16 [B][H*gr][W*gr][64], sub-pixel plane sp (gr = 1, sp = 0: plain).
extern "C" int tup_conv3x3_c64_wgrad(const void* x, const void* gmap, float* dwp, float* dbias,
                                     int B, int H, int W, int gr, int sp, void* stream)
{
    return conv_c64_wgrad_launch(x, gmap, dwp, dbias, B, H, W, gr, sp, 1, 0, stream);
}

// Weight gradient of a stride-xr conv seen as a 3x3 conv over the space-to-depth input (packing.pack_conv_c64_stride2):
// x NHWC bf16 [B][H*xr][W*xr][64], plane xsp = si*xr + sj; gmap NHWC bf16 [B][H][W][64]; dwp as above (the block taps
// of that plane), dbias or NULL.
extern "C" int tup_conv3x3_c64_wgrad_s2d(const void* x, const void* gmap, float* dwp, float* dbias,
                                         int B, int H, int W, int xr, int xsp, void* stream)
{
    return conv_c64_wgrad_launch(x, gmap, dwp, dbias, B, H, W, 1, 0, xr, xsp, stream);
}

// thin conv (cout 3): gpl fp32 [B][3][H][W]; dwp fp32 [3][9][64] +=, dbias fp32 [3] += (or NULL).
extern "C" int tup_conv3x3_thin_wgrad(const void* x, const float* gpl, float* dwp, float* dbias,
                                      int B, int H, int W, void* stream)
{
    if (B <= 0) return 0;
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long nt = (long long)tilesX * tilesY * B;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    static const int thin_per_cu = TUP_ENV_INT("TUP_WGRADTHIN_WG_PER_CU", 2);        // tuning knob (diagnostic build)
    conv3x3_wgrad_thin_kernel<<<dim3(persistent_grid(nt, thin_per_cu)), dim3(256), X_TILE_BYTES + 256 * 32,
                                reinterpret_cast<hipStream_t>(stream)>>>((const bf16_t*)x, gpl, dwp, dbias, B, H, W, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    return 0;
}

// conv1: x fp32 [B][3][H][W], gmap NHWC bf16 [B][H][W][64]; dw fp32 [64][3][3][3] +=, dbias [64] +=.
extern "C" int tup_conv3x3_c3_wgrad(const float* x, const void* gmap, float* dw, float* dbias,
                                    int B, int H, int W, void* stream)
{
    if (B <= 0) return 0;
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long nt = (long long)tilesX * tilesY * B;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    conv3x3_wgrad_c3_kernel<<<dim3(persistent_grid(nt, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        x, (const bf16_t*)gmap, dw, dbias, B, H, W, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    return 0;
}

// planar convs: x fp32 [B][3][H][W]; gpl fp32 [B][3][H*r][W*r]; dw fp32 [3*r*r][3][3][3] +=, dbias [3*r*r] +=.
extern "C" int tup_conv3x3_planar_wgrad(const float* x, const float* gpl, float* dw, float* dbias,
                                        int B, int H, int W, int r, void* stream)
{
    if (B <= 0) return 0;
    if (r < 1 || r > 6) return (int)hipErrorInvalidValue;
    const int cout = 3 * r * r;
    if (cout * 28 > 12 * 256) return (int)hipErrorInvalidValue;
    // rows per tile: as many as keep four workgroups' LDS tiles on a CU (a tile's staging loops and two barriers are fixed costs:
    // 8-row tiles at r = 1 left each thread ONE pixel of arithmetic per tile)
    const int th = (r >= 4) ? 2 : (r == 3 ? 4 : (r == 2 ? 16 : 32));
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + th - 1) / th;
    const long long nt = (long long)tilesX * tilesY * B;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = ((size_t)3 * (th + 2) * HALO_W + (size_t)cout * th * 32 + (size_t)r * r * 84) * sizeof(float);
    // persistent workgroups, one resident set.  The r = 1 and r = 2 instances hold 239 / 256 registers = TWO workgroups per CU (the r = 2
    // one compiled to 258 without the bound above: ONE workgroup per CU, and its grid of three or four per CU ran as three or four
    // rounds); the generic instance (118 registers) four
    const dim3 grid(persistent_grid(nt, r <= 2 ? 2 : 4));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (r == 1) conv3x3_wgrad_planar_kernel<1, 32><<<grid, dim3(256), lds, st>>>(x, gpl, dw, dbias, B, H, W, r, th, tilesX, tilesY);
    else if (r == 2) conv3x3_wgrad_planar_kernel<2, 16><<<grid, dim3(256), lds, st>>>(x, gpl, dw, dbias, B, H, W, r, th, tilesX, tilesY);
    else conv3x3_wgrad_planar_kernel<0, 0><<<grid, dim3(256), lds, st>>>(x, gpl, dw, dbias, B, H, W, r, th, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    return 0;
}

// gx fp32 [B][3][H][W] = dgrad of Conv2d(3, 3*r*r, 3)+PixelShuffle(r); gpl fp32 [B][3][H*r][W*r];
// w fp32 [3*r*r][3][3][3] (reference layout, unpacked).
extern "C" int tup_conv3x3_planar_dgrad(const float* gpl, const float* w, float* gx, int B, int H, int W, int r, void* stream)
{
    if (B <= 0) return 0;
    if (r < 1 || r > 6 || B > 65535) return (int)hipErrorInvalidValue;
    static const bool one_px = TUP_ENV_FLAG("TUP_PLANAR_ONE_PIXEL");           // A/B switch
    if (r == 2 && W % 4 == 0 && !one_px) {
        conv3x3_dgrad_planar_r2x4_kernel<<<dim3((W / 4 + 63) / 64, (H + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(gpl, w, gx, H, W);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    conv3x3_dgrad_planar_kernel<<<grid, dim3(256), (size_t)3 * r * r * 27 * sizeof(float), reinterpret_cast<hipStream_t>(stream)>>>(
        gpl, w, gx, H, W, r);
    TUP_CHECK_LAUNCH();
    return 0;
}

// gin fp32 [planes][Hi][Wi] (overwritten) = backward of resize (+ clamp mask from `pre`, may be NULL).
extern "C" int tup_resize_aa_bwd(const float* gout, const float* pre, float* gin, const int* ymin, const float* yw, int KY,
                                 const int* xmin, const float* xw, int KX, const int* oy0, const int* oyn,
                                 const int* ox0, const int* oxn, int planes, int Hi, int Wi, int Ho, int Wo,
                                 const float* l1_scale, void* stream)
{
    if (planes <= 0) return 0;
    if (planes > 65535 || (l1_scale && !pre)) return (int)hipErrorInvalidValue;
    // separable kernel when a column is referenced by <= 6 outputs (an output reads KX adjacent columns, so an input column feeds at
    // most KX * Wo / Wi + 2 outputs) and RB_TR input rows never reference more than RB_MAXR output rows
    static const bool gather = TUP_ENV_FLAG("TUP_RESIZE_GATHER");             // A/B switch
    if (!gather && (long long)KX * Wo / Wi + 2 <= 6 && (long long)RB_TR * Ho / Hi + KY + 2 <= RB_MAXR) {
        resize_aa_bwd_sep_kernel<<<dim3((Wi + 255) / 256, (Hi + RB_TR - 1) / RB_TR, planes), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            gout, pre, gin, ymin, yw, KY, xmin, xw, KX, oy0, oyn, ox0, oxn, Hi, Wi, Ho, Wo, l1_scale);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((Wi + 63) / 64, (Hi + 3) / 4, planes);
    resize_aa_bwd_kernel<<<grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        gout, pre, gin, ymin, yw, KY, xmin, xw, KX, oy0, oyn, ox0, oxn, Hi, Wi, Ho, Wo, l1_scale);
    TUP_CHECK_LAUNCH();
    return 0;
}

// gin = gout * [0 <= pre <= 1] * [relu_src > 0]   (either mask source may be NULL)
extern "C" int tup_mask_bwd(const float* gout, const float* pre, const float* relu_src, float* gin, long long n,
                            const float* l1_scale, void* stream)
{
    if (n <= 0) return 0;
    if (l1_scale && !pre) return (int)hipErrorInvalidValue;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    mask_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(gout, pre, relu_src, gin, (size_t)n, l1_scale);
    TUP_CHECK_LAUNCH();
    return 0;
}

// out = (a + b + fold_reflect(gpe)) * (feat > 0); all NHWC bf16, gpe is [B][ceil8(H)][ceil8(W)][64]; b may be NULL.
extern "C" int tup_feat_grad_combine(const void* a, const void* b, const void* gpe, const void* feat, void* out,
                                     int B, int H, int W, void* stream)
{
    if (B <= 0) return 0;
    const int Hp = (H + 7) / 8 * 8, Wp = (W + 7) / 8 * 8;
    long long blocks = ((long long)B * H * W * 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    feat_grad_combine_kernel<<<dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)gpe, (const bf16_t*)feat, (bf16_t*)out, B, H, W, Hp, Wp);
    TUP_CHECK_LAUNCH();
    return 0;
}
