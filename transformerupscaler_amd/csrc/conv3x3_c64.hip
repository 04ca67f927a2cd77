// 3x3 same-pad convolution with 64 input channels as an implicit GEMM on MFMA (gfx950).
//
// Stands in for the aten conv2d call sites of the reference:
//   conv2            models/FastTransformer/model.py:204,252   (64->64, +bias, ReLU)
//   up1 convs        models/FastTransformer/utils.py:62,74,83  (64->64*r*r, +bias) + PixelShuffle(r) utils.py:63,75,84
//   up1_conv         models/FastTransformer/utils.py:32-40     (64->3, no bias, ReLU)
//   decoder_conv1/2  models/FastTransformer/model.py:228-229,312-313
//
// Layout: activations NHWC bf16 (128 B per pixel); weights pre-packed on the host to
// [ntile][tap][n_local][cin] bf16 so that one (ntile, tap) slab is contiguous.  One
// workgroup = 4 waves = an 8x32-pixel output tile; the (8+2)x(32+2) halo tile is staged
// once into LDS (XOR-swizzled 128-B rows) and reused by all 9 taps and all cout tiles
// (for the up-convs the PixelShuffle sub-pixel index IS the cout tile, so every tile
// iteration writes whole 128-B NHWC pixels of the shuffled image -- the shuffle is free).
// MFMA operands: A = weights (rows = cout), B = pixels (cols), v_mfma_f32_16x16x32_bf16,
// so a lane ends up with 16 consecutive output channels of one pixel (32-B stores).
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, HALO_W = TW + 2, HALO_H = TH + 2, NPIX_HALO = HALO_H * HALO_W;
constexpr int IN_TILE_BYTES = NPIX_HALO * 128;

enum { OUT_NHWC_BF16 = 0, OUT_PLANAR_F32 = 1 };

template <int CT, int OUT_MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const bf16_t* __restrict__ add, const bf16_t* __restrict__ mask,
    void* __restrict__ out, int H, int W, int ntiles, int r, int cout_valid, int relu,
    int tilesX, int tilesY, int in_r)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + IN_TILE_BYTES;
    constexpr int WROWS = CT * 16;            // weight rows per (ntile, tap) slab
    constexpr int WSLAB = WROWS * 128;        // bytes
    constexpr int WCH = WROWS * 8;            // 16-byte chunks per slab
    constexpr int WREGS = (WCH + 255) / 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int ty0 = ty * TH, tx0 = tx * TW;

    // ---- stage the haloed input tile (zero outside the image).  With in_r > 1 the logical input has
    // 64*in_r^2 channels stored pixel-shuffled as [B][H*in_r][W*in_r][64]; chunk ci = sub-pixel (si, sj)
    // (the gradient of an Upsampler conv output, read back through PixelShuffle^-1) ----
    const int nci = in_r * in_r;
    const int Hin = H * in_r, Win = W * in_r;
    const bf16_t* xb = x + (size_t)b * Hin * Win * 64;
    auto stage_input = [&](int ci) {
        const int si = ci / in_r, sj = ci - si * in_r;
        for (int idx = tid; idx < NPIX_HALO * 8; idx += 256) {
            const int q = idx >> 3, c = idx & 7;
            const int yy = q / HALO_W, xx = q - yy * HALO_W;
            const int iy = ty0 - 1 + yy, ix = tx0 - 1 + xx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = *reinterpret_cast<const u32x4*>(xb + ((size_t)(iy * in_r + si) * Win + (ix * in_r + sj)) * 64 + c * 8);
            *reinterpret_cast<u32x4*>(in_lds + swz128(q, c)) = v;
        }
    };
    stage_input(0);

    u32x4 wreg[WREGS];
    auto wload = [&](int it) {
#pragma unroll
        for (int u = 0; u < WREGS; ++u) {
            const int idx = tid + u * 256;
            if (idx < WCH) wreg[u] = *reinterpret_cast<const u32x4*>(wp + (size_t)it * WROWS * 64 + idx * 8);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < WREGS; ++u) {
            const int idx = tid + u * 256;
            if (idx < WCH) *reinterpret_cast<u32x4*>(w_lds + buf * WSLAB + swz128(idx >> 3, idx & 7)) = wreg[u];
        }
    };

    // halo index of this lane's pixel for each of the wave's 4 pixel groups (tap (0,0) corner)
    int qb[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) qb[pg] = (2 * wave + (pg >> 1)) * HALO_W + (pg & 1) * 16 + p;

    f32x4 acc[4][CT];
    const int total = ntiles * nci * 9;
    wload(0);
    wstore(0);
    __syncthreads();

    for (int it = 0; it < total; ++it) {
        const int tap = it % 9, chunk_it = it / 9;
        const int ci = chunk_it % nci, nt = chunk_it / nci;
        const int buf = it & 1;
        if (tap == 0 && it > 0 && nci > 1) {      // next input-channel chunk (all waves are past the last barrier)
            stage_input(ci);
            __syncthreads();
        }
        if (it + 1 < total) wload(it + 1);
        if (tap == 0 && ci == 0) {
#pragma unroll
            for (int pg = 0; pg < 4; ++pg)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int dy = tap / 3, dx = tap - dy * 3;
        const int qoff = dy * HALO_W + dx;
        const char* wb = w_lds + buf * WSLAB;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int chunk = kh * 4 + g;
            bf16x8 pf[4], wf[CT];
#pragma unroll
            for (int pg = 0; pg < 4; ++pg)
                pf[pg] = *reinterpret_cast<const bf16x8*>(in_lds + swz128(qb[pg] + qoff, chunk));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                wf[ct] = *reinterpret_cast<const bf16x8*>(wb + swz128(ct * 16 + p, chunk));
#pragma unroll
            for (int pg = 0; pg < 4; ++pg)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[pg][ct] = mfma16x16x32(wf[ct], pf[pg], acc[pg][ct]);
        }
        if (it + 1 < total) wstore(buf ^ 1);
        __syncthreads();

        if (tap == 8 && ci == nci - 1) {
            // ---- epilogue for cout tile nt ----
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const int oy = ty0 + 2 * wave + (pg >> 1);
                const int ox = tx0 + (pg & 1) * 16 + p;
                if (oy >= H || ox >= W) continue;
                if constexpr (OUT_MODE == OUT_NHWC_BF16) {
                    // lane holds channels g*16 + ct*4 + reg (host packed the weight rows that way)
                    const int si = nt / r, sj = nt - si * r;
                    const int Hr = H * r, Wr = W * r;
                    bf16_t* o = reinterpret_cast<bf16_t*>(out) +
                                (((size_t)b * Hr + (oy * r + si)) * Wr + (ox * r + sj)) * 64 + g * 16;
                    uint32_t pk[8];
                    // optional fused "+ add" and "* (mask > 0)" (ReLU backward) on same-shaped NHWC tensors
                    const size_t eoff = (((size_t)b * Hr + (oy * r + si)) * Wr + (ox * r + sj)) * 64 + g * 16;
                    uint32_t aw[8], mw[8];
                    if (add) {
                        const u32x4 a0 = *reinterpret_cast<const u32x4*>(add + eoff), a1 = *reinterpret_cast<const u32x4*>(add + eoff + 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { aw[q] = a0[q]; aw[4 + q] = a1[q]; }
                    }
                    if (mask) {
                        const u32x4 m0 = *reinterpret_cast<const u32x4*>(mask + eoff), m1 = *reinterpret_cast<const u32x4*>(mask + eoff + 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { mw[q] = m0[q]; mw[4 + q] = m1[q]; }
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[pg][ct][e];
                            if (bias) v[e] += bias[nt * 64 + g * 16 + ct * 4 + e];
                            if (relu) v[e] = fmaxf(v[e], 0.f);
                            const int wi = (ct * 4 + e) >> 1;
                            if (add) v[e] += __builtin_bit_cast(float, (e & 1) ? (aw[wi] & 0xffff0000u) : (aw[wi] << 16));
                            if (mask) {
                                const float mv = __builtin_bit_cast(float, (e & 1) ? (mw[wi] & 0xffff0000u) : (mw[wi] << 16));
                                if (!(mv > 0.f)) v[e] = 0.f;
                            }
                        }
                        pk[ct * 2 + 0] = pack_bf16x2(v[0], v[1]);
                        pk[ct * 2 + 1] = pack_bf16x2(v[2], v[3]);
                    }
                    if constexpr (CT == 4) {
                        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                    }
                } else {
                    // thin output: rows 4g+reg are the real couts (< cout_valid); fp32 planar NCHW
                    float* o = reinterpret_cast<float*>(out);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = 4 * g + e;
                        if (co < cout_valid) {
                            float v = acc[pg][0][e];
                            if (bias) v += bias[co];
                            if (relu) v = fmaxf(v, 0.f);
                            o[(((size_t)b * cout_valid + co) * H + oy) * W + ox] = v;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

// x: [B][H*in_r][W*in_r][64] bf16 (in_r = 1: plain 64-channel map; in_r > 1: 64*in_r^2 logical input
// channels stored pixel-shuffled, i.e. the gradient of an Upsampler stage's output -- this makes the
// same kernel the input-gradient conv of the up-convs).  wp: [ntiles][in_r^2][9][rows][64] bf16.
// out_mode 0: out = [B][H*r][W*r][64] bf16, ntiles = r*r, bias fp32 [ntiles][64] or NULL; optional
//   add / mask (bf16, same shape as out): out = (conv + bias [relu] + add) * (mask > 0).
// out_mode 1: out = [B][cout_valid][H][W] fp32, cout_valid <= 16, ntiles = 1, r = 1.
extern "C" int tup_conv3x3_c64_fwd(const void* x, const void* wp, const float* bias, const void* add,
                                   const void* mask, void* out, int B, int H, int W, int ntiles, int r,
                                   int cout_valid, int relu, int out_mode, int in_r, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (in_r < 1 || in_r > 6) return (int)hipErrorInvalidValue;
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long nblk = (long long)tilesX * tilesY * B;
    if (nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (out_mode == OUT_NHWC_BF16) {
        if (ntiles != r * r || r < 1) return (int)hipErrorInvalidValue;
        const size_t lds = IN_TILE_BYTES + 2 * 64 * 128;
        conv3x3_c64_kernel<4, OUT_NHWC_BF16><<<dim3((unsigned)nblk), dim3(256), lds, s>>>(
            (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)add, (const bf16_t*)mask, out, H, W, ntiles, r,
            64, relu, tilesX, tilesY, in_r);
    } else if (out_mode == OUT_PLANAR_F32) {
        if (ntiles != 1 || r != 1 || cout_valid < 1 || cout_valid > 16 || add || mask) return (int)hipErrorInvalidValue;
        const size_t lds = IN_TILE_BYTES + 2 * 16 * 128;
        conv3x3_c64_kernel<1, OUT_PLANAR_F32><<<dim3((unsigned)nblk), dim3(256), lds, s>>>(
            (const bf16_t*)x, (const bf16_t*)wp, bias, nullptr, nullptr, out, H, W, 1, 1, cout_valid, relu, tilesX,
            tilesY, in_r);
    } else {
        return (int)hipErrorInvalidValue;
    }
    TUP_CHECK_LAUNCH();
    return 0;
}
