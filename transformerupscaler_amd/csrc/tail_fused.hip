// Fused output tail for inference (gfx950): the last final_upscale stage (Conv2d(3, 3rr, 3) + PixelShuffle(r)),
// final_upscale_conv (Conv2d(3, 3, 3)), "+ upscaled_input", the antialiased Resize and the clamp
// (reference models/FastTransformer/model.py:316-327, utils.py:62-63,74-75,83-84) as ONE kernel.
//
// All of it is 3-channel work: unfused it is four HBM round trips over HR-sized fp32 planes (t1, sum, out)
// for ~170 MACs per pixel.  Here one workgroup owns a 16x64 tile of the FINAL image and walks the stencil
// chain backwards through LDS: the HR window its resize taps touch (+1 halo for the 3x3), the LR window
// under that (+1 halo) -- three small LDS tiles, three barriers, and HBM sees only the LR residual, the
// upscaled_input plane and the output.  Bound: HBM.
#include "common.h"
#include <stdlib.h>

namespace {

#ifndef TUP_TAIL_OT_H
#define TUP_TAIL_OT_H 16
#endif
// 16 x 64 output tiles, 8 waves, two workgroups per CU.  Measured alternatives (round 2, same box, rocprofv3): 8 x 64 tiles / 4
// waves / four workgroups per CU 779 us, 32 x 64 tiles / 16 waves / one workgroup per CU 738 us, this geometry 685 us; with
// 170 or 256 VGPRs instead of 128 (no scratch) every geometry is within 3 % of its 128-VGPR time.
constexpr int OT_H = TUP_TAIL_OT_H, OT_W = 64, NT = 32 * OT_H, NROW = NT / 64;
__device__ unsigned long long tup_tail_stamps[16];        // timing experiments (TUP_TAIL_STAMPS=1): s_memtime per stage, one workgroup

struct TailParams {
    const float* x;            // [B][3][H][W] input of the last final_upscale stage
    const float* wfu;          // [3rr][28] (27 taps (cin, ky, kx) + pad)
    const float* bfu;          // [3rr]
    const float* wfc;          // [3][28]
    const float* bfc;          // [3]
    const float* ui;           // [B][3][H*r][W*r] upscaled_input (already ReLU'd)
    float* out;                // [B][3][Ho][Wo]
    const int* ymin; const int* ysize; const float* yw; int KY;
    const int* xmin; const int* xsize; const float* xw; int KX;
    int H, W, r, Ho, Wo, EH, EW, LH, LW, clamp01, stamps, abl;
};

// STAMPS is a build variant, and the stamps are parked in LDS until the end of the kernel: a (conditional) global store
// ahead of the weight loads makes hipcc treat global memory as clobbered and fetch the wave-uniform weights with vector
// loads into VGPRs instead of s_load.
template <bool S> __device__ unsigned long long* stamp_buf()
{
    if constexpr (S) { __shared__ unsigned long long b[16]; return b; } else return nullptr;
}

template <int OCC, bool STAMPS>
__global__ __launch_bounds__(NT, OCC) void tail_fused_kernel(const TailParams p)
{
    extern __shared__ __attribute__((aligned(16))) float fl[];
    unsigned long long* st_lds = stamp_buf<STAMPS>();
    int nst = 0;
    auto stamp = [&]() {
        if constexpr (STAMPS) {
            if (threadIdx.x == 0) st_lds[nst] = __builtin_amdgcn_s_memtime();
            ++nst;
        }
    };
    stamp();
    const int r = p.r, rr = r * r, nfu = 3 * rr;
    const int Hs = p.H * r, Ws = p.W * r;
    const int TH1 = p.EH + 2, TW1 = p.EW + 2;            // t1 tile (sum window + 3x3 halo)
    float* wfu = fl;                                    // [nfu][28]
    float* bfu = wfu + nfu * 28;                        // [nfu]
    float* wfc = bfu + ((nfu + 1) & ~1);                // [3][28] + [3] bias at wfc[84..86]  (every array starts 8-B aligned,
                                                        //  every row pitch is even: the pixel-pair reads below are ds_read_b64)
    float* lr = wfc + 88;                               // [3][LH][LW]
    float* t1 = lr + 3 * p.LH * p.LW;                   // [3][TH1][TW1]
    float* sm = t1 + 3 * TH1 * TW1;                     // [3][EH][EW]
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * OT_H, ox0 = blockIdx.x * OT_W;
    const int oy1 = min(oy0 + OT_H, p.Ho) - 1, ox1 = min(ox0 + OT_W, p.Wo) - 1;
    // HR window touched by this tile's resize taps (tables are monotone)
    const int hy0 = p.ymin[oy0], hx0 = p.xmin[ox0];
    const int eh = p.ymin[oy1] + p.ysize[oy1] - hy0, ew = p.xmin[ox1] + p.xsize[ox1] - hx0;
    // LR window under the t1 tile [hy0-1, hy0+eh] x [hx0-1, hx0+ew] (clipped), plus the 3x3 halo
    const int ly0 = max(hy0 - 1, 0) / r - 1, lx0 = max(hx0 - 1, 0) / r - 1;

    for (int i = tid; i < nfu * 28; i += NT) wfu[i] = p.wfu[i];
    for (int i = tid; i < nfu; i += NT) bfu[i] = p.bfu[i];
    if (tid < 84) wfc[tid] = p.wfc[tid];
    // ---- stage A: LR window (zero outside the image = the conv's zero padding) ----
    // (all tile loops below are 2-D with power-of-two thread strides: runtime integer divisions cost ~40
    //  instructions each and used to outweigh the arithmetic)
    const int t_row = tid >> 6, t_col = tid & 63;
    {
        // all of a thread's window loads are issued before any is stored: a load -> LDS-store loop with run-time bounds
        // is not unrolled and pays one global round trip per iteration (stamped: 8.7 k of the tile's 39 k cycles).
        // The upscaled_input window goes straight to its final place (sm; stage C adds the conv to it): every global
        // read of the tile is in flight at once, here, and overlaps the other resident workgroup's arithmetic.
        constexpr int AMAX = 6, UMAX = 14;
        const int total = 3 * p.LH * p.LW, plane = p.LH * p.LW;
        const int hw = eh * ew, utotal = 3 * hw, splane = p.EH * p.EW;
        const float inv_hw = 1.0f / (float)hw, inv_ew = 1.0f / (float)ew;      // exact quotients for these small integers
        const size_t cstride = (size_t)Hs * Ws;
        const float* uib = p.ui + ((size_t)b * 3 * Hs + hy0) * Ws + hx0;
        auto ui_pos = [&](int idx, int& c, int& sy, int& sx) {
            c = (int)(((float)idx + 0.5f) * inv_hw);
            const int rem = idx - c * hw;
            sy = (int)(((float)rem + 0.5f) * inv_ew);
            sx = rem - sy * ew;
        };
        float v[AMAX], uv[UMAX];
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            const int idx = tid + k * NT;
            v[k] = 0.f;
            if (idx < total) {
                const int c = idx / plane, q = idx - c * plane;
                const int yy = q / p.LW, xx = q - yy * p.LW;
                const int iy = ly0 + yy, ix = lx0 + xx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v[k] = p.x[(((size_t)b * 3 + c) * p.H + iy) * p.W + ix];
            }
        }
#pragma unroll
        for (int k = 0; k < UMAX; ++k) {
            const int idx = tid + k * NT;
            uv[k] = 0.f;
            if (idx < utotal) {
                int c, sy, sx;
                ui_pos(idx, c, sy, sx);
                uv[k] = uib[c * cstride + (size_t)sy * Ws + sx];
            }
        }
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            const int idx = tid + k * NT;
            if (idx < total) lr[idx] = v[k];
        }
#pragma unroll
        for (int k = 0; k < UMAX; ++k) {
            const int idx = tid + k * NT;
            if (idx < utotal) {
                int c, sy, sx;
                ui_pos(idx, c, sy, sx);
                sm[c * splane + sy * p.EW + sx] = uv[k];
            }
        }
        for (int idx = tid + AMAX * NT; idx < total; idx += NT) {      // larger windows (not reached for r <= 6 tiles)
            const int c = idx / plane, q = idx - c * plane;
            const int yy = q / p.LW, xx = q - yy * p.LW;
            const int iy = ly0 + yy, ix = lx0 + xx;
            lr[idx] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? p.x[(((size_t)b * 3 + c) * p.H + iy) * p.W + ix] : 0.f;
        }
        for (int idx = tid + UMAX * NT; idx < utotal; idx += NT) {     // larger windows (strong down-scaling ratios)
            int c, sy, sx;
            ui_pos(idx, c, sy, sx);
            sm[c * splane + sy * p.EW + sx] = uib[c * cstride + (size_t)sy * Ws + sx];
        }
    }
    stamp();          // 1: stage A loads issued + written
    __syncthreads();
    stamp();          // 2: barrier
    // ---- stage B: t1 = PixelShuffle(conv3x3(x)) on the haloed HR window (zero outside the HR image).
    // A thread owns ONE sub-pixel phase (si, sj): its 81 weights live in registers and it walks LR pixels, so
    // the inner loop is 27 LDS reads per 81 FMAs (weights from LDS per FMA made this stage LDS-issue bound). ----
    for (int i = tid; i < 3 * TH1 * TW1; i += NT) t1[i] = 0.f;
    __syncthreads();
    stamp();          // 3: t1 zeroed
    {
        // A WAVE owns one sub-pixel phase (si, sj) at a time, so its 81 weights are wave-uniform (all 81 as scalar
        // operands need 81 + ~25 SGPRs: hipcc then spills SGPRs to VGPR lanes and pays a v_readlane per FMA; all 81 in
        // VGPRs push the loop past the 128 registers of two workgroups per CU -- hence the split below).  The 64 lanes
        // walk the LR pixels of the window two at a time: the 3x4 taps of a (ci, ky) row give the pairs (v0,v1),
        // (v1,v2), (v2,v3), so every weight is one v_pk_fma_f32 (weight broadcast through op_sel) for two pixels.
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lplane = p.LH * p.LW;
        const int Ya = max(hy0 - 1, 0), Yb = min(hy0 + eh, Hs - 1), Xa = max(hx0 - 1, 0), Xb = min(hx0 + ew, Ws - 1);
        // rr < 8 phases (r = 1, 2): NROW / rr waves share a phase and interleave its pixel pairs
        const int nparts = rr < NROW ? NROW / rr : 1;
        const int part = wv / rr;                       // 0 when rr >= NROW
        for (int ph = wv % rr; ph < rr && part < nparts && !(p.abl & 1); ph += NROW) {
            const int si = ph / r, sj = ph - si * r;
            // LR pixels whose phase-(si,sj) child lies in the t1 window [hy0-1, hy0+eh] x [hx0-1, hx0+ew] and the image
            const int ya = (Ya - si + r - 1) / r, yb = (Yb - si >= 0) ? (Yb - si) / r : -1;
            const int xa = (Xa - sj + r - 1) / r, xb = (Xb - sj >= 0) ? (Xb - sj) / r : -1;
            // pairs start on an even column of the LR tile (8-B aligned, conflict-free ds_read_b64): one pixel early if needed
            const int x0 = xa - ((xa - 1 - lx0) & 1);
            const int npair = (xb - x0 + 2) >> 1, nq = (yb - ya + 1) * npair;
            if (npair <= 0 || yb < ya) continue;
            const float inv_np = 1.0f / (float)npair;
            // register budget: one output channel's weights as scalar operands (s_load from the kernel argument: the
            // phase is wave-uniform), the other two channels' in VGPRs (read from the LDS copy) -- 27 SGPRs + 54 VGPRs
            const float* __restrict__ w0 = p.wfu + ph * 28;
            float w1[27], w2[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) { w1[k] = wfu[(rr + ph) * 28 + k]; w2[k] = wfu[(2 * rr + ph) * 28 + k]; }
            const float bb0 = bfu[ph], bb1 = bfu[rr + ph], bb2 = bfu[2 * rr + ph];
            for (int q = part * 64 + t_col; q < nq; q += 64 * nparts) {
                const int yr = (int)(((float)q + 0.5f) * inv_np);
                const int y = ya + yr, x = x0 + 2 * (q - yr * npair);
                const float* base = lr + (y - 1 - ly0) * p.LW + (x - 1 - lx0);
                f32x2 a0 = {bb0, bb0}, a1 = {bb1, bb1}, a2 = {bb2, bb2};
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float* rp = base + ci * lplane + ky * p.LW;
                        const f32x2 lo = *reinterpret_cast<const f32x2*>(rp), hi = *reinterpret_cast<const f32x2*>(rp + 2);
                        const f32x2 pr[3] = {lo, f32x2{lo[1], hi[0]}, hi};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int k = ci * 9 + ky * 3 + kx;
                            a0 = __builtin_elementwise_fma(f32x2{w0[k], w0[k]}, pr[kx], a0);
                            a1 = __builtin_elementwise_fma(f32x2{w1[k], w1[k]}, pr[kx], a1);
                            a2 = __builtin_elementwise_fma(f32x2{w2[k], w2[k]}, pr[kx], a2);
                        }
                    }
                const int ty = y * r + si - (hy0 - 1), tx = x * r + sj - (hx0 - 1);
                float* tp = t1 + ty * TW1 + tx;
                if (x >= xa) { tp[0] = a0[0]; tp[TH1 * TW1] = a1[0]; tp[2 * TH1 * TW1] = a2[0]; }
                if (x + 1 <= xb) { tp[r] = a0[1]; tp[TH1 * TW1 + r] = a1[1]; tp[2 * TH1 * TW1 + r] = a2[1]; }
            }
        }
    }
    stamp();          // 4: stage B done
    __syncthreads();
    stamp();          // 5: barrier
    // ---- stage C: sum = conv3x3(t1) + bias + upscaled_input on the HR window; the t1 taps come from LDS ----
    {
        const float* __restrict__ wg = p.wfc;          // channel 0: scalar operands; channels 1, 2: VGPRs (see stage B)
        float wg1[27], wg2[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) { wg1[k] = wfc[28 + k]; wg2[k] = wfc[56 + k]; }
        const float b0 = p.bfc[0], b1 = p.bfc[1], b2 = p.bfc[2];
        // two adjacent HR pixels per thread (packed FMAs as in stage B); the (row, pair) space is walked flat so that
        // the 512 threads stay busy whatever the window shape (rows = q / npair through an exact float reciprocal)
        const int npair = (ew + 1) >> 1, nq = eh * npair;
        const float inv_np = 1.0f / (float)npair;
        const int tplane = TH1 * TW1, splane = p.EH * p.EW;
        for (int q = tid; q < nq && !(p.abl & 2); q += NT) {
            const int sy = (int)(((float)q + 0.5f) * inv_np);
            const int sx = 2 * (q - sy * npair);
            const bool two = sx + 1 < ew;
            f32x2 a0 = {b0, b0}, a1 = {b1, b1}, a2 = {b2, b2};
            const float* base = t1 + sy * TW1 + sx;           // t1 tile origin is (hy0-1, hx0-1)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* rp = base + ci * tplane + ky * TW1;
                    const f32x2 lo = *reinterpret_cast<const f32x2*>(rp), hi = *reinterpret_cast<const f32x2*>(rp + 2);
                    const f32x2 pr[3] = {lo, f32x2{lo[1], hi[0]}, hi};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int k = ci * 9 + ky * 3 + kx;
                        a0 = __builtin_elementwise_fma(f32x2{wg[k], wg[k]}, pr[kx], a0);
                        a1 = __builtin_elementwise_fma(f32x2{wg1[k], wg1[k]}, pr[kx], a1);
                        a2 = __builtin_elementwise_fma(f32x2{wg2[k], wg2[k]}, pr[kx], a2);
                    }
                }
            float* sp = sm + sy * p.EW + sx;                  // holds upscaled_input since stage A
            const f32x2 u0 = *reinterpret_cast<const f32x2*>(sp), u1 = *reinterpret_cast<const f32x2*>(sp + splane),
                        u2 = *reinterpret_cast<const f32x2*>(sp + 2 * splane);
            if (two) {
                *reinterpret_cast<f32x2*>(sp) = a0 + u0;
                *reinterpret_cast<f32x2*>(sp + splane) = a1 + u1;
                *reinterpret_cast<f32x2*>(sp + 2 * splane) = a2 + u2;
            } else {
                sp[0] = a0[0] + u0[0]; sp[splane] = a1[0] + u1[0]; sp[2 * splane] = a2[0] + u2[0];
            }
        }
    }
    stamp();          // 6: stage C done
    __syncthreads();
    stamp();          // 7: barrier
    // ---- stage D: antialiased resize taps + clamp: a thread owns one output column (its x taps are loaded once) ----
    {
        const int ox = ox0 + t_col;
        if (ox < p.Wo && !(p.abl & 4)) {
            const int x0 = p.xmin[ox] - hx0, nx = p.xsize[ox];
            float wx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wx[j] = (j < nx && j < p.KX) ? p.xw[ox * p.KX + j] : 0.f;
            const int nxc = nx < 8 ? nx : 8;
            for (int oy = oy0 + t_row; oy <= oy1; oy += NROW) {
                const int y0 = p.ymin[oy] - hy0, ny = p.ysize[oy];
                float acc[3] = {0.f, 0.f, 0.f};
                for (int a = 0; a < ny; ++a) {
                    const float wy = p.yw[oy * p.KY + a];
                    float h[3] = {0.f, 0.f, 0.f};
                    const float* s0 = sm + (y0 + a) * p.EW + x0;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < nxc) {
                            h[0] = fmaf(wx[j], s0[j], h[0]); h[1] = fmaf(wx[j], s0[p.EH * p.EW + j], h[1]);
                            h[2] = fmaf(wx[j], s0[2 * p.EH * p.EW + j], h[2]);
                        }
                    acc[0] = fmaf(wy, h[0], acc[0]); acc[1] = fmaf(wy, h[1], acc[1]); acc[2] = fmaf(wy, h[2], acc[2]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = acc[c];
                    if (p.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                    p.out[(((size_t)b * 3 + c) * p.Ho + oy) * p.Wo + ox] = v;
                }
            }
        }
    }
    stamp();          // 8: stage D done
    if constexpr (STAMPS) {
        if (blockIdx.x == 10 && blockIdx.y == 20 && blockIdx.z == 3 && threadIdx.x == 0)
            for (int i = 0; i < nst; ++i) tup_tail_stamps[i] = st_lds[i];
    }
}

}  // namespace

// x fp32 [B][3][H][W]; wfu fp32 [3rr][28], bfu [3rr]; wfc fp32 [3][28], bfc [3]; ui fp32 [B][3][H*r][W*r];
// out fp32 [B][3][Ho][Wo].  Tap tables as tup_resize_aa_fwd (identity tables when Ho x Wo == H*r x W*r).
// EH / EW: the largest HR window (rows / cols) any 16x64 output tile's taps touch (computed by the caller from
// the same tables; they size the LDS tiles).
extern "C" int tup_tail_fused_fwd(const float* x, const float* wfu, const float* bfu, const float* wfc, const float* bfc,
                                  const float* ui, float* out, const int* ymin, const int* ysize, const float* yw, int KY,
                                  const int* xmin, const int* xsize, const float* xw, int KX, int B, int H, int W, int r,
                                  int Ho, int Wo, int EH, int EW, int clamp01, void* stream)
{
    if (B <= 0) return 0;
    if (r < 1 || r > 6 || B > 65535 || EH < 1 || EW < 1 || KX > 8) return (int)hipErrorInvalidValue;
    TailParams p{};
    p.x = x; p.wfu = wfu; p.bfu = bfu; p.wfc = wfc; p.bfc = bfc; p.ui = ui; p.out = out;
    p.ymin = ymin; p.ysize = ysize; p.yw = yw; p.KY = KY; p.xmin = xmin; p.xsize = xsize; p.xw = xw; p.KX = KX;
    EW = (EW + 1) & ~1;                   // even row pitches (see the kernel's LDS layout)
    p.H = H; p.W = W; p.r = r; p.Ho = Ho; p.Wo = Wo; p.EH = EH; p.EW = EW; p.clamp01 = clamp01;
    static const int stamps_on = TUP_ENV_FLAG("TUP_TAIL_STAMPS") ? 1 : 0;
    p.stamps = stamps_on;
    static const int abl = TUP_ENV_INT("TUP_TAIL_ABLATE", 0);      // timing experiments only (results are wrong)
    p.abl = abl;
    p.LH = (EH + 2) / r + 4; p.LW = ((EW + 2) / r + 4 + 1) & ~1;
    const int nfu = 3 * r * r;
    const size_t lds = ((size_t)nfu * 28 + ((nfu + 1) & ~1) + 88 + 3 * (size_t)p.LH * p.LW + 3 * (size_t)(EH + 2) * (EW + 2) + 3 * (size_t)EH * EW + 2) * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    // OCC = waves per SIMD the register allocator must allow: 4 (<= 128 VGPRs, 16 waves per CU) or 2
    static const bool occ2 = TUP_ENV_FLAG("TUP_TAIL_OCC2");
    static const bool occ3 = TUP_ENV_FLAG("TUP_TAIL_OCC3");        // 170 VGPRs: no spills, three workgroups per CU
    dim3 grid((Wo + OT_W - 1) / OT_W, (Ho + OT_H - 1) / OT_H, B);
#define TUP_TAIL_LAUNCH(OCC, ST) do { \
        hipError_t e = hipFuncSetAttribute((const void*)tail_fused_kernel<OCC, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e; \
        tail_fused_kernel<OCC, ST><<<grid, dim3(NT), lds, reinterpret_cast<hipStream_t>(stream)>>>(p); } while (0)
#ifdef TUP_DIAG          // `make diag` only
    if (stamps_on && occ2) TUP_TAIL_LAUNCH(2, true);      // spill-free stamps (the <4, true> build spills 143 VGPRs: its shares mislead)
    else if (stamps_on) TUP_TAIL_LAUNCH(4, true);
    else
#endif
    if (occ2) TUP_TAIL_LAUNCH(2, false);
    else if (occ3) TUP_TAIL_LAUNCH(3, false);
    else TUP_TAIL_LAUNCH(4, false);
#undef TUP_TAIL_LAUNCH
    TUP_CHECK_LAUNCH();
    return 0;
}

#ifdef TUP_DIAG
// Timing experiments only (`make diag`): s_memtime stamps of the last launch under TUP_TAIL_STAMPS=1 (9 values).
extern "C" int tup_debug_tail_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tup_tail_stamps), sizeof(unsigned long long) * 16);
}
#endif
