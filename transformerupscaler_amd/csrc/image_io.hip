// Frame pre/post-processing either side of the model, HBM-bound byte work:
//   u8 HWC (RGB or BGR) -> fp32 planar CHW / 255      torchvision ToTensor on a uint8 image
//                                                     (reference data_handling/data_class.py:61-71, inference.py:65-75)
//   fp32 planar CHW -> u8 HWC (RGB or BGR)            (x * 255).clamp(0, 255).to(uint8).permute(1, 2, 0)[..., [2, 1, 0]]
//                                                     (reference app_overlay.py:381-388; ToPILImage's mul(255).byte(),
//                                                     inference.py:123-124)
// Each lane handles 4 consecutive pixels: three 4-byte words on the interleaved side, one 16-byte vector per plane
// on the planar side, so both sides move whole dwords.  Results are bit-exact with the torch expressions
// (IEEE fp32 divide / multiply, truncating float -> u8 conversion).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                              long long hw, int swap_rb)
{
    const int b = blockIdx.y;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;          // group of 4 pixels
    const long long p0 = q * 4;
    if (p0 >= hw) return;
    const uint8_t* s = src + (size_t)b * hw * 3 + p0 * 3;
    float* d = dst + (size_t)b * hw * 3 + p0;
    const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;
    if (p0 + 4 <= hw && ((size_t)b * hw * 3) % 4 == 0 && hw % 4 == 0) {
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(s), w1 = *reinterpret_cast<const uint32_t*>(s + 4),
                       w2 = *reinterpret_cast<const uint32_t*>(s + 8);
        uint8_t v[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (w0 >> (8 * i)) & 0xff; v[4 + i] = (w1 >> (8 * i)) & 0xff; v[8 + i] = (w2 >> (8 * i)) & 0xff; }
        f32x4 o0, o1, o2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o0[i] = (float)v[3 * i + c0] / 255.0f;
            o1[i] = (float)v[3 * i + 1] / 255.0f;
            o2[i] = (float)v[3 * i + c2] / 255.0f;
        }
        *reinterpret_cast<f32x4*>(d) = o0;
        *reinterpret_cast<f32x4*>(d + hw) = o1;
        *reinterpret_cast<f32x4*>(d + 2 * hw) = o2;
    } else {
        for (int i = 0; i < 4 && p0 + i < hw; ++i) {
            d[i] = (float)s[3 * i + c0] / 255.0f;
            d[hw + i] = (float)s[3 * i + 1] / 255.0f;
            d[2 * hw + i] = (float)s[3 * i + c2] / 255.0f;
        }
    }
}

TUP_DEVICE uint32_t to_u8(float x) {
    const float v = fminf(fmaxf(x * 255.0f, 0.0f), 255.0f);                 // NaN -> 0 like clamp on a NaN-free model output
    return (uint32_t)v;                                                     // truncation, as tensor.to(torch.uint8)
}

__global__ __launch_bounds__(256) void f32chw_to_u8hwc_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst,
                                                              long long hw, int swap_rb)
{
    const int b = blockIdx.y;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long p0 = q * 4;
    if (p0 >= hw) return;
    const float* s = src + (size_t)b * hw * 3 + p0;
    uint8_t* d = dst + (size_t)b * hw * 3 + p0 * 3;
    const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;
    if (p0 + 4 <= hw && hw % 4 == 0) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s + (size_t)c0 * hw), g = *reinterpret_cast<const f32x4*>(s + hw),
                    c = *reinterpret_cast<const f32x4*>(s + (size_t)c2 * hw);
        uint32_t v[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[3 * i] = to_u8(a[i]); v[3 * i + 1] = to_u8(g[i]); v[3 * i + 2] = to_u8(c[i]); }
        *reinterpret_cast<uint32_t*>(d) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        *reinterpret_cast<uint32_t*>(d + 4) = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
        *reinterpret_cast<uint32_t*>(d + 8) = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);
    } else {
        for (int i = 0; i < 4 && p0 + i < hw; ++i) {
            d[3 * i] = (uint8_t)to_u8(s[(size_t)c0 * hw + i]);
            d[3 * i + 1] = (uint8_t)to_u8(s[hw + i]);
            d[3 * i + 2] = (uint8_t)to_u8(s[(size_t)c2 * hw + i]);
        }
    }
}

// ---- transforms.Resize on a uint8 image = Pillow's two-pass 8-bit resampler (reference data_handling/data_class.py:61-71,
// inference.py:65-75).  Integer arithmetic, bit-exact with Pillow: out = clip8((2^21 + sum(pixel * k)) >> 22), k = the
// normalised triangle weights with 22 fractional bits (tables from resize_taps.pil_bilinear_coeffs). ----
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;
TUP_DEVICE uint32_t clip8(int v) { v >>= PIL_PRECISION_BITS; return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// horizontal pass: src u8 [B][H][W][3] -> dst u8 [B][H][Wo][3]; a thread owns one output pixel (its taps are 3*n contiguous bytes)
__global__ __launch_bounds__(256) void resize_u8_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                             const int* __restrict__ kk, int ksize, int H, int W, int Wo)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (ox >= Wo || y >= H) return;
    const uint8_t* s = src + ((size_t)((size_t)b * H + y) * W + xmin[ox]) * 3;
    const int* k = kk + (size_t)ox * ksize;
    const int n = xsize[ox];
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int kv = k[i];
        s0 += (int)s[3 * i] * kv; s1 += (int)s[3 * i + 1] * kv; s2 += (int)s[3 * i + 2] * kv;
    }
    uint8_t* d = dst + ((size_t)((size_t)b * H + y) * Wo + ox) * 3;
    d[0] = (uint8_t)clip8(s0); d[1] = (uint8_t)clip8(s1); d[2] = (uint8_t)clip8(s2);
}

// vertical pass: src u8 [B][H][W][3] -> dst u8 [B][Ho][W][3] and / or fp32 planar [B][3][Ho][W] = value / 255 (ToTensor fused)
__global__ __launch_bounds__(256) void resize_u8_cols_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst_u8,
                                                             float* __restrict__ dst_f32, const int* __restrict__ ymin,
                                                             const int* __restrict__ ysize, const int* __restrict__ kk, int ksize,
                                                             int H, int W, int Ho, int swap_rb)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || oy >= Ho) return;
    const uint8_t* s = src + ((size_t)((size_t)b * H + ymin[oy]) * W + x) * 3;
    const int* k = kk + (size_t)oy * ksize;
    const int n = ysize[oy];
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int kv = k[i];
        const uint8_t* r = s + (size_t)i * W * 3;
        s0 += (int)r[0] * kv; s1 += (int)r[1] * kv; s2 += (int)r[2] * kv;
    }
    const uint32_t v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2);
    if (dst_u8) {
        uint8_t* d = dst_u8 + ((size_t)((size_t)b * Ho + oy) * W + x) * 3;
        d[0] = (uint8_t)v0; d[1] = (uint8_t)v1; d[2] = (uint8_t)v2;
    }
    if (dst_f32) {
        const size_t hw = (size_t)Ho * W;
        float* d = dst_f32 + (size_t)b * 3 * hw + (size_t)oy * W + x;
        d[0] = (float)(swap_rb ? v2 : v0) / 255.0f; d[hw] = (float)v1 / 255.0f; d[2 * hw] = (float)(swap_rb ? v0 : v2) / 255.0f;
    }
}

}  // namespace

// Pillow's horizontal 8-bit resampling pass (Image.resize(..., BILINEAR) on an RGB image, the arithmetic behind
// transforms.Resize on a PIL image: reference data_handling/data_class.py:61-71, inference.py:65-75).
// src u8 [B][H][W][3] -> dst u8 [B][H][Wo][3]; xmin / xsize int32 [Wo], k int32 [Wo][ksize] (22 fractional bits).
extern "C" int tup_resize_u8_rows(const void* src, void* dst, const int* xmin, const int* xsize, const int* k, int ksize,
                                  int B, int H, int W, int Wo, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || Wo <= 0) return 0;
    if (B > 65535 || ksize < 1 || (H + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
    resize_u8_rows_kernel<<<dim3((Wo + 63) / 64, (H + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const uint8_t*)src, (uint8_t*)dst, xmin, xsize, k, ksize, H, W, Wo);
    TUP_CHECK_LAUNCH();
    return 0;
}

// ... and its vertical pass, optionally fused with ToTensor: src u8 [B][H][W][3] -> dst_u8 u8 [B][Ho][W][3] (may be NULL) and /
// or dst_f32 fp32 [B][3][Ho][W] = value / 255 (may be NULL; swap_rb = 1 reads BGR into RGB planes).
extern "C" int tup_resize_u8_cols(const void* src, void* dst_u8, float* dst_f32, const int* ymin, const int* ysize, const int* k,
                                  int ksize, int B, int H, int W, int Ho, int swap_rb, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0) return 0;
    if (B > 65535 || ksize < 1 || (Ho + 3) / 4 > 65535 || (!dst_u8 && !dst_f32)) return (int)hipErrorInvalidValue;
    resize_u8_cols_kernel<<<dim3((W + 63) / 64, (Ho + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const uint8_t*)src, (uint8_t*)dst_u8, dst_f32, ymin, ysize, k, ksize, H, W, Ho, swap_rb);
    TUP_CHECK_LAUNCH();
    return 0;
}

// src u8 [B][H][W][3] -> dst fp32 [B][3][H][W] = src / 255; swap_rb = 1 reads BGR frames (screen grabs) into RGB planes.
extern "C" int tup_u8hwc_to_f32chw(const void* src, float* dst, int B, int H, int W, int swap_rb, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (B > 65535) return (int)hipErrorInvalidValue;
    const long long hw = (long long)H * W;
    const long long groups = (hw + 3) / 4;
    u8hwc_to_f32chw_kernel<<<dim3((unsigned)((groups + 255) / 256), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const uint8_t*)src, dst, hw, swap_rb);
    TUP_CHECK_LAUNCH();
    return 0;
}

// src fp32 [B][3][H][W] -> dst u8 [B][H][W][3] = trunc(clamp(src * 255, 0, 255)); swap_rb = 1 writes BGR.
extern "C" int tup_f32chw_to_u8hwc(const float* src, void* dst, int B, int H, int W, int swap_rb, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (B > 65535) return (int)hipErrorInvalidValue;
    const long long hw = (long long)H * W;
    const long long groups = (hw + 3) / 4;
    f32chw_to_u8hwc_kernel<<<dim3((unsigned)((groups + 255) / 256), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        src, (uint8_t*)dst, hw, swap_rb);
    TUP_CHECK_LAUNCH();
    return 0;
}
