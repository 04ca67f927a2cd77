// Kernels specific to the ResidualTransformer path (reference models/ResidualTransformer/model.py, BASELINE.json
// config 5): global multi-head attention over the 3600-token grid, LayerNorm(128), and the fused
// "bicubic(x) + bicubic(residual) -> clamp" output stage.  Everything else reuses the FastTransformer kernels.
//
//   tup_rt_attention_fwd     nn.MultiheadAttention core (q scale, QK^T, softmax, PV, head concat)  model.py:31,43
//   tup_layernorm128_fwd     nn.LayerNorm(128)                                                     model.py:30,32,42,47
//   tup_rt_bicubic_sum_fwd   F.interpolate(bicubic) x2 + add + clamp                               model.py:125,160-164
#include "common.h"

namespace {

constexpr int RD = 128, RH = 8, HD = 16;

// ---- flash-style attention, one wave per (batch, head, 64-query tile); S^T tiles in registers, online softmax ----
__global__ __launch_bounds__(256) void rt_attention_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int B, int N, int qtiles)
{
    __shared__ __attribute__((aligned(16))) bf16_t vlds[4][64 * HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    int wid = blockIdx.x * 4 + wave;
    const int total = B * RH * qtiles;
    const bool active = wid < total;
    if (!active) wid = total - 1;
    const int qt0 = wid % qtiles;
    const int h = (wid / qtiles) % RH;
    const int b = wid / (qtiles * RH);
    const bf16_t* base = qkv + (size_t)b * N * (3 * RD) + h * HD;
    const int q0 = qt0 * 64;

    s16x4 qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = min(q0 + 16 * t + p, N - 1);
        qf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + 4 * g);
    }
    f32x4 oacc[4];
    float mrun[4], lrun[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; mrun[t] = -INFINITY; lrun[t] = 0.f; }

    const int ktiles = (N + 63) / 64;
    for (int kt0 = 0; kt0 < ktiles; ++kt0) {
        const int k0 = kt0 * 64;
        // stage V tile [64 keys][16] for the transposed fragments (wave-private region)
        {
            const int row = min(k0 + lane, N - 1);
            const bf16_t* vr = base + (size_t)row * (3 * RD) + 2 * RD;
            *reinterpret_cast<u32x4*>(&vlds[wave][lane * HD]) = *reinterpret_cast<const u32x4*>(vr);
            *reinterpret_cast<u32x4*>(&vlds[wave][lane * HD + 8]) = *reinterpret_cast<const u32x4*>(vr + 8);
        }
        s16x4 kf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = min(k0 + 16 * t + p, N - 1);
            kf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + RD + 4 * g);
        }
        f32x4 st[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const f32x4 s = mfma16x16x16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + 16 * kt + 4 * g + e;
                    st[kt][qt][e] = key < N ? s[e] * 0.25f : -INFINITY;        // q * head_dim^-0.5
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        s16x4 vf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = vlds[wave][(16 * kt + 4 * g + j) * HD + p];
            vf[kt] = __builtin_bit_cast(s16x4, t);
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            float mx = mrun[qt];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, st[kt][qt][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float alpha = __expf(mrun[qt] - mx);             // 0 on the first tile (mrun = -inf)
            float sum = 0.f;
            f32x4 o = oacc[qt];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { pv[e] = __expf(st[kt][qt][e] - mx); sum += pv[e]; }
                const u32x2 pp = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
                o = mfma16x16x16(vf[kt], __builtin_bit_cast(s16x4, pp), o);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            lrun[qt] = lrun[qt] * alpha + sum;
            mrun[qt] = mx;
            oacc[qt] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!active) return;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = q0 + 16 * qt + p;
        if (q >= N) continue;
        const float inv = 1.0f / lrun[qt];
        bf16_t* op = out + ((size_t)b * N + q) * RD + h * HD + 4 * g;
        *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(oacc[qt][0] * inv, oacc[qt][1] * inv), pack_bf16x2(oacc[qt][2] * inv, oacc[qt][3] * inv)};
    }
}

__global__ __launch_bounds__(256) void layernorm128_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, int M)
{
    const int sub = threadIdx.x & 15;                   // 16 lanes per row, 8 elements per lane
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < M;
    const float* xr = x + (size_t)(ok ? row : 0) * RD;
    f32x4 v[2];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        v[q] = *reinterpret_cast<const f32x4*>(xr + q * 64 + sub * 4);
        s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / RD);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mean; ss += d * d; }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss * (1.0f / RD) + 1e-5f);
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = q * 64 + sub * 4;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[q][e] - mean) * rstd * gm[e] + bt[e];
        *reinterpret_cast<u32x2*>(y + (size_t)row * RD + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
}

// out = clamp(bicubic(a) + bicubic(b)); tap tables: per output row/col 4 clamped source indices + weights
__global__ __launch_bounds__(256) void rt_bicubic_sum_kernel(
    const float* __restrict__ a, const float* __restrict__ bsrc, float* __restrict__ out,
    const int* __restrict__ ayi, const float* __restrict__ ayw, const int* __restrict__ axi, const float* __restrict__ axw,
    const int* __restrict__ byi, const float* __restrict__ byw, const int* __restrict__ bxi, const float* __restrict__ bxw,
    int Ha, int Wa, int Hb, int Wb, int Ho, int Wo, int clamp01)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int plane = blockIdx.z;
    if (ox >= Wo || oy >= Ho) return;
    const float* pa = a + (size_t)plane * Ha * Wa;
    const float* pb = bsrc + (size_t)plane * Hb * Wb;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* ra = pa + (size_t)ayi[oy * 4 + i] * Wa;
        const float* rb = pb + (size_t)byi[oy * 4 + i] * Wb;
        float ha = 0.f, hb = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ha = fmaf(axw[ox * 4 + j], ra[axi[ox * 4 + j]], ha);
            hb = fmaf(bxw[ox * 4 + j], rb[bxi[ox * 4 + j]], hb);
        }
        acc = fmaf(ayw[oy * 4 + i], ha, acc);
        acc = fmaf(byw[oy * 4 + i], hb, acc);
    }
    if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
    out[((size_t)plane * Ho + oy) * Wo + ox] = acc;
}

}  // namespace

// qkv bf16 [B][N][384] (q | k | v, each 8 heads x 16); out bf16 [B][N][128].  Eval-mode attention (no dropout).
extern "C" int tup_rt_attention_fwd(const void* qkv, void* out, int B, int N, void* stream)
{
    if (B <= 0 || N <= 0) return 0;
    const int qtiles = (N + 63) / 64;
    const long long waves = (long long)B * RH * qtiles;
    rt_attention_kernel<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)qkv, (bf16_t*)out, B, N, qtiles);
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_layernorm128_fwd(const float* x, const float* gamma, const float* beta, void* y, int M, void* stream)
{
    if (M <= 0) return 0;
    layernorm128_kernel<<<dim3((M + 15) / 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(x, gamma, beta, (bf16_t*)y, M);
    TUP_CHECK_LAUNCH();
    return 0;
}

// out fp32 [planes][Ho][Wo] = clamp(bicubic(a [planes][Ha][Wa]) + bicubic(b [planes][Hb][Wb])); index/weight tables
// int/float [Ho][4], [Wo][4] per source (aten upsample_bicubic2d, align_corners=False, A=-0.75, clamped indices).
extern "C" int tup_rt_bicubic_sum_fwd(const float* a, const float* b, float* out, const int* ayi, const float* ayw,
                                      const int* axi, const float* axw, const int* byi, const float* byw, const int* bxi,
                                      const float* bxw, int planes, int Ha, int Wa, int Hb, int Wb, int Ho, int Wo,
                                      int clamp01, void* stream)
{
    if (planes <= 0) return 0;
    if (planes > 65535) return (int)hipErrorInvalidValue;
    dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, planes);
    rt_bicubic_sum_kernel<<<grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        a, b, out, ayi, ayw, axi, axw, byi, byw, bxi, bxw, Ha, Wa, Hb, Wb, Ho, Wo, clamp01);
    TUP_CHECK_LAUNCH();
    return 0;
}
