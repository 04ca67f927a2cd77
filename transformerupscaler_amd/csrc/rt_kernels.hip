// Kernels specific to the ResidualTransformer path (reference models/ResidualTransformer/model.py, BASELINE.json
// config 5): global multi-head attention over the 3600-token grid, LayerNorm(128), and the fused
// "bicubic(x) + bicubic(residual) -> clamp" output stage.  Everything else reuses the FastTransformer kernels.
//
//   tup_rt_attention_fwd     nn.MultiheadAttention core (q scale, QK^T, softmax, PV, head concat)  model.py:31,43
//   tup_layernorm128_fwd     nn.LayerNorm(128)                                                     model.py:30,32,42,47
//   tup_rt_bicubic_sum_fwd   F.interpolate(bicubic) x2 + add + clamp                               model.py:125,160-164
#include "common.h"

namespace {

constexpr float SCALE_LOG2E = 0.25f * 1.44269504088896340736f;      // head_dim^-0.5 * log2(e)

// Two 4-element fragments -> one v_mfma_f32_16x16x32_bf16 operand (twice the rate of the x16 form on gfx950): k index
// 8g + j <-> element j of the first fragment (j < 4) or j - 4 of the second; both operands of a product use the same map,
// so any two 16-wide slices of the contracted axis can be paired.
TUP_DEVICE bf16x8 join4(s16x4 lo, s16x4 hi) {
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}


constexpr int RD = 128, RH = 8, HD = 16;

// ---- flash-style attention, one wave per (batch, head, 64-query tile); S^T tiles in registers, online softmax ----
// (151 registers = three workgroups per CU: the 912 workgroups of a B = 2 step are 1.19 rounds.  Bounded to four per CU -- 1,024 slots,
// one round -- the kernel spills 13 registers and is slower, 138 vs 130 us.)
template <bool DROP>
__global__ __launch_bounds__(256) void rt_attention_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int B, int N, int qtiles,
    uint32_t thresh, float inv_keep, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) bf16_t vlds[4][64 * HD];
    __shared__ __attribute__((aligned(16))) float comb_o[4][4][64][4];     // [wave][qt][lane][e] partial outputs
    __shared__ float comb_m[4][4][16], comb_l[4][4][16];                   // [wave][qt][query column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    // one workgroup per (image, head, 64-query tile); its 4 waves split the key tiles and merge at the end
    const int wid = blockIdx.x;
    const int qt0 = wid % qtiles;
    const int h = (wid / qtiles) % RH;
    const int b = wid / (qtiles * RH);
    const bf16_t* base = qkv + (size_t)b * N * (3 * RD) + h * HD;
    const int q0 = qt0 * 64;
    const uint32_t hseed = seed + (uint32_t)(b * RH + h) * 0x9E3779B9u;      // dropout stream of this (image, head)

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
    for (int kt0 = wave; kt0 < ktiles; kt0 += 4) {
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
                // scores in the log2 domain: s * (head_dim^-0.5 * log2 e), so the softmax needs v_exp_f32 only (no multiply
                // per element); keys beyond N exist only in the last key tile
                if (k0 + 64 <= N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[kt][qt][e] = s[e] * SCALE_LOG2E;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = k0 + 16 * kt + 4 * g + e;
                        st[kt][qt][e] = key < N ? s[e] * SCALE_LOG2E : -INFINITY;
                    }
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
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mx);        // 0 on the first tile (mrun = -inf)
            float sum = 0.f;
            f32x4 o = oacc[qt];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= alpha;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {        // two key tiles per x32 MFMA
                s16x4 pp[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kt = 2 * kp + hh;
                    float pv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pv[e] = __builtin_amdgcn_exp2f(st[kt][qt][e] - mx); sum += pv[e]; }
                    if constexpr (DROP) {       // nn.MultiheadAttention drops the normalised probabilities: l keeps the full sum
                        // four consecutive keys of one query row: two hashes (common.h drop_pair4; N is a multiple of 4, so is qi)
                        const uint32_t qi = (uint32_t)(q0 + 16 * qt + p) * (uint32_t)N + (uint32_t)(k0 + 16 * kt + 4 * g);
                        float dm[4];
                        drop_pair4(hseed, qi, thresh, inv_keep, dm);
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] *= dm[e];
                    }
                    pp[hh] = __builtin_bit_cast(s16x4, u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])});
                }
                o = mfma16x16x32(join4(vf[2 * kp], vf[2 * kp + 1]), join4(pp[0], pp[1]), o);
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
    // merge the four key splits: common max, rescale, then wave w finishes query sub-tile w
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
        if (g == 0) comb_m[wave][qt][p] = mrun[qt];
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const float M = fmaxf(fmaxf(comb_m[0][qt][p], comb_m[1][qt][p]), fmaxf(comb_m[2][qt][p], comb_m[3][qt][p]));
        const float sc = __builtin_amdgcn_exp2f(mrun[qt] - M);      // 0 for a wave that saw no key tile
        f32x4 o = oacc[qt];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= sc;
        *reinterpret_cast<f32x4*>(&comb_o[wave][qt][lane][0]) = o;
        if (g == 0) comb_l[wave][qt][p] = lrun[qt] * sc;
    }
    __syncthreads();
    {
        const int qt = wave;
        const int q = q0 + 16 * qt + p;
        if (q >= N) return;
        f32x4 o = *reinterpret_cast<const f32x4*>(&comb_o[0][qt][lane][0]);
        float l = comb_l[0][qt][p];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&comb_o[w][qt][lane][0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += t[e];
            l += comb_l[w][qt][p];
        }
        const float M = fmaxf(fmaxf(comb_m[0][qt][p], comb_m[1][qt][p]), fmaxf(comb_m[2][qt][p], comb_m[3][qt][p]));
        const float inv = 1.0f / l;
        if (lse && g == 0) lse[((size_t)b * RH + h) * N + q] = (M + __log2f(l)) * 0.69314718055994530942f;     // natural-log LSE, saved for the backward
        bf16_t* op = out + ((size_t)b * N + q) * RD + h * HD + 4 * g;
        *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv)};
    }
}

__global__ __launch_bounds__(256) void layernorm128_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int M)
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
    if (mean_out && sub == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
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

// ------------------------------------------------------------------------------------------------
// Attention backward (flash-style, P recomputed from q, k and the saved log-sum-exp), two passes so that no
// gradient needs cross-workgroup accumulation:
//   pass A, one wave per (b, h, 64-query tile), loops over key tiles:   dQ
//   pass B, one wave per (b, h, 64-key tile),   loops over query tiles: dK, dV
// Same register-orientation trick as the window kernel (attention_bwd.hip): the products whose contraction runs
// over keys use the T-layout tiles (rows = key), those over queries the N-layout tiles (rows = query), so every
// B operand is an accumulator tile converted in place.  D[q] = sum_d dO[q][d] O[q][d] comes from a tiny pre-pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rt_attn_bwd_prep_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ go,
                                                               float* __restrict__ dsum, int B, int N)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // over B*N*8
    if (idx >= (long long)B * N * RH) return;
    const int h = (int)(idx % RH);
    const long long row = idx / RH;                                            // b*N + q
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(o + row * RD + h * HD), a1 = *reinterpret_cast<const u32x4*>(o + row * RD + h * HD + 8);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(go + row * RD + h * HD), b1 = *reinterpret_cast<const u32x4*>(go + row * RD + h * HD + 8);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s += __builtin_bit_cast(float, a0[q] << 16) * __builtin_bit_cast(float, b0[q] << 16) +
             __builtin_bit_cast(float, a0[q] & 0xffff0000u) * __builtin_bit_cast(float, b0[q] & 0xffff0000u) +
             __builtin_bit_cast(float, a1[q] << 16) * __builtin_bit_cast(float, b1[q] << 16) +
             __builtin_bit_cast(float, a1[q] & 0xffff0000u) * __builtin_bit_cast(float, b1[q] & 0xffff0000u);
    }
    const int b = (int)(row / N), qi = (int)(row % N);
    dsum[((size_t)b * RH + h) * N + qi] = s;
}

TUP_DEVICE s16x4 f4_to_bf16x4(const f32x4 v) {
    const u32x2 p = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    return __builtin_bit_cast(s16x4, p);
}
TUP_DEVICE void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool DROP>
__global__ __launch_bounds__(256) void rt_attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ go, const float* __restrict__ lse,
    const float* __restrict__ dsum, bf16_t* __restrict__ gqkv, int B, int N, int tiles,
    uint32_t thresh, float inv_keep, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) bf16_t klds[4][64 * HD];
    __shared__ __attribute__((aligned(16))) float red[4][4][64][4];          // [wave][qt][lane][e]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    const int wid = blockIdx.x;                    // (image, head, query tile); the waves split the key tiles
    const int qt0 = wid % tiles, h = (wid / tiles) % RH, b = wid / (tiles * RH);
    const bf16_t* base = qkv + (size_t)b * N * (3 * RD) + h * HD;
    const bf16_t* gbase = go + (size_t)b * N * RD + h * HD;
    const int q0 = qt0 * 64;
    const uint32_t hseed = seed + (uint32_t)(b * RH + h) * 0x9E3779B9u;
    s16x4 qf[4], dof[4];
    float lc[4], dc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = min(q0 + 16 * t + p, N - 1);
        qf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)q * (3 * RD) + 4 * g);
        dof[t] = *reinterpret_cast<const s16x4*>(gbase + (size_t)q * RD + 4 * g);
        lc[t] = lse[((size_t)b * RH + h) * N + q] * 1.4426950408889634f;          // log2 domain: P = exp2(s * (0.25 log2 e) - lc), one FMA + one v_exp per score
        dc[t] = dsum[((size_t)b * RH + h) * N + q];
    }
    f32x4 dq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt0 = wave; kt0 < tiles; kt0 += 4) {
        const int k0 = kt0 * 64;
        {
            const int row = min(k0 + lane, N - 1);
            const bf16_t* kr = base + (size_t)row * (3 * RD) + RD;
            *reinterpret_cast<u32x4*>(&klds[wave][lane * HD]) = *reinterpret_cast<const u32x4*>(kr);
            *reinterpret_cast<u32x4*>(&klds[wave][lane * HD + 8]) = *reinterpret_cast<const u32x4*>(kr + 8);
        }
        s16x4 kf[4], vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = min(k0 + 16 * t + p, N - 1);
            kf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + RD + 4 * g);
            vf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + 2 * RD + 4 * g);
        }
        wave_sync_lds();
        s16x4 kT[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = klds[wave][(16 * kt + 4 * g + j) * HD + p];
            kT[kt] = __builtin_bit_cast(s16x4, t);
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {        // dQ contracts over keys: two key tiles per x32 MFMA
                s16x4 dsb[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kt = 2 * kp + hh;
                    const f32x4 s = mfma16x16x16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
                    const f32x4 dp = mfma16x16x16(vf[kt], dof[qt], f32x4{0.f, 0.f, 0.f, 0.f});
                    f32x4 ds;
                    float dm[4] = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (DROP) drop_pair4(hseed, (uint32_t)(q0 + 16 * qt + p) * (uint32_t)N + (uint32_t)(k0 + 16 * kt + 4 * g), thresh, inv_keep, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = k0 + 16 * kt + 4 * g + e;
                        const float pr = key < N ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], 0.25f * 1.4426950408889634f, -lc[qt])) : 0.f;
                        float dpe = dp[e];
                        if constexpr (DROP) dpe *= dm[e];
                        ds[e] = pr * (dpe - dc[qt]);
                    }
                    dsb[hh] = f4_to_bf16x4(ds);
                }
                dq[qt] = mfma16x16x32(join4(kT[2 * kp], kT[2 * kp + 1]), join4(dsb[0], dsb[1]), dq[qt]);
            }
        wave_sync_lds();
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) *reinterpret_cast<f32x4*>(&red[wave][qt][lane][0]) = dq[qt];
    __syncthreads();
    {
        const int qt = wave;
        const int q = q0 + 16 * qt + p;
        if (q >= N) return;
        f32x4 o = *reinterpret_cast<const f32x4*>(&red[0][qt][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&red[w][qt][lane][0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += t[e];
        }
        bf16_t* op = gqkv + ((size_t)b * N + q) * (3 * RD) + h * HD + 4 * g;
        *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(o[0] * 0.25f, o[1] * 0.25f), pack_bf16x2(o[2] * 0.25f, o[3] * 0.25f)};
    }
}

template <bool DROP>
__global__ __launch_bounds__(256) void rt_attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ go, const float* __restrict__ lse,
    const float* __restrict__ dsum, bf16_t* __restrict__ gqkv, int B, int N, int tiles,
    uint32_t thresh, float inv_keep, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) bf16_t lds[4][2][64 * HD];          // per wave: Q tile, dO tile
    __shared__ __attribute__((aligned(16))) float red[4][8][64][4];             // [wave][dV kt | dK kt][lane][e]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    const int wid = blockIdx.x;                    // (image, head, key tile); the waves split the query tiles
    const int kt0 = wid % tiles, h = (wid / tiles) % RH, b = wid / (tiles * RH);
    const bf16_t* base = qkv + (size_t)b * N * (3 * RD) + h * HD;
    const bf16_t* gbase = go + (size_t)b * N * RD + h * HD;
    const float* lrow = lse + ((size_t)b * RH + h) * N;
    const float* drow = dsum + ((size_t)b * RH + h) * N;
    const int k0 = kt0 * 64;
    const uint32_t hseed = seed + (uint32_t)(b * RH + h) * 0x9E3779B9u;
    s16x4 kf[4], vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = min(k0 + 16 * t + p, N - 1);
        kf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + RD + 4 * g);
        vf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)row * (3 * RD) + 2 * RD + 4 * g);
    }
    f32x4 dvT[4], dkT[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dvT[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dkT[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16_t* ql = lds[wave][0];
    bf16_t* dol = lds[wave][1];
    for (int qt0 = wave; qt0 < tiles; qt0 += 4) {
        const int q0 = qt0 * 64;
        {
            const int row = min(q0 + lane, N - 1);
            const bf16_t* qr = base + (size_t)row * (3 * RD);
            const bf16_t* gr = gbase + (size_t)row * RD;
            *reinterpret_cast<u32x4*>(ql + lane * HD) = *reinterpret_cast<const u32x4*>(qr);
            *reinterpret_cast<u32x4*>(ql + lane * HD + 8) = *reinterpret_cast<const u32x4*>(qr + 8);
            *reinterpret_cast<u32x4*>(dol + lane * HD) = *reinterpret_cast<const u32x4*>(gr);
            *reinterpret_cast<u32x4*>(dol + lane * HD + 8) = *reinterpret_cast<const u32x4*>(gr + 8);
        }
        s16x4 qf[4], dof[4];
        f32x4 lr[4], dr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = min(q0 + 16 * t + p, N - 1);
            qf[t] = *reinterpret_cast<const s16x4*>(base + (size_t)q * (3 * RD) + 4 * g);
            dof[t] = *reinterpret_cast<const s16x4*>(gbase + (size_t)q * RD + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qq = q0 + 16 * t + 4 * g + e;
                lr[t][e] = qq < N ? lrow[qq] * 1.4426950408889634f : 0.f;          // log2 domain, as in the dq kernel
                dr[t][e] = qq < N ? drow[qq] : 0.f;
            }
        }
        wave_sync_lds();
        s16x4 qT[4], doT[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x4 a, c;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = ql[(16 * t + 4 * g + j) * HD + p]; c[j] = dol[(16 * t + 4 * g + j) * HD + p]; }
            qT[t] = __builtin_bit_cast(s16x4, a); doT[t] = __builtin_bit_cast(s16x4, c);
        }
#pragma unroll
        for (int qp = 0; qp < 2; ++qp)              // dV / dK contract over queries: two query tiles per x32 MFMA
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s16x4 prb[2], dsb[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int qt = 2 * qp + hh;
                    const f32x4 s = mfma16x16x16(qf[qt], kf[kt], f32x4{0.f, 0.f, 0.f, 0.f});       // rows query 4g+e, cols key p
                    const f32x4 dp = mfma16x16x16(dof[qt], vf[kt], f32x4{0.f, 0.f, 0.f, 0.f});
                    f32x4 pr, ds;
                    float dm[4] = {1.f, 1.f, 1.f, 1.f};
                    // queries 4g .. 4g+3 of one key per lane; the key's pair partner is the neighbouring lane.  (Measured, backward of a block at
                    // B = 2: this form 380 us, one pair hash per element 428 us, round 2's one plain hash per element 382 us -- in THIS
                    // kernel the halved hash count only pays for the exchange; forward and dQ, whose lanes hold the pair, gained 20 / 39 us.)
                    if constexpr (DROP)
                        drop_pair4_rows(hseed, (uint32_t)(q0 + 16 * qt + 4 * g) * (uint32_t)N + (uint32_t)(k0 + 16 * kt + p), (uint32_t)N, thresh, inv_keep, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qq = q0 + 16 * qt + 4 * g + e;
                                                pr[e] = qq < N ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], 0.25f * 1.4426950408889634f, -lr[qt][e])) : 0.f;      // (vs __expf: 378 vs 383 us per backward)
                        ds[e] = pr[e] * (dp[e] * dm[e] - dr[qt][e]);
                        pr[e] *= dm[e];                      // dV sees the dropped probabilities
                    }
                    prb[hh] = f4_to_bf16x4(pr);
                    dsb[hh] = f4_to_bf16x4(ds);
                }
                dvT[kt] = mfma16x16x32(join4(doT[2 * qp], doT[2 * qp + 1]), join4(prb[0], prb[1]), dvT[kt]);
                dkT[kt] = mfma16x16x32(join4(qT[2 * qp], qT[2 * qp + 1]), join4(dsb[0], dsb[1]), dkT[kt]);
            }
        wave_sync_lds();
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        *reinterpret_cast<f32x4*>(&red[wave][kt][lane][0]) = dvT[kt];
        *reinterpret_cast<f32x4*>(&red[wave][4 + kt][lane][0]) = dkT[kt];
    }
    __syncthreads();
    {
        const int kt = wave;
        const int key = k0 + 16 * kt + p;
        if (key >= N) return;
        f32x4 dv = *reinterpret_cast<const f32x4*>(&red[0][kt][lane][0]);
        f32x4 dk = *reinterpret_cast<const f32x4*>(&red[0][4 + kt][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&red[w][kt][lane][0]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&red[w][4 + kt][lane][0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dv[e] += a[e]; dk[e] += c[e]; }
        }
        bf16_t* op = gqkv + ((size_t)b * N + key) * (3 * RD) + h * HD + 4 * g;
        *reinterpret_cast<u32x2*>(op + RD) = u32x2{pack_bf16x2(dk[0] * 0.25f, dk[1] * 0.25f), pack_bf16x2(dk[2] * 0.25f, dk[3] * 0.25f)};
        *reinterpret_cast<u32x2*>(op + 2 * RD) = u32x2{pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3])};
    }
}

// Separable form for large upscaling ratios (>= 3, e.g. the 6x configuration): a 16 x 64 output tile touches at most 16 source
// rows of either source, so the horizontal interpolation is done once per (source row, output column) into LDS and the
// vertical one reads it back -- ~6 FMAs and LDS reads per output instead of 32 FMAs + 48 global loads.  Same fmaf chains
// as the direct kernel, so the results are identical.
constexpr int BIC_TR = 16, BIC_MAXROWS = 16;           // BIC_TR: smallest tile height (rows); the launcher passes the largest that fits
__global__ __launch_bounds__(256) void rt_bicubic_sum_sep_kernel(
    const float* __restrict__ a, const float* __restrict__ bsrc, float* __restrict__ out,
    const int* __restrict__ ayi, const float* __restrict__ ayw, const int* __restrict__ axi, const float* __restrict__ axw,
    const int* __restrict__ byi, const float* __restrict__ byw, const int* __restrict__ bxi, const float* __restrict__ bxw,
    int Ha, int Wa, int Hb, int Wb, int Ho, int Wo, int clamp01, int tr)
{
    // one thread = one output column for all `tr` rows of the tile; the workgroup's 256 columns are adjacent, so its four waves
    // store 1 KB of a row together (64-column tiles wrote 256-byte pieces 30 KB apart: 1.4 TB/s).  A thread reads back only its own
    // column of the horizontally interpolated rows -- LDS serves as per-thread indexed storage, no barrier.
    __shared__ float hbuf[2][BIC_MAXROWS][256];
    const int col = threadIdx.x;
    const int ox = blockIdx.x * 256 + col, oxc = min(ox, Wo - 1);
    const int oy0 = blockIdx.y * tr, oy1 = min(oy0 + tr, Ho) - 1;
    const int plane = blockIdx.z;
    const float* pa = a + (size_t)plane * Ha * Wa;
    const float* pb = bsrc + (size_t)plane * Hb * Wb;
    // source row ranges of the tile (tap tables are monotone, indices clamped)
    const int ya0 = ayi[oy0 * 4], ya1 = ayi[oy1 * 4 + 3], yb0 = byi[oy0 * 4], yb1 = byi[oy1 * 4 + 3];
    int xa[4], xb[4];
    float wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xa[j] = axi[oxc * 4 + j]; wa[j] = axw[oxc * 4 + j];
        xb[j] = bxi[oxc * 4 + j]; wb[j] = bxw[oxc * 4 + j];
    }
    for (int r = 0; r <= ya1 - ya0; ++r) {
        const float* ra = pa + (size_t)(ya0 + r) * Wa;
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) h = fmaf(wa[j], ra[xa[j]], h);
        hbuf[0][r][col] = h;
    }
    for (int r = 0; r <= yb1 - yb0; ++r) {
        const float* rb = pb + (size_t)(yb0 + r) * Wb;
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) h = fmaf(wb[j], rb[xb[j]], h);
        hbuf[1][r][col] = h;
    }
    if (ox >= Wo) return;
    for (int oy = oy0; oy <= oy1; ++oy) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = fmaf(ayw[oy * 4 + i], hbuf[0][ayi[oy * 4 + i] - ya0][col], acc);
            acc = fmaf(byw[oy * 4 + i], hbuf[1][byi[oy * 4 + i] - yb0][col], acc);
        }
        if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
        out[((size_t)plane * Ho + oy) * Wo + ox] = acc;
    }
}

// Backward of out = clamp(bicubic(a) + bicubic(b)) w.r.t. a, separable, in gather form: the clamp gate is read from
// the saved output (0 < out < 1), pass 1 reduces output rows onto source rows (CSR lists per source row: which
// output rows touch it and with what weight), pass 2 does the same along x.
__global__ __launch_bounds__(256) void rt_bicubic_bwd_rows_kernel(
    const float* __restrict__ gout, const float* __restrict__ out, float* __restrict__ tmp,
    const int* __restrict__ ystart, const int* __restrict__ yo, const float* __restrict__ yw, int Ha, int Ho, int Wo)
{
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, plane = blockIdx.z;
    if (ox >= Wo) return;
    const int t0 = ystart[y], t1 = ystart[y + 1];
    const float* gp = gout + (size_t)plane * Ho * Wo + ox;
    const float* op = out ? out + (size_t)plane * Ho * Wo + ox : nullptr;
    float acc = 0.f;
    // four list entries per trip, every load of the trip requested before the first use (index clamped, weight zeroed past the
    // end): the rolled loop paid one global round trip per entry (24 per output at x6)
    for (int t = t0; t < t1; t += 4) {
        float w[4], gv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = min(t + u, t1 - 1);
            w[u] = t + u < t1 ? yw[tt] : 0.f;
            const size_t off = (size_t)yo[tt] * Wo;
            gv[u] = gp[off];
            ov[u] = op ? op[off] : 0.5f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += w[u] * ((ov[u] > 0.f && ov[u] < 1.f) ? gv[u] : 0.f);
    }
    tmp[((size_t)plane * Ha + y) * Wo + ox] = acc;
}

// Row pass in bands of BIC_YB source rows: the output rows a band touches are one contiguous range [r0, r0 + n), every one of
// them is read ONCE (gradient + clamp gate) and feeds the band's BIC_YB accumulators through a dense, wave-uniform weight row
// (scalar loads).  The per-source-row gather above reads every output row four times at x6 (6.4 GB of requests for 1.6 GB).
// L1 = true: the upstream gradient is nn.L1Loss's, formed on the fly -- gout is the TARGET of the loss and the gradient of output
// element o is sign(o - target) * l1_scale[0] (l1_scale = grad_of_loss / numel, on the device): the 796 MB gradient tensor of the
// x6 configuration is never written or read (l1_bwd_kernel: 0.5 ms per step).
constexpr int BIC_YB = 16;
template <bool L1>
__global__ __launch_bounds__(256) void rt_bicubic_bwd_rows_band_kernel(
    const float* __restrict__ gout, const float* __restrict__ out, float* __restrict__ tmp,
    const int* __restrict__ band_r0, const int* __restrict__ band_n, const float* __restrict__ band_w, int nr_max,
    int Ha, int Ho, int Wo, const float* __restrict__ l1_scale)
{
    const float gs = L1 ? l1_scale[0] : 0.f;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int band = blockIdx.y, plane = blockIdx.z;
    if (ox >= Wo) return;
    const int r0 = band_r0[band], n = band_n[band];
    const float* __restrict__ wrow = band_w + (size_t)band * nr_max * BIC_YB;
    const float* gp = gout + ((size_t)plane * Ho + r0) * Wo + ox;
    const float* op = out ? out + ((size_t)plane * Ho + r0) * Wo + ox : nullptr;
    float acc[BIC_YB];
#pragma unroll
    for (int k = 0; k < BIC_YB; ++k) acc[k] = 0.f;
    for (int i = 0; i < n; i += 4) {
        float gv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t off = (size_t)min(i + u, n - 1) * Wo;
            gv[u] = gp[off];
            ov[u] = op ? op[off] : 0.5f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u >= n) break;                           // wave-uniform
            float g = gv[u];
            if constexpr (L1) { const float d = ov[u] - gv[u]; g = d > 0.f ? gs : (d < 0.f ? -gs : 0.f); }
            g = (ov[u] > 0.f && ov[u] < 1.f) ? g : 0.f;
#pragma unroll
            for (int k = 0; k < BIC_YB; ++k) acc[k] = fmaf(wrow[(size_t)(i + u) * BIC_YB + k], g, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < BIC_YB; ++k) {
        const int y = band * BIC_YB + k;
        if (y < Ha) tmp[((size_t)plane * Ha + y) * Wo + ox] = acc[k];
    }
}

// Column pass: 256 adjacent source columns per workgroup; the stretch of the tmp row their lists touch (a contiguous range, the
// lists are monotone) is staged in LDS with coalesced loads, the per-column gathers then read LDS.  (Gathering straight from
// global: 24 scattered 4-byte loads per output, 519 us for 22 MB of output.)
constexpr int BIC_CSPAN = 4096;            // floats of LDS for the row stretch; wider stretches (ratios > ~15) use the direct form
__global__ __launch_bounds__(256) void rt_bicubic_bwd_cols_kernel(
    const float* __restrict__ tmp, float* __restrict__ ga, const int* __restrict__ xstart, const int* __restrict__ xo,
    const float* __restrict__ xw, int Ha, int Wa, int Wo)
{
    __shared__ float seg[BIC_CSPAN];
    const int x0 = blockIdx.x * 256, x = x0 + threadIdx.x;
    const int y = blockIdx.y, plane = blockIdx.z;
    const float* tp = tmp + ((size_t)plane * Ha + y) * Wo;
    const int xl = min(x0 + 255, Wa - 1);
    const int c0 = xo[xstart[x0]], c1 = xo[xstart[xl + 1] - 1];          // first / last output column of the workgroup's lists
    const bool staged = c1 - c0 < BIC_CSPAN;
    if (staged) {
        for (int i = threadIdx.x; i <= c1 - c0; i += 256) seg[i] = tp[c0 + i];
        __syncthreads();
    }
    if (x >= Wa) return;
    float acc = 0.f;
    const int t0 = xstart[x], t1 = xstart[x + 1];
    for (int t = t0; t < t1; t += 8) {            // eight entries per trip, loads batched as in the row pass
        float w[8], v[8];
        int idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = min(t + u, t1 - 1);
            w[u] = t + u < t1 ? xw[tt] : 0.f;
            idx[u] = xo[tt];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = staged ? seg[idx[u] - c0] : tp[idx[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[u] * v[u];
    }
    ga[((size_t)plane * Ha + y) * Wa + x] = acc;
}

// Column pass with dense per-column tap tables: xoT / xwT [kmax][Wa] (entry k of source column x; padded with weight 0), so a
// thread's kmax index / weight loads are independent and coalesced across the workgroup (the CSR form above chains
// xstart -> xo -> value: ~7 dependent round trips per workgroup); blk_c0 / blk_n = the stretch of the tmp row a workgroup stages.
__global__ __launch_bounds__(256) void rt_bicubic_bwd_cols_dense_kernel(
    const float* __restrict__ tmp, float* __restrict__ ga, const int* __restrict__ xoT, const float* __restrict__ xwT, int kmax,
    const int* __restrict__ blk_c0, const int* __restrict__ blk_n, int Ha, int Wa, int Wo)
{
    __shared__ float seg[BIC_CSPAN];
    const int x = blockIdx.x * 256 + threadIdx.x, xc = min(x, Wa - 1);
    const int y = blockIdx.y, plane = blockIdx.z;
    const float* tp = tmp + ((size_t)plane * Ha + y) * Wo;
    const int c0 = blk_c0[blockIdx.x], n = blk_n[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) seg[i] = tp[c0 + i];
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < kmax; k += 8) {
        float w[8];
        int idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = min(k + u, kmax - 1);
            w[u] = k + u < kmax ? xwT[(size_t)kk * Wa + xc] : 0.f;
            idx[u] = xoT[(size_t)kk * Wa + xc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[u] * seg[idx[u] - c0];
    }
    if (x < Wa) ga[((size_t)plane * Ha + y) * Wa + x] = acc;
}

}  // namespace

// qkv bf16 [B][N][384] (q | k | v, each 8 heads x 16); out bf16 [B][N][128]; lse fp32 [B][8][N] (optional: log-sum-exp
// of the scaled scores per query, saved for the backward).  Eval-mode attention (no dropout).
extern "C" int tup_rt_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, float drop_p,
                                    unsigned int drop_seed, void* stream)
{
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && N % 4 != 0)) return (int)hipErrorInvalidValue;      // the mask pairs keys: rows of 4 | N
    uint32_t thresh; float inv_keep;
    drop_pair_params(drop_p, thresh, inv_keep);
    if (B <= 0 || N <= 0) return 0;
    const int qtiles = (N + 63) / 64;
    const long long blocks = (long long)B * RH * qtiles;
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    if (drop_p > 0.f)
        rt_attention_kernel<true><<<grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            (const bf16_t*)qkv, (bf16_t*)out, lse, B, N, qtiles, thresh, inv_keep, drop_seed);
    else
        rt_attention_kernel<false><<<grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
            (const bf16_t*)qkv, (bf16_t*)out, lse, B, N, qtiles, 0u, 1.0f, 0u);
    TUP_CHECK_LAUNCH();
    return 0;
}

extern "C" int tup_layernorm128_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    int M, void* stream)
{
    if (M <= 0) return 0;
    if ((mean == nullptr) != (rstd == nullptr)) return (int)hipErrorInvalidValue;
    layernorm128_kernel<<<dim3((M + 15) / 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(x, gamma, beta, (bf16_t*)y, mean, rstd, M);
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
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 16 output rows span at most 16/ratio + 4 source rows (the separable kernel's LDS tile holds 16).  Measured: at
    // ratio 6 / 12 (720p -> 8K) it is 1.9x faster than the direct kernel, at 1.5 / 3 (720p -> 1080p) 3x slower.
    if (Ho >= 3 * Ha && Ho >= 3 * Hb && (Ho + BIC_TR - 1) / BIC_TR <= 65535) {
        // tile height: as many output rows as keep the source-row span (rows / ratio + 4 taps, + 1 for the phase) inside the LDS
        // tile; at x6 that is 64 rows instead of 16 -- 194 k workgroups of ~4 outputs per thread were launch-bound (797 us for
        // 796 MB written)
        const int hmax = Ha > Hb ? Ha : Hb;
        int tr = BIC_TR;
        while (tr + 4 <= 64 && (long long)(tr + 4) * hmax / Ho + 5 <= BIC_MAXROWS) tr += 4;
        dim3 grid((Wo + 255) / 256, (Ho + tr - 1) / tr, planes);
        rt_bicubic_sum_sep_kernel<<<grid, dim3(256), 0, s>>>(a, b, out, ayi, ayw, axi, axw, byi, byw, bxi, bxw, Ha, Wa, Hb, Wb, Ho, Wo, clamp01, tr);
        TUP_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, planes);
    rt_bicubic_sum_kernel<<<grid, dim3(256), 0, s>>>(
        a, b, out, ayi, ayw, axi, axw, byi, byw, bxi, bxw, Ha, Wa, Hb, Wb, Ho, Wo, clamp01);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Backward of tup_rt_attention_fwd: qkv bf16 [B][N][384], out (forward output) and gout bf16 [B][N][128], lse fp32
// [B][8][N]; work fp32 [B][8][N] scratch; gqkv bf16 [B][N][384] (overwritten).
extern "C" int tup_rt_attention_bwd(const void* qkv, const void* out, const void* gout, const float* lse, float* work,
                                    void* gqkv, int B, int N, float drop_p, unsigned int drop_seed, void* stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && N % 4 != 0)) return (int)hipErrorInvalidValue;
    uint32_t thresh; float inv_keep;
    drop_pair_params(drop_p, thresh, inv_keep);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long nprep = (long long)B * N * RH;
    rt_attn_bwd_prep_kernel<<<dim3((unsigned)((nprep + 255) / 256)), dim3(256), 0, s>>>((const bf16_t*)out, (const bf16_t*)gout, work, B, N);
    TUP_CHECK_LAUNCH();
    const int tiles = (N + 63) / 64;
    const long long blocks = (long long)B * RH * tiles;
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const unsigned grid = (unsigned)blocks;
    if (drop_p > 0.f) {
        rt_attn_bwd_dq_kernel<true><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)qkv, (const bf16_t*)gout, lse, work, (bf16_t*)gqkv, B, N, tiles, thresh, inv_keep, drop_seed);
        TUP_CHECK_LAUNCH();
        rt_attn_bwd_dkv_kernel<true><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)qkv, (const bf16_t*)gout, lse, work, (bf16_t*)gqkv, B, N, tiles, thresh, inv_keep, drop_seed);
        TUP_CHECK_LAUNCH();
    } else {
        rt_attn_bwd_dq_kernel<false><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)qkv, (const bf16_t*)gout, lse, work, (bf16_t*)gqkv, B, N, tiles, 0u, 1.0f, 0u);
        TUP_CHECK_LAUNCH();
        rt_attn_bwd_dkv_kernel<false><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)qkv, (const bf16_t*)gout, lse, work, (bf16_t*)gqkv, B, N, tiles, 0u, 1.0f, 0u);
        TUP_CHECK_LAUNCH();
    }
    return 0;
}

// Backward of tup_rt_bicubic_sum_fwd w.r.t. its first source `a`: gout / out fp32 [planes][Ho][Wo] (out = the saved
// forward output; NULL = no clamp gate), ga fp32 [planes][Ha][Wa], tmp fp32 [planes][Ha][Wo] scratch.  ystart [Ha+1] /
// yo / yw and xstart [Wa+1] / xo / xw are the transposed tap lists (per source row / column: the outputs it feeds).
extern "C" int tup_rt_bicubic_bwd(const float* gout, const float* out, float* ga, float* tmp, const int* ystart, const int* yo,
                                  const float* yw, const int* xstart, const int* xo, const float* xw, int planes, int Ha, int Wa,
                                  int Ho, int Wo, void* stream)
{
    if (planes <= 0) return 0;
    if (planes > 65535 || Ha > 65535) return (int)hipErrorInvalidValue;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rt_bicubic_bwd_rows_kernel<<<dim3((Wo + 255) / 256, Ha, planes), dim3(256), 0, s>>>(gout, out, tmp, ystart, yo, yw, Ha, Ho, Wo);
    TUP_CHECK_LAUNCH();
    rt_bicubic_bwd_cols_kernel<<<dim3((Wa + 255) / 256, Ha, planes), dim3(256), 0, s>>>(tmp, ga, xstart, xo, xw, Ha, Wa, Wo);
    TUP_CHECK_LAUNCH();
    return 0;
}

// tup_rt_bicubic_bwd with the row pass in bands of 16 source rows (rt_bicubic_bwd_rows_band_kernel) and dense column tables:
// band_r0 / band_n int [nbands] = first output row and row count of band b (source rows 16b .. 16b+15), band_w fp32
// [nbands][nr_max][16] = weight of output row r0 + i for source row 16b + k (zero where none; rows i >= n are not read);
// xoT int / xwT fp32 [kmax][Wa] = entry k of source column x's transposed tap list (padded: weight 0, any in-range index),
// blk_c0 / blk_n int [ceil(Wa/256)] = first tmp column and count (<= 4096) the lists of source columns 256j .. 256j+255 touch.
// Same results up to the order of the fp32 sums.  l1_scale != NULL: `gout` is the TARGET of nn.L1Loss on the forward output and
// the upstream gradient is sign(out - target) * l1_scale[0] (device scalar = d loss / numel), never materialised.
extern "C" int tup_rt_bicubic_bwd_banded(const float* gout, const float* out, float* ga, float* tmp, const int* band_r0,
                                         const int* band_n, const float* band_w, int nr_max, const int* xoT, const float* xwT,
                                         int kmax, const int* blk_c0, const int* blk_n, int planes, int Ha, int Wa, int Ho, int Wo,
                                         const float* l1_scale, void* stream)
{
    if (planes <= 0) return 0;
    const int nbands = (Ha + BIC_YB - 1) / BIC_YB;
    if (planes > 65535 || Ha > 65535 || nbands > 65535 || nr_max < 1 || kmax < 1) return (int)hipErrorInvalidValue;
    if (l1_scale != nullptr && out == nullptr) return (int)hipErrorInvalidValue;      // the L1 form needs the forward output
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (l1_scale != nullptr)
        rt_bicubic_bwd_rows_band_kernel<true><<<dim3((Wo + 255) / 256, nbands, planes), dim3(256), 0, s>>>(
            gout, out, tmp, band_r0, band_n, band_w, nr_max, Ha, Ho, Wo, l1_scale);
    else
    rt_bicubic_bwd_rows_band_kernel<false><<<dim3((Wo + 255) / 256, nbands, planes), dim3(256), 0, s>>>(
        gout, out, tmp, band_r0, band_n, band_w, nr_max, Ha, Ho, Wo, nullptr);
    TUP_CHECK_LAUNCH();
    rt_bicubic_bwd_cols_dense_kernel<<<dim3((Wa + 255) / 256, Ha, planes), dim3(256), 0, s>>>(tmp, ga, xoT, xwT, kmax, blk_c0, blk_n, Ha, Wa, Wo);
    TUP_CHECK_LAUNCH();
    return 0;
}

