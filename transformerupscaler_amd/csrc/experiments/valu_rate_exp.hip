// EXPERIMENT (`make exp`): issue cost of vector instructions on one SIMD of gfx950, alone and beside v_mfma_f32_32x32x16_bf16, with one
// and two waves per SIMD.  Each test: REP x (one MFMA (optional) + NF independent filler instructions), s_memtime around the loop.
#include "../common.h"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2v;
namespace {
template <int KIND, int CNT = 8> __device__ __forceinline__ void filler(float (&a)[8], float b, float c, uint32_t& sc, u32x4 (&q)[4], uint32_t lds_a, f32x2v (&d2)[4], f32x2v e2) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 2) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 3) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        else if constexpr (KIND == 6) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 10) asm volatile("s_nop 0");
        else if constexpr (KIND == 11) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) :: "scc");
        else if constexpr (KIND == 12) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(lds_a));
        else if constexpr (KIND == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
        else if constexpr (KIND == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 15) asm volatile("v_pk_fmac_f16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else if constexpr (KIND == 16) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else if constexpr (KIND == 17) asm volatile("s_waitcnt lgkmcnt(15)");
        else if constexpr (KIND == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d2[i & 3]) : "v"(e2));
        else if constexpr (KIND == 19) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d2[i & 3]) : "v"(e2));
        else if constexpr (KIND == 20) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(c));
    }
}
template <int KIND, int NF8, bool MF, int CNT = 8> __global__ __launch_bounds__(512) void rate_kernel(unsigned long long* out, float seed, int reps)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    f32x16 acc = {};
    __shared__ __attribute__((aligned(16))) char lds[65536];
    uint32_t sc = 0; u32x4 q[4] = {};
    f32x2v d2[4] = {{seed, seed}, {seed, 1.f}, {2.f, seed}, {seed, 3.f}}, e2 = {seed, seed};
    const uint32_t lds_a = lds_addr(lds) + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    bf16x8 fa = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), fb = fa;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (MF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
#pragma unroll
            for (int k = 0; k < NF8; ++k) filler<KIND, CNT>(a, seed, seed, sc, q, lds_a, d2, e2);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = d2[0][0] + d2[1][1] + d2[2][0] + d2[3][1] + (float)sc + __builtin_bit_cast(float, q[0][0] ^ q[1][1] ^ q[2][2] ^ q[3][3]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}
}  // namespace

// kind: filler instruction; nf8: fillers per MFMA gap / 8; mf: with MFMAs; waves: 4 (one per SIMD) or 8; out: [8] cycles of each wave
extern "C" int tup_exp_valu_rate(unsigned long long* out, int kind, int nf8, int mf, int waves, int reps, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define L(K, N, M) if (kind == K && nf8 == N && mf == M) { rate_kernel<K, N, (M != 0)><<<dim3(256), dim3(64 * waves), 0, st>>>(out, 1.0f, reps); return (int)hipGetLastError(); }
#define L2(K, C) if (kind == K && nf8 == 100 + C && mf == 1) { rate_kernel<K, 1, true, C><<<dim3(256), dim3(64 * waves), 0, st>>>(out, 1.0f, reps); return (int)hipGetLastError(); }
#define LK(K) L(K, 1, 0) L(K, 1, 1) L(K, 0, 1) L(K, 2, 1) L2(K, 2) L2(K, 4) L2(K, 6)
    LK(0) LK(1) LK(2) LK(3) LK(4) LK(5) LK(6) LK(7) LK(8) LK(9) LK(10) LK(11) LK(12) LK(13) LK(14) LK(15) LK(16) LK(17) LK(18) LK(19) LK(20)
    return 1;
}
