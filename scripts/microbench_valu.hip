// Issue-cost microbenchmark for the SIMD of gfx950 (MI355X): what one vector instruction costs a wave's stream, alone and
// beside v_mfma_f32_16x16x32_bf16, at one and two waves per SIMD.  Feeds the instruction accounting of the whole-block kernel
// (DESIGN 5c): is v_pk_fma_f32 one issue or two, is v_pk_fma_f16 full rate, how many VALU slots does an MFMA stream leave.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_valu.hip -o /tmp/mb_valu && /tmp/mb_valu
//
// Every case: 256 workgroups (one per CU), ITER iterations of an unrolled body written in inline asm (16 independent chains),
// s_memtime around the loop, median over workgroups of cycles per body instruction.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

constexpr int ITER = 4000;

enum { FMA32 = 0, PKFMA32, PKFMA16, PKMUL32, EXP32, CVTBF16, MED3, NOPS };

template <int OP>
__device__ __forceinline__ void body16(float (&a)[16], f32x2 (&p)[16], h2 (&h)[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
        if constexpr (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
        if constexpr (OP == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
        if constexpr (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
        if constexpr (OP == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if constexpr (OP == CVTBF16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
        if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
    }
}

// MODE 0: every wave runs the VALU body.  MODE 1: every wave runs NM MFMAs + NV body instructions interleaved per step.
// MODE 2 (two waves per SIMD): waves 0-3 run MFMAs only, waves 4-7 the VALU body only.
template <int OP, int MODE, int NV>
__global__ void k(unsigned long long* out, float seed)
{
    float a[16]; f32x2 p[16]; h2 h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1); p[i] = f32x2{seed * i, seed}; h[i] = h2{(_Float16)(seed * i), (_Float16)seed}; }
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed * i); fb[i] = (__bf16)(seed + i); }
    f32x4 acc[4] = {};
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = MODE == 1 || (MODE == 2 && wave < 4);
    const bool valu_role = MODE == 0 || MODE == 1 || (MODE == 2 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
            body16<OP>(a, p, h);
        } else if (MODE == 1) {
            // 4 MFMAs, each followed by NV VALU instructions (16 chains rotate)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fa), "v"(fb));
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int i = (m * NV + v) & 15;
                    if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                    if constexpr (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
                    if constexpr (OP == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
                    if constexpr (OP == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
            }
        } else {
            if (mfma_role) {
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(fa), "v"(fb));
            } else {
                body16<OP>(a, p, h);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1] + (float)h[i][0] + (float)h[i][1];
#pragma unroll
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (s == 123.456f) out[0] = 0;         // keep everything live
    (void)valu_role;
}

template <int OP, int MODE, int NV>
static void run(const char* name, int threads, unsigned long long* dbuf)
{
    hipMemset(dbuf, 0, 256 * 16 * 8);
    for (int r = 0; r < 3; ++r) k<OP, MODE, NV><<<256, threads>>>(dbuf, 1e-6f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    auto med = [&](int w0, int w1) {
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) for (int w = w0; w < w1; ++w) v.push_back((double)h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        return v[v.size() / 2] / ITER;
    };
    if (MODE == 0) printf("%-46s waves/SIMD %d: %.2f cycles per instruction\n", name, nw / 4, med(0, nw) / 16);
    if (MODE == 1) printf("%-46s waves/SIMD %d: %.2f cycles per (MFMA + %d VALU)\n", name, nw / 4, med(0, nw) / 4, NV);
    if (MODE == 2) printf("%-46s MFMA waves %.2f cycles per MFMA | VALU waves %.2f cycles per instruction\n", name, med(0, 4) / 16, med(4, 8) / 16);
}

int main()
{
    unsigned long long* dbuf;
    hipMalloc(&dbuf, 256 * 16 * 8);
    run<FMA32, 0, 0>("v_fma_f32 alone", 256, dbuf);
    run<FMA32, 0, 0>("v_fma_f32 alone", 512, dbuf);
    run<PKFMA32, 0, 0>("v_pk_fma_f32 alone", 256, dbuf);
    run<PKFMA32, 0, 0>("v_pk_fma_f32 alone", 512, dbuf);
    run<PKFMA16, 0, 0>("v_pk_fma_f16 alone", 256, dbuf);
    run<PKFMA16, 0, 0>("v_pk_fma_f16 alone", 512, dbuf);
    run<PKMUL32, 0, 0>("v_pk_mul_f32 alone", 256, dbuf);
    run<EXP32, 0, 0>("v_exp_f32 alone", 256, dbuf);
    run<EXP32, 0, 0>("v_exp_f32 alone", 512, dbuf);
    run<CVTBF16, 0, 0>("v_cvt_pk_bf16_f32 alone", 256, dbuf);
    run<MED3, 0, 0>("v_med3_f32 alone", 256, dbuf);
    run<FMA32, 1, 0>("MFMA 16x16x32 alone", 256, dbuf);
    run<FMA32, 1, 0>("MFMA 16x16x32 alone", 512, dbuf);
    run<FMA32, 1, 1>("MFMA + v_fma_f32", 256, dbuf);
    run<FMA32, 1, 2>("MFMA + v_fma_f32", 256, dbuf);
    run<FMA32, 1, 3>("MFMA + v_fma_f32", 256, dbuf);
    run<FMA32, 1, 4>("MFMA + v_fma_f32", 256, dbuf);
    run<FMA32, 1, 2>("MFMA + v_fma_f32", 512, dbuf);
    run<FMA32, 1, 4>("MFMA + v_fma_f32", 512, dbuf);
    run<PKFMA32, 1, 1>("MFMA + v_pk_fma_f32", 256, dbuf);
    run<PKFMA32, 1, 2>("MFMA + v_pk_fma_f32", 256, dbuf);
    run<PKFMA32, 1, 2>("MFMA + v_pk_fma_f32", 512, dbuf);
    run<PKFMA16, 1, 1>("MFMA + v_pk_fma_f16", 256, dbuf);
    run<PKFMA16, 1, 2>("MFMA + v_pk_fma_f16", 256, dbuf);
    run<PKFMA16, 1, 4>("MFMA + v_pk_fma_f16", 256, dbuf);
    run<PKFMA16, 1, 2>("MFMA + v_pk_fma_f16", 512, dbuf);
    run<PKFMA16, 1, 4>("MFMA + v_pk_fma_f16", 512, dbuf);
    run<EXP32, 1, 1>("MFMA + v_exp_f32", 256, dbuf);
    run<FMA32, 2, 0>("role split: MFMA waves | v_fma_f32 waves", 512, dbuf);
    run<PKFMA32, 2, 0>("role split: MFMA waves | v_pk_fma_f32 waves", 512, dbuf);
    run<PKFMA16, 2, 0>("role split: MFMA waves | v_pk_fma_f16 waves", 512, dbuf);
    run<EXP32, 2, 0>("role split: MFMA waves | v_exp_f32 waves", 512, dbuf);
    hipFree(dbuf);
    return 0;
}
