// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the FastTransformer path.
// Wave = 64 lanes; MFMA fragments follow the gfx950 maps:
//   v_mfma_f32_16x16x32_bf16: A[row = l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col = l&15],
//                             C/D[row = 4*(l>>4)+reg][col = l&15]
//   v_mfma_f32_16x16x16_bf16: A[row = l&15][k = 4*(l>>4)+j], B[k = 4*(l>>4)+j][col = l&15], same C/D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TUP_DEVICE __device__ __forceinline__

// ---- LDS reads the compiler does not track (hand-pipelined MFMA loops) ----
// hipcc sinks every ds_read to just above its first use and waits lgkmcnt(0) there; at one or two waves per
// SIMD that exposes the LDS latency in front of every few MFMAs.  These helpers issue ds_read_b128 from
// inline asm (invisible to the waitcnt pass) so the caller can request the next K-step's fragments early and
// wait with a hand-counted lgkmcnt(N); follow the wait with __builtin_amdgcn_sched_barrier(0) so the MFMAs
// stay below it (guide 5.4 rule 18).  No compiler-visible LDS access may be in flight across such a section.
TUP_DEVICE uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(p);
}
TUP_DEVICE bf16x8 lds_read_b128_asm(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// Same with the instruction's 16-bit immediate offset: unrolled loops then share a few base registers instead of
// one VGPR per (loop-invariant, hence hoisted) address.  `off` must fold to a constant < 65536.
TUP_DEVICE bf16x8 lds_read_b128_asm_off(uint32_t addr, int off) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off));
    return v;
}
// ... and with "^ 64" (the kh = 1 half of a 128-B row) applied inside the asm, so that the flipped address is a
// 1-instruction temporary instead of a second hoisted register per table entry.
TUP_DEVICE bf16x8 lds_read_b128_asm_off_x64(uint32_t addr, int off) {
    bf16x8 v;
    uint32_t tmp;
    asm volatile("v_xor_b32 %1, 64, %2\n\tds_read_b128 %0, %1 offset:%3" : "=v"(v), "=&v"(tmp) : "v"(addr), "i"(off));
    return v;
}
template <int N> TUP_DEVICE void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

TUP_DEVICE float bf16_to_f32(bf16_t v) { return static_cast<float>(v); }
TUP_DEVICE bf16_t f32_to_bf16(float v) { return static_cast<bf16_t>(v); }   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

TUP_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 p = {static_cast<bf16_t>(lo), static_cast<bf16_t>(hi)};
    return __builtin_bit_cast(uint32_t, p);
}

TUP_DEVICE f32x4 mfma16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
TUP_DEVICE f32x4 mfma16x16x16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// LDS image of a [rows][64 bf16] tile (128-byte rows, eight 16-byte chunks per row).
// Chunk XOR-swizzle keyed on (row>>1)&7 makes the 16 consecutive rows a ds_read_b128
// lane group touches land on 16 distinct 16-byte slots of the 256-byte bank row.
TUP_DEVICE int swz128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// max / sum over the four lane groups g (lanes l, l^16, l^32, l^48) without LDS: v_permlane16_swap / v_permlane32_swap
// (hipcc folds an arithmetic combination of the swap's two results to its first operand when both inputs are the same value --
// it drops the cross-lane exchange -- so the second result passes through an empty asm; v_max in asm also avoids the
// canonicalising v_max x, x pair hipcc puts in front of fmaxf)
// The lane id out of thin air (two VALU ops, no input register): inside a long loop per-lane addresses rebuilt from it cost a few
// instructions, where the hoisted originals get spilled by hipcc and come back through scratch_load + s_waitcnt vmcnt(0) -- a
// wait that also drains every LDS-DMA in flight.  volatile: not hoisted, not merged.
TUP_DEVICE int lane_id_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
TUP_DEVICE uint32_t opaque_copy(uint32_t u) { asm volatile("" : "+v"(u)); return u; }
TUP_DEVICE float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TUP_DEVICE float rows_max(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = vmax(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, opaque_copy(r[1])));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return vmax(__builtin_bit_cast(float, q[0]), __builtin_bit_cast(float, opaque_copy(q[1])));
}
TUP_DEVICE float rows_sum(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, opaque_copy(r[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, opaque_copy(q[1]));
}

// erf(z) ~= zc * Q(zc^2), zc = clamp(z, +-2.9), Q = degree-9 minimax fit of erf(z)/z on [0, 2.9^2] CONSTRAINED to 2.9 * Q(2.9^2) = 1,
// so the clamped tail is exactly saturated (an unconstrained fit leaves erf at 0.99996 there and GELU's error then grows as
// |x| * 2e-5 for x < -4.1).  |error| <= 1.3e-4 in erf, <= 9.1e-5 in GELU for every x (below the bf16 rounding of every
// consumer); pure FMA chain -- no v_exp / v_rcp (libm erff is ~40 instructions; the GELU epilogue of mlp.0 evaluates 94 M of
// these per 8-image forward and was VALU-bound on it).  The f32x2 form lets hipcc emit v_pk_fma_f32 (2 values / issue).
typedef __attribute__((ext_vector_type(2))) float f32x2;

TUP_DEVICE f32x2 fast_erf2(f32x2 z) {
    f32x2 zc;
    zc[0] = fminf(fmaxf(z[0], -2.9f), 2.9f);
    zc[1] = fminf(fmaxf(z[1], -2.9f), 2.9f);
    const f32x2 w = zc * zc;
    f32x2 q = w * 1.2033207800518448e-08f + -4.4140491617154536e-07f;
    q = q * w + 6.2528506609435753e-06f;
    q = q * w + -3.6053944566514376e-05f;
    q = q * w + -7.2369640518118916e-05f;
    q = q * w + 0.0026871317284937087f;
    q = q * w + -0.02176163837525465f;
    q = q * w + 0.10692235311582286f;
    q = q * w + -0.37271276562605715f;
    q = q * w + 1.1276929106383919f;
    return zc * q;
}
TUP_DEVICE float fast_erf(float z) { return fast_erf2(f32x2{z, z})[0]; }

// GELU of N value pairs with the N polynomial chains advanced in lockstep (step-major order): a dependent
// v_pk_fma_f32 needs its predecessor's result, so one chain at a time leaves the VALU waiting (hipcc pads it with
// s_nop); N independent chains fill those slots.  Same arithmetic as gelu_erf2 below, value for value.
template <int N>
TUP_DEVICE void gelu_erf2_batch(f32x2 (&x)[N]) {
    // gelu(x) = x * (0.5 + xc * R(xc^2)), xc = clamp(x, +-C): R = a degree-7 minimax fit of (Phi(x) - 0.5) / x on [0, C^2],
    // weighted by x (the error that matters is GELU's, x times Phi's), constrained to C * R(C^2) = 0.5 so that the clamped tail is
    // exactly saturated, with the clamp point itself part of the search (C = 4.16).  |error| <= 8.7e-5 for every x when evaluated
    // in fp32 -- below the degree-9 fit with the fixed clamp it replaces (9.7e-5) at two FMAs less: 12 instructions per value pair
    // (2 v_med3, 1 packed multiply, 7 + 1 packed FMAs, 1 packed multiply).
    f32x2 xc[N], u[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        xc[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -4.16f, 4.16f);
        xc[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -4.16f, 4.16f);
        u[i] = xc[i] * xc[i];
        q[i] = u[i] * -9.3762207218685175e-10f + 8.1762115989189489e-08f;
    }
    constexpr float C[6] = {-3.0998062963740323e-06f, 6.7728779820009944e-05f, -0.00095683661677124407f, 0.0093223106686782069f,
                            -0.065572048515984249f, 0.39849236879961203f};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = q[i] * u[i] + C[k];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] * (xc[i] * q[i] + 0.5f);
}

// ---- GELU in packed fp16, for the fused inference MLPs (fused_attn.hip, fused_blocks.hip) ----
// mlp.0's weight and bias reach those kernels scaled by 1/4 (exact in bf16 / fp32: packing.pack_fc1_fused_q), so FC1's accumulators
// hold x' = x / 4.  gelu(x) / 4 = x' (0.5 + xc R(xc^2 - 0.5)), xc = clamp(x', +-1) [i.e. x clamped at +-4], R of degree 6 fitted to
// (Phi(4 x') - 0.5) / x' with weight x^2 and the clamped tail pinned (C R(C^2 - 0.5) = 0.5); the hidden tile stays fp16 and FC2 runs
// on v_mfma_f32_16x16x32_f16 against 4 W2 in fp16 (packing.pack_fc2_h4), so the three factors of 4 cancel exactly.
// Why fp16: a v_pk_*_f16 instruction costs ONE VALU issue for two values where v_pk_*_f32 costs two and stalls beside MFMAs
// (scripts/microbench_valu.hip: 12.8 vs 29.2 cycles per instruction next to a wave streaming MFMAs); 12 instructions per value
// pair including the conversion.  Accuracy, fp16 arithmetic emulated over x in [-6, 6] and N(0, 1.5): rms |error| 4.6e-4, max 5.1e-3
// -- the bf16 hidden tile it replaces has rms 2.4e-3, max 1.6e-2 (the polynomial's fp16 rounding is below the bf16 rounding of the
// value it used to be stored as).  The scaling keeps every coefficient of R between 0.67 and 2.2 (in u = xc^2 alone the high
// orders underflow fp16) and the Horner chain well conditioned (|s| <= 0.5).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int N>
TUP_DEVICE void gelu16_batch(h2 (&x)[N]) {
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    h2 xc[N], sv[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        xc[i] = __builtin_elementwise_min(__builtin_elementwise_max(x[i], -one), one);
        sv[i] = __builtin_elementwise_fma(xc[i], xc[i], h2{(_Float16)-0.5f, (_Float16)-0.5f});
        q[i] = __builtin_elementwise_fma(sv[i], h2{(_Float16)1.51615563f, (_Float16)1.51615563f}, h2{(_Float16)-2.11659751f, (_Float16)-2.11659751f});
    }
    constexpr float C[5] = {1.54543088f, -1.13520344f, 0.88632128f, -0.6753973f, 0.70388307f};
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(q[i], sv[i], h2{(_Float16)C[k], (_Float16)C[k]});
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] * __builtin_elementwise_fma(xc[i], q[i], h2{(_Float16)0.5f, (_Float16)0.5f});
}
// FC1 accumulators (x / 4, fp32) of both token tiles -> the fp16 hidden fragments gelu(x) / 4 (the B operand of FC2)
template <int TGN>
TUP_DEVICE void gelu16_fragments(const f32x4 (&acc1)[TGN][2], bf16x8 (&hfr)[TGN]) {
    h2 hv[4 * TGN];                    // all token tiles in lockstep: 4 independent chains per tile
#pragma unroll
    for (int tg = 0; tg < TGN; ++tg)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            hv[tg * 4 + hh * 2 + 0] = __builtin_convertvector(f32x2{acc1[tg][hh][0], acc1[tg][hh][1]}, h2);
            hv[tg * 4 + hh * 2 + 1] = __builtin_convertvector(f32x2{acc1[tg][hh][2], acc1[tg][hh][3]}, h2);
        }
#ifndef TUP_EXP_NOGELU          // timing experiment (wrong results): the whole-block kernel without the GELU arithmetic = the most that
                                // hiding it behind the fc1 / fc2 MFMAs could gain
    gelu16_batch<4 * TGN>(hv);
#endif
#pragma unroll
    for (int tg = 0; tg < TGN; ++tg) {
        u32x4 pk;
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[q] = __builtin_bit_cast(uint32_t, hv[tg * 4 + q]);
        hfr[tg] = __builtin_bit_cast(bf16x8, pk);          // fp16 bits in the kernels' 16-byte fragment type
    }
}
TUP_DEVICE f32x4 mfma16x16x32_f16(bf16x8 a, bf16x8 b, f32x4 c) {         // operands: fp16 bits carried as 16-byte fragments
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// nn.GELU() default = exact erf form (reference model.py:148)
TUP_DEVICE f32x2 gelu_erf2(f32x2 x) {
    f32x2 v[1] = {x};
    gelu_erf2_batch<1>(v);
    return v[0];
}
TUP_DEVICE float gelu_erf(float x) { return gelu_erf2(f32x2{x, x})[0]; }

TUP_DEVICE float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Stateless dropout mask: keep(element) = hash(seed, element index) >= thresh, thresh = p * 2^32.
// The same function is evaluated by the forward and the backward kernels (nothing is stored) and by the
// CPU checker in tests (tests/test_hip_dropout.py), bit for bit.
TUP_DEVICE uint32_t drop_hash(uint32_t seed, uint32_t idx) {
#ifdef TUP_CHEAP_HASH          // timing experiment only: what the hash costs the attention kernels
    return idx * 0x9E3779B1u + seed;
#endif
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
TUP_DEVICE float drop_scale(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
    return drop_hash(seed, idx) >= thresh ? inv_keep : 0.f;
}
// Attention-probability masks (attn_drop, model.py:80,127; nn.MultiheadAttention's dropout in the ResidualTransformer): ONE hash
// decides TWO neighbouring keys.  Element index idx = row * ncols + key with ncols even (64 or 3600): h = drop_hash(seed, idx >> 1);
// the even key takes the low 16 bits, the odd key the high 16 bits; keep iff field >= thresh16 = round(p * 65536).  The kernels get
// thresh_hi = thresh16 << 16 (a field compares against it in place) and inv_keep = 65536 / (65536 - thresh16) (the exact keep
// probability of the mask, so E[mask * inv_keep] = 1).  The hash is three quarter-rate multiplies + five xor / shift per call and was
// the most expensive thing per score of the attention kernels (1.0 ms of the 9.2 ms ResidualTransformer step): a lane
// that holds four consecutive keys of a row now pays two of them instead of four.
TUP_DEVICE float drop_pair_lo(uint32_t h, uint32_t thresh_hi, float inv_keep) { return (h << 16) >= thresh_hi ? inv_keep : 0.f; }
TUP_DEVICE float drop_pair_hi(uint32_t h, uint32_t thresh_hi, float inv_keep) { return h >= thresh_hi ? inv_keep : 0.f; }
TUP_DEVICE float drop_pair(uint32_t seed, uint32_t idx, uint32_t thresh_hi, float inv_keep) {      // one element (a lane whose values are not neighbours)
    const uint32_t h = drop_hash(seed, idx >> 1);
    return ((idx & 1u) ? h : (h << 16)) >= thresh_hi ? inv_keep : 0.f;
}
// four consecutive elements idx0 .. idx0 + 3, idx0 a multiple of 4
TUP_DEVICE void drop_pair4(uint32_t seed, uint32_t idx0, uint32_t thresh_hi, float inv_keep, float (&m)[4]) {
    const uint32_t h0 = drop_hash(seed, idx0 >> 1), h1 = drop_hash(seed, (idx0 >> 1) + 1u);
    m[0] = drop_pair_lo(h0, thresh_hi, inv_keep); m[1] = drop_pair_hi(h0, thresh_hi, inv_keep);
    m[2] = drop_pair_lo(h1, thresh_hi, inv_keep); m[3] = drop_pair_hi(h1, thresh_hi, inv_keep);
}
// four elements of ONE key column, rows r0 .. r0 + 3 (idx0 = r0 * ncols + key; the layout of the backward kernels that keep queries along
// a lane's values).  The two keys of a hash pair sit in neighbouring lanes (key parity = lane parity: the key tile starts at an even
// key, ncols is even), so each lane hashes TWO of the four rows and takes the other two from its neighbour (quad_perm DPP): two
// hashes per four decisions here as well.  All lanes of the wave must be active.
TUP_DEVICE void drop_pair4_rows(uint32_t seed, uint32_t idx0, uint32_t ncols, uint32_t thresh_hi, float inv_keep, float (&m)[4]) {
    const bool odd = (idx0 & 1u) != 0u;
    const uint32_t ia = (idx0 + (odd ? 2u * ncols : 0u)) >> 1;          // even lane: rows 0, 1; odd lane: rows 2, 3
    const uint32_t ha = drop_hash(seed, ia), hb = drop_hash(seed, ia + (ncols >> 1));
    const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ha, 0xB1, 0xf, 0xf, false);      // quad_perm(1,0,3,2): lane ^ 1
    const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hb, 0xB1, 0xf, 0xf, false);
    const uint32_t h[4] = {odd ? pa : ha, odd ? pb : hb, odd ? ha : pa, odd ? hb : pb};
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = (odd ? h[e] : (h[e] << 16)) >= thresh_hi ? inv_keep : 0.f;
}
// host side: p -> (thresh_hi, inv_keep)
inline void drop_pair_params(float p, uint32_t& thresh_hi, float& inv_keep) {
    if (!(p > 0.f)) { thresh_hi = 0u; inv_keep = 1.f; return; }
    long t = (long)((double)p * 65536.0 + 0.5);
    if (t < 1) t = 1;
    if (t > 65535) t = 65535;
    thresh_hi = (uint32_t)t << 16;
    inv_keep = (float)(65536.0 / (65536.0 - (double)t));
}

// Raises a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; a process-wide flag would leave
// every device after the first at the 64 KB default).
#define TUP_SET_DYN_LDS(fn, bytes) do { \
        static unsigned long long done_ = 0; int dev_ = 0; \
        hipError_t e_ = hipGetDevice(&dev_); if (e_ != hipSuccess) return (int)e_; \
        if (!((done_ >> (dev_ & 63)) & 1ull)) { \
            e_ = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) return (int)e_; \
            done_ |= 1ull << (dev_ & 63); } } while (0)

// A/B switches, tuning knobs and timing ablations are read from the environment ONLY in the diagnostic library (`make diag`,
// -DTUP_DIAG); the product library compiles the defaults in and reads no TUP_* variable.
#ifdef TUP_DIAG
#define TUP_ENV_FLAG(name) (getenv(name) != nullptr)
#define TUP_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define TUP_ENV_FLAG(name) false
#define TUP_ENV_INT(name, dflt) (dflt)
#endif

#define TUP_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
