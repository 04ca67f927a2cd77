// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the FastTransformer path.
// Wave = 64 lanes; MFMA fragments follow the gfx950 maps:
//   v_mfma_f32_16x16x32_bf16: A[row = l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col = l&15],
//                             C/D[row = 4*(l>>4)+reg][col = l&15]
//   v_mfma_f32_16x16x16_bf16: A[row = l&15][k = 4*(l>>4)+j], B[k = 4*(l>>4)+j][col = l&15], same C/D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TUP_DEVICE __device__ __forceinline__

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

TUP_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

TUP_DEVICE float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Stateless dropout mask: keep(element) = hash(seed, element index) >= thresh, thresh = p * 2^32.
// The same function is evaluated by the forward and the backward kernels (nothing is stored) and by the
// CPU checker in tests (tests/test_hip_dropout.py), bit for bit.
TUP_DEVICE uint32_t drop_hash(uint32_t seed, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
TUP_DEVICE float drop_scale(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
    return drop_hash(seed, idx) >= thresh ? inv_keep : 0.f;
}

#define TUP_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
