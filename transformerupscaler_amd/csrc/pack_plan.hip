// Weight re-packing after an optimizer step (training) as two launches.
//
// The packed layouts the kernels read (bf16 MFMA tiles, per-head slices, transposes for the input-gradient GEMMs, fp32
// copies) are pure gathers of the module's parameters: element i of a packed tensor is parameter element map[i], or zero
// padding.  packing.py builds them with torch index / permute / cat / cast calls -- ~75 aten launches per training step,
// once per step because Adam moves every weight (reference train.py:139).  pack_plan.py derives map[] once from that same
// code (bit-plane tracing) and every later re-pack is this gather: all bf16 outputs in one launch, all fp32 outputs in
// another.  The parameters stay where torch keeps them: `src` is a table of their base pointers, `offs` the prefix sum of
// their sizes, map[i] an index into the concatenation.
// Bound: HBM (reads ~4 B + 4 B, writes 2-4 B per element; ~25 M elements per step).
#include "common.h"

namespace {

template <bool TO_BF16>
__global__ __launch_bounds__(256) void pack_gather_kernel(const float* const* __restrict__ src, const int* __restrict__ offs, int nparam,
                                                          const int* __restrict__ map, void* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int m = map[i];
        float v = 0.f;
        if (m >= 0) {
            int lo = 0, hi = nparam;                  // offs[lo] <= m < offs[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= m) lo = mid; else hi = mid; }
            v = src[lo][m - offs[lo]];
        }
        if constexpr (TO_BF16) reinterpret_cast<bf16_t*>(dst)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

}  // namespace

// dst[i] = map[i] < 0 ? 0 : concat(src[0], ..., src[nparam-1])[map[i]]   (dst bf16 when to_bf16, else fp32; round to nearest even,
// as Tensor.to(torch.bfloat16)).  src: device array of nparam device pointers (fp32 parameters); offs: device int [nparam + 1],
// offs[p] = first concatenated index of parameter p; map: device int [n].
extern "C" int tup_pack_gather(const void* src, const int* offs, int nparam, const int* map, void* dst, long long n, int to_bf16,
                               void* stream)
{
    if (n <= 0) return 0;
    if (nparam < 1) return (int)hipErrorInvalidValue;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (to_bf16)
        pack_gather_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const float* const*)src, offs, nparam, map, dst, (size_t)n);
    else
        pack_gather_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const float* const*)src, offs, nparam, map, dst, (size_t)n);
    TUP_CHECK_LAUNCH();
    return 0;
}
