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

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam's update (reference train.py:104,139: Adam, default betas / eps, no weight decay, no amsgrad) for ALL
// parameters in one launch.  The parameters, their gradients and the two moment buffers stay separate tensors: a segment table
// carries their base pointers and the step-dependent scalars, a chunk table maps workgroups to (segment, first element).
//   m = m + (g - m) * (1 - beta1)                 (exp_avg.lerp_(grad, 1 - beta1))
//   v = v * beta2 + g * g * (1 - beta2)           (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2))
//   p = p - step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps),   step_size = lr / bias_correction1
// Bound: HBM (4 reads + 3 writes of 4 B per element).
// ------------------------------------------------------------------------------------------------
struct AdamSeg {
    float* p; const float* g; float* m; float* v;
    long long n;
    float step_size, inv_sqrt_bc2, beta2, omb1, omb2, eps;      // omb = 1 - beta, rounded from double on the host as torch does
};
static_assert(sizeof(AdamSeg) == 64, "segment record = 64 bytes (the host packs it as 8 int64 words)");

namespace {
constexpr int ADAM_CHUNK = 4096;
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamSeg* __restrict__ segs, const int* __restrict__ chunks)
{
    const int seg = chunks[2 * blockIdx.x], first = chunks[2 * blockIdx.x + 1];
    const AdamSeg s = segs[seg];
    const long long end = min((long long)first + ADAM_CHUNK, s.n);
    const float omb1 = s.omb1, omb2 = s.omb2;
    for (long long i = first + threadIdx.x; i < end; i += 256) {
        const float g = s.g[i];
        float m = s.m[i], v = s.v[i];
        m = m + (g - m) * omb1;
        v = v * s.beta2 + g * g * omb2;
        const float denom = sqrtf(v) * s.inv_sqrt_bc2 + s.eps;
        s.m[i] = m; s.v[i] = v;
        s.p[i] = s.p[i] - s.step_size * (m / denom);
    }
}
}  // namespace

// segs: device array [nseg] of 64-byte records {p, g, m, v (device pointers), n (int64), step_size, 1/sqrt(bias_correction2),
// beta2, 1 - beta1, 1 - beta2, eps (fp32)}; chunks: device int [nchunks][2] = (segment, first element), 4096 elements per chunk.
extern "C" int tup_adam_step(const void* segs, const int* chunks, int nchunks, void* stream)
{
    if (nchunks <= 0) return 0;
    adam_step_kernel<<<dim3((unsigned)nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>((const AdamSeg*)segs, chunks);
    TUP_CHECK_LAUNCH();
    return 0;
}
