// nn.L1Loss() of the training step (reference train.py:103,132: mean |output - target|) and its backward
// (sign(output - target) * grad / N) as two streaming kernels.  aten evaluates the forward as sub, abs, mean and the
// backward as sub, sign, mul -- six passes over tensors that are 796 MB each in the ResidualTransformer 6x step.
// Bound: HBM (forward reads 2 tensors, backward reads 2 and writes 1).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ partial, size_t n4, size_t n)
{
    float acc = 0.f;
    if (blockIdx.x == 0 && n4 * 4 + threadIdx.x < n) acc = fabsf(a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x]);   // n % 4 tail
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        acc += (fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ gout, float inv_n, float* __restrict__ ga, size_t n4, size_t n)
{
    const float gs = gout[0] * inv_n;
    if (blockIdx.x == 0 && n4 * 4 + threadIdx.x < n) {                                                                   // n % 4 tail
        const float d = a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x];
        ga[n4 * 4 + threadIdx.x] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[e] - y[e]; o[e] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f); }
        reinterpret_cast<f32x4*>(ga)[i] = o;
    }
}

}  // namespace

// partial[nblocks] = per-workgroup sums of |a - b| over n fp32 elements (any n; 16-byte aligned pointers); the caller adds them (in fp64) and
// divides by n.  nblocks <= 65535.
extern "C" int tup_l1_loss_partial(const float* a, const float* b, float* partial, long long n, int nblocks, void* stream)
{
    if (n <= 0) return 0;
    if (nblocks < 1 || nblocks > 65535) return (int)hipErrorInvalidValue;
    l1_partial_kernel<<<dim3(nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(a, b, partial, (size_t)(n / 4), (size_t)n);
    TUP_CHECK_LAUNCH();
    return 0;
}

// ga = sign(a - b) * gout[0] / n  (gout: the scalar upstream gradient, on the device).
extern "C" int tup_l1_loss_bwd(const float* a, const float* b, const float* gout, float* ga, long long n, void* stream)
{
    if (n <= 0) return 0;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    l1_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(a, b, gout, (float)(1.0 / (double)n), ga, (size_t)(n / 4), (size_t)n);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Measurement aid of bench.py (`sustained.clock_GHz`): out[0] = s_memtime (shader cycles), out[1] = s_memrealtime (100 MHz) of one
// wave, in stream order.  The shader clock a stretch of work ran at = (difference of two probes' out[0]) / (difference of their out[1])
// x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6).  Not part of the model path.
namespace {
__global__ void clock_probe_kernel(unsigned long long* out)
{
    if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime(); out[1] = __builtin_amdgcn_s_memrealtime(); }
}
}  // namespace
extern "C" int tup_clock_probe(void* out2, void* stream)
{
    clock_probe_kernel<<<dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream)>>>((unsigned long long*)out2);
    TUP_CHECK_LAUNCH();
    return 0;
}
