// Largest finite magnitude of a tensor as AMAX_PARTS partial maxima: the power-of-two operand scales of PASTA_MATH_F16X3
// (conv_common.h) come from it.  One pass over the tensor at HBM rate; bytes = numel * sizeof(T).
#include "conv_common.h"

namespace pasta {

// 256 workgroups (one partial each) of 1024 threads, 16-byte loads, four in flight per thread.  Non-finite elements are
// skipped (compare on the bit pattern: |v| as an integer is below 0x7f800000 exactly for finite v), so one inf or NaN
// poisons only the outputs that really contain it, not the scale of the whole tensor.
template <class T>
__global__ __launch_bounds__(1024) void tensor_amax_kernel(const T* __restrict__ x, int64_t numel, float* __restrict__ parts) {
    constexpr int V = 4;                 // elements per ld4
    uint32_t m = 0;
    auto take = [&](float v) {
        const uint32_t b = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
        m = (b < 0x7f800000u && b > m) ? b : m;
    };
    // elements in front of the first 16-byte boundary (views into a larger tensor) are taken one by one
    int64_t head = (int64_t)(((16 - ((uintptr_t)x & 15)) & 15) / sizeof(T));
    head = head < numel ? head : numel;
    if (blockIdx.x == 1 % gridDim.x && threadIdx.x < head) take(ld<T>(x + threadIdx.x));
    x += head; numel -= head;
    const int64_t packs = numel / V;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 3 * stride < packs; i += 4 * stride) {
        const float4 a = ld4<T>(x + i * V), b = ld4<T>(x + (i + stride) * V), c = ld4<T>(x + (i + 2 * stride) * V), d = ld4<T>(x + (i + 3 * stride) * V);
        take(a.x); take(a.y); take(a.z); take(a.w); take(b.x); take(b.y); take(b.z); take(b.w);
        take(c.x); take(c.y); take(c.z); take(c.w); take(d.x); take(d.y); take(d.z); take(d.w);
    }
    for (; i < packs; i += stride) {
        const float4 a = ld4<T>(x + i * V);
        take(a.x); take(a.y); take(a.z); take(a.w);
    }
    // tail elements (numel not a multiple of the pack)
    if (blockIdx.x == 0)
        for (int64_t e = packs * V + threadIdx.x; e < numel; e += 1024) take(ld<T>(x + e));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    __shared__ uint32_t wm[16];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        for (int w = 0; w < 16; w++) r = wm[w] > r ? wm[w] : r;
        parts[blockIdx.x] = __builtin_bit_cast(float, r);
    }
}

template <class T> static void launch_amax(const void* x, int64_t numel, float* parts, hipStream_t s) {
    hipLaunchKernelGGL(tensor_amax_kernel<T>, dim3(AMAX_PARTS), dim3(1024), 0, s, (const T*)x, numel, parts);
}

int tensor_amax(const void* x, int64_t numel, int dtype, float* parts, hipStream_t s) {
    // PASTA_MATH_F16X3 exists for fp32 storage only (16-bit tensors are their own operands)
    switch (dtype) {
        case PASTA_F32: launch_amax<float>(x, numel, parts, s); break;
        default: return fail("tensor_amax: dtype code %d is not supported (fp32 tensors only)", dtype);
    }
    return launch_status("tensor_amax");
}

}  // namespace pasta

extern "C" int pasta_tensor_amax(const void* x, int64_t numel, int dtype, float* parts, void* stream) {
    using namespace pasta;
    PASTA_CHECK(x && parts, "tensor_amax: null pointer");
    PASTA_CHECK(numel >= 1, "tensor_amax: empty tensor");
    PASTA_CHECK(((uintptr_t)x & 3) == 0 && ((uintptr_t)parts & 15) == 0, "tensor_amax: x must be 4-byte aligned and parts 16-byte aligned");
    return tensor_amax(x, numel, dtype, parts, (hipStream_t)stream);
}
