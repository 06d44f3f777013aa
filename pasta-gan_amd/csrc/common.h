// Shared helpers for libpasta_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/pasta_hip.h"

namespace pasta {

// Thread-local error text returned by pasta_last_error().
char* error_buffer();
int   fail(const char* fmt, ...);

#define PASTA_CHECK(cond, ...)                         \
    do {                                               \
        if (!(cond)) return ::pasta::fail(__VA_ARGS__); \
    } while (0)

#define PASTA_HIP_CHECK(expr)                                                      \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess)                                                     \
            return ::pasta::fail("%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE).  Each call site remembers on which devices it has
// applied it as one bit per device in an atomic word: no `static bool` that a second device of the same process would find already
// set, and two threads racing here at worst both set the same value.
static inline void set_max_lds(std::atomic<uint64_t>& done, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}
#define PASTA_SET_LDS(FN, BYTES) do { static std::atomic<uint64_t> done_{0}; ::pasta::set_max_lds(done_, (const void*)(FN), (int)(BYTES)); } while (0)

static inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// Arithmetic type used inside kernels for a storage type.
template <class T> struct acc_of            { typedef float  type; };
template <>        struct acc_of<double>    { typedef double type; };

template <class T> __device__ __forceinline__ typename acc_of<T>::type ld(const T* p)        { return (typename acc_of<T>::type)(*p); }
template <>        __device__ __forceinline__ float ld<__half>(const __half* p)              { return __half2float(*p); }
template <class T> __device__ __forceinline__ void st(T* p, typename acc_of<T>::type v)      { *p = (T)v; }
template <>        __device__ __forceinline__ void st<__half>(__half* p, float v)            { *p = __float2half(v); }
template <>        __device__ __forceinline__ float ld<__bf16>(const __bf16* p)              { return (float)(*p); }
template <>        __device__ __forceinline__ void st<__bf16>(__bf16* p, float v)            { *p = (__bf16)v; }

// Four consecutive elements (pointer aligned to the four-pack) as fp32, and back: one 16-byte or 8-byte access.
template <class T> __device__ __forceinline__ float4 ld4(const T* p) { return make_float4(ld<T>(p), ld<T>(p + 1), ld<T>(p + 2), ld<T>(p + 3)); }
template <>        __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <>        __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p) {
    const uint2 r = *(const uint2*)p;
    return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                       __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
template <>        __device__ __forceinline__ float4 ld4<__half>(const __half* p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 h = *(const h4*)p;
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <class T> __device__ __forceinline__ void st4(T* p, float4 v) { st<T>(p, v.x); st<T>(p + 1, v.y); st<T>(p + 2, v.z); st<T>(p + 3, v.w); }
template <>        __device__ __forceinline__ void st4<float>(float* p, float4 v) { *(float4*)p = v; }
template <>        __device__ __forceinline__ void st4<__bf16>(__bf16* p, float4 v) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 f = {v.x, v.y, v.z, v.w};
    *(b4*)p = __builtin_convertvector(f, b4);
}
template <>        __device__ __forceinline__ void st4<__half>(__half* p, float4 v) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 f = {v.x, v.y, v.z, v.w};
    *(h4*)p = __builtin_convertvector(f, h4);
}

// V elements moved as one 16-byte (or narrower) access.
template <class T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

// Producer-side |max| (PASTA_MATH_F16X3, conv_common.h): a kernel that writes a tensor can leave the tensor's largest finite
// magnitude behind in `parts` -- PASTA_AMAX_PARTS floats the CALLER has zeroed -- so that the convolution consuming the tensor
// needs no scan of its own.  Non-negative floats order like their bit patterns, so the running maximum is an integer maximum
// and the commit one agent-scope atomic per wave (order-independent: the result is deterministic).
__device__ __forceinline__ void amax_take(uint32_t& m, float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
    m = (b < 0x7f800000u && b > m) ? b : m;
}
// Where this wave will leave its maximum, and what the slot holds NOW -- read at the start of the kernel, so that the round trip
// hides behind the kernel's own work.  A slot only grows: a stale value can cause a redundant atomic, never a missed one.
struct AmaxSlot { unsigned int* slot; uint32_t seen; };
__device__ __forceinline__ AmaxSlot amax_begin(float* parts) {
    AmaxSlot a = {nullptr, 0u};
    if (parts) {
        a.slot = (unsigned int*)parts + ((blockIdx.x + 37u * blockIdx.y + (threadIdx.x >> 6)) & 255u);
        if ((threadIdx.x & 63) == 0) a.seen = __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return a;
}
// One no-return atomic per wave, and only when it would raise what the wave saw (measured: sixteen thousand unconditional atomics
// on these eight cache lines cost a kernel as much as the scan they replace; a look at commit time holds every wave for an L2
// round trip at the end of its life).  After the first round of workgroups most waves see a slot at or above their own maximum.
__device__ __forceinline__ void amax_commit(uint32_t m, const AmaxSlot& a) {
    if (!a.slot) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    // waves of the first round all saw an empty slot: those look once more now (they finish at different times, so most find the
    // slot already raised) before spending an atomic
    if ((threadIdx.x & 63) == 0 && m > a.seen && m > __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.slot, m);
}

// The same for a whole workgroup of NT threads that owns a few words of LDS anyway (kernels of many small workgroups: one
// commit per workgroup instead of one per wave).  `red`: NT / 64 words of shared memory; contains a barrier.
template <int NT>
__device__ __forceinline__ void amax_commit_block(uint32_t m, const AmaxSlot& a, uint32_t* red) {
    if (!a.slot) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NT / 64; i++) m = red[i] > m ? red[i] : m;
        if (m > a.seen && m > __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.slot, m);
    }
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// floor(a / b) for b > 0 and any sign of a.
__host__ __device__ __forceinline__ int floordiv(int a, int b) {
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
// a mod b in [0, b) for b > 0.
__host__ __device__ __forceinline__ int posmod(int a, int b) {
    int r = a % b;
    return r < 0 ? r + b : r;
}

}  // namespace pasta
