// Shared helpers for libpasta_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

// V elements moved as one 16-byte (or narrower) access.
template <class T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

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
