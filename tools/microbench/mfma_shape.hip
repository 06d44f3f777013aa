// Which bf16 MFMA shape sustains more FLOP/s in an LDS-fed loop shaped like the split-bf16 convolution step?
//   A: 12 ds_read_b128 + 24 x v_mfma_f32_32x32x16_bf16   (wave tile 64x64, K = 16, six products: the shipped kernels)
//   B: 20 ds_read_b128 + 48 x v_mfma_f32_16x16x32_bf16   (same tile, the six products paired into three K = 32 MFMAs)
// Same FLOPs per iteration, random operands, 256-thread workgroups, two per CU.  MI355X_MICROARCH.md (DVFS give-back, item 7)
// reports the 16x16x32 loop 12-15 % faster in wall time at equal cycles on random data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

constexpr int LDS_UNITS = 3072;      // 48 KB of 16-byte units

__device__ void fill(uint4* lds) {
    for (int i = threadIdx.x; i < LDS_UNITS; i += 256) {
        uint32_t h = hash(i * 977u + blockIdx.x * 131u);
        // random bf16 values in [-2, 2): sign + exponent 126..128 + random mantissa
        auto w = [&](uint32_t r) { uint32_t lo = ((r & 0x8000u) | ((126u + (r & 1u)) << 7) | ((r >> 1) & 0x7fu)); uint32_t r2 = hash(r);
                                   uint32_t hi = ((r2 & 0x8000u) | ((126u + (r2 & 1u)) << 7) | ((r2 >> 1) & 0x7fu)); return lo | (hi << 16); };
        lds[i] = make_uint4(w(h), w(h + 1), w(h + 2), w(h + 3));
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void loop32(float* out, int iters) {
    __shared__ uint4 lds[LDS_UNITS];
    fill(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; it++) {
        const int base = ((it & 3) * 512 + wave * 32) % (LDS_UNITS - 1024);
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                fa[t][p] = __builtin_bit_cast(bf16x8, lds[base + (p * 2 + t) * 64 + lane]);
                fb[t][p] = __builtin_bit_cast(bf16x8, lds[base + 512 + (p * 2 + t) * 64 + lane]);
            }
#pragma unroll
        for (int pa = 0; pa < 3; pa++)
#pragma unroll
            for (int pb = 0; pb < 3 - pa; pb++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][pa], fb[b][pb], acc[a][b], 0, 0, 0);
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void loop16(float* out, int iters) {
    __shared__ uint4 lds[LDS_UNITS];
    fill(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][4];
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int r = 0; r < 4; r++) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; it++) {
        const int base = ((it & 3) * 512 + wave * 32) % (LDS_UNITS - 1400);
        bf16x8 fa[4][2], fb[4][3];        // A forms [a1|a2], [a1|a3]; B forms [b2|b1], [b3|b1], [b1|b2]
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int f = 0; f < 2; f++) fa[t][f] = __builtin_bit_cast(bf16x8, lds[base + (f * 4 + t) * 64 + lane]);
#pragma unroll
            for (int f = 0; f < 3; f++) fb[t][f] = __builtin_bit_cast(bf16x8, lds[base + 512 + (f * 4 + t) * 64 + lane]);
        }
#pragma unroll
        for (int prod = 0; prod < 3; prod++)
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][prod == 1 ? 1 : 0], fb[b][prod], acc[a][b], 0, 0, 0);
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int r = 0; r < 4; r++) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 2048, iters = 6000;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = (double)blocks * 4 * iters * 24.0 * 32 * 32 * 16 * 2;
    std::vector<float> t32, t16;
    for (int round = 0; round < 7; round++) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(loop32, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (round) t32.push_back(ms);
        hipEventRecord(e0); hipLaunchKernelGGL(loop16, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (round) t16.push_back(ms);
    }
    std::sort(t32.begin(), t32.end()); std::sort(t16.begin(), t16.end());
    printf("32x32x16: median %.3f ms (min %.3f) = %.0f TFLOP/s bf16\n", t32[t32.size() / 2], t32[0], flop / (t32[t32.size() / 2] * 1e-3) / 1e12);
    printf("16x16x32: median %.3f ms (min %.3f) = %.0f TFLOP/s bf16\n", t16[t16.size() / 2], t16[0], flop / (t16[t16.size() / 2] * 1e-3) / 1e12);
    printf("ratio 16x16x32 / 32x32x16 (wall time) = %.3f\n", t16[t16.size() / 2] / t32[t32.size() / 2]);
    return 0;
}
