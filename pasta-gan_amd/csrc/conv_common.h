// Shared declarations of the convolution family (conv_igemm.hip and the kernel headers it includes).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace pasta {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_TAPS = 49;   // up to 7x7

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// bias + activation + gain + clamp of one output element (forward semantics of bias_act.cu:38-146, act 1..3)
__device__ __forceinline__ float conv_epilogue(float v, float b, int act, float alpha, float gain, float clamp) {
    v += b;
    if (act == 2) v = v > 0.f ? v : 0.f;
    else if (act == 3) v = v > 0.f ? v : v * alpha;
    v *= gain;
    if (clamp >= 0.f) v = (v > -clamp && v < clamp) ? v : (v >= 0.f ? clamp : -clamp);
    return v;
}

//------------------------------------------------------------------------------------
// Storage type of the activations (x, y, dy) of the split-bf16 kernel family.  IO_F32: fp32 tensors, split into bf16 pieces
// by the staging code.  IO_BF16 / IO_F16: 16-bit tensors in HBM (BASELINE config 5; the fp16 discriminator blocks of the
// reference, networks.py:1107-1120): the element IS the matrix-core operand, one product per multiply-add on
// v_mfma_f32_32x32x16_bf16 / _f16, fp32 accumulation, fp32 epilogue, result rounded once on the way out.  Weights and
// weight gradients stay fp32 in HBM (master copies); the packing kernel rounds the weights to the operand type.
enum { IO_F32 = 0, IO_F16 = 1, IO_BF16 = 3 };          // = the PASTA_F32 / PASTA_F16 / PASTA_BF16 dtype codes

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int IO> struct io_size { static constexpr int value = IO == IO_F32 ? 4 : 2; };

// one element at BYTE offset `off` of `base`, as fp32
template <int IO> __device__ __forceinline__ float io_ld(const char* base, unsigned off) {
    if constexpr (IO == IO_F32) return *(const float*)(base + off);
    else if constexpr (IO == IO_BF16) return __builtin_bit_cast(float, (uint32_t)(*(const uint16_t*)(base + off)) << 16);
    else return (float)(*(const _Float16*)(base + off));
}
// four consecutive elements starting at element pointer `p` (naturally aligned to the four-pack), as fp32
template <int IO> __device__ __forceinline__ float4 io_ld4(const void* p) {
    if constexpr (IO == IO_F32) return *(const float4*)p;
    else {
        const uint2 r = *(const uint2*)p;
        if constexpr (IO == IO_BF16)
            return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                               __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
        else {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            const f16x4 h = __builtin_bit_cast(f16x4, r);
            return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
    }
}
template <int IO> __device__ __forceinline__ float io_ld1(const void* p) {
    if constexpr (IO == IO_F32) return *(const float*)p;
    else if constexpr (IO == IO_BF16) return __builtin_bit_cast(float, (uint32_t)(*(const uint16_t*)p) << 16);
    else return (float)(*(const _Float16*)p);
}
// store fp32 `v` as element `idx` of `base`
template <int IO> __device__ __forceinline__ void io_st(void* base, int64_t idx, float v) {
    if constexpr (IO == IO_F32) ((float*)base)[idx] = v;
    else if constexpr (IO == IO_BF16) ((__bf16*)base)[idx] = (__bf16)v;
    else ((_Float16*)base)[idx] = (_Float16)v;
}
// two fp32 values -> one dword of two matrix-core operand elements (bf16 for fp32 / bf16 storage, f16 for f16 storage)
template <int IO> __device__ __forceinline__ uint32_t io_pack2(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    if constexpr (IO == IO_F16) {
        typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_));
    } else {
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
    }
}
// acc += A * B on the matrix cores, operands given as eight packed 16-bit elements
template <int IO> __device__ __forceinline__ f32x16 io_mfma(bf16x8_t a, bf16x8_t b, f32x16 acc) {
    if constexpr (IO == IO_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// accumulator -> output value: demodulation scale and noise as mod_bias_act_kernel rounds them (fma(acc, d, noise * strength)),
// residual, then bias / activation / gain / clamp.  d = 1 and nz = 0 leave the accumulator bit-identical.
__device__ __forceinline__ float conv_scale_noise(float v, const float* osb, int o, float nz) {
    return osb ? fmaf(v, osb[o], nz) : v + nz;
}

// Parameters of the forward-type kernels (conv2d, conv_transpose2d and both input gradients).
struct ConvFwdParams {
    const float* x; const float* wp; float* y;         // x, y (and res): elements of type `io` behind these pointers
    const float* iscale; const float* oscale;
    int io;                                             // IO_F32 / IO_F16 / IO_BF16
    int N, Cin, H, W;
    int Cout, OH, OW;
    int G, Ig, Og, Ig_pad, Og_pad, KK;   // KK = kh*kw slabs per group in wp
    int P, Q;                             // lattice extent
    int oy0, ox0, osy, osx;               // output pixel = (oy0 + p*osy, ox0 + q*osx)
    int isy, isx;                         // input base   = (p*isy, q*isx)
    int T;                                // taps of this lattice
    int ksplit;                           // > 1: K is cut into slices, partial sums go to `partial`
    float* partial;                       // [ksplit][N*Cout*OH*OW] when ksplit > 1
    int o_tiles;                          // output-channel tiles (blockIdx.y = ks * o_tiles + tile)
    int bf16x6;                           // weights packed as split-bf16 pieces, run conv_fwd_bf16x6_kernel
    const float* bias;                    // fused epilogue (pasta_conv_epilogue); act == 0: none
    const float* res;                     // [N, Cout, OH, OW] added before bias / activation, or null
    const float* noise;                   // [OH*OW] or [N][OH*OW] fp32, times noise_strength[0], added after the output scale (SynthesisLayer), or null
    const float* noise_strength;
    int noise_ps;                         // noise is per sample
    int act;
    float alpha, gain, clamp;
    int tap_dy[MAX_TAPS], tap_dx[MAX_TAPS], tap_slab[MAX_TAPS];     // int so that a wave-uniform index reads them with s_load_dword
    // conv_fwd_bf16x6_kernel reads its lattice from here: ncls lattices (the output parity classes of a stride-2
    // conv_transpose2d, else one) share a grid; class c owns taps [tap0, tap0 + T) of the tables above.
    int ncls;
    struct Lattice { int P, Q, oy0, ox0, T, tap0; } cls[4];
    // conv_fwd_rows_bf16x6_kernel (3-wide stride-1 lattices): smallest horizontal tap offset, and whether the three
    // taps of a kernel row are stored with descending offsets (input-gradient launches)
    int rows, rows_d0, rows_rev;
    int rows_y0;                          // conv_fwd_rows2d_bf16x6_kernel: smallest vertical tap offset
    // parity-pair mode of the row-reuse kernel (stride-2 conv_transpose2d): B-image offset of tap c, and whether the
    // two-tap column (taps 0 and 2) is the odd output column
    int pair_off[3], pair_bx;
};

// Tile choice.  O_pad multiple returned so that the caller can pack weights accordingly.
enum FwdTile { T128x128 = 0, T64x256 = 1, T32x256 = 2, T64x64 = 3 };

static int fwd_tile_bm(FwdTile t) { return t == T128x128 ? 128 : t == T32x256 ? 32 : 64; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// An empty asm that makes a VGPR value opaque to the SLP vectoriser (no instruction is emitted).
#define PASTA_KEEP_SCALAR(x) asm("" : "+v"(x))
// every component of a 16-byte vector counts as used: a partly used LDS read stays one ds_read_b128
#define PASTA_KEEP_WHOLE(q) asm("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))

// Parameters of the weight-gradient kernels: dW[tap][a][b] = sum_pix S[a][pix] * L[b][pix * st + tap offset].
struct WgradParams {
    const float* S; const float* L; float* slab;        // S, L: elements of type `io`; slab: fp32
    int io;                 // IO_F32 / IO_F16 / IO_BF16
    int N, SC, P, Q;        // S: [N, SC, P, Q]
    int LC, LH, LW;         // L: [N, LC, LH, LW]
    int G, Ag, Bg;
    int kh, kw, st, pad_h, pad_w;
    int cw_log2;            // chunk width = 1 << cw_log2, chunk height = 32 >> cw_log2
    int rows_total;         // N * P rows of S
    int qblocks;            // ceil(Q / CW)
    int chunks_total;       // ceil(rows_total / CHH) * qblocks
    int ksplit;             // K slices
    int a_tiles, b_tiles, tap_groups_r, tap_groups_s;
};

}  // namespace pasta
