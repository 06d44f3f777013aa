// Shared declarations of the convolution family (conv_igemm.hip and the kernel headers it includes).
#pragma once
#include <type_traits>
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace pasta {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // sixteen bytes in registers (HIP's uint4 is a struct: arrays of it handed to a lambda end up in scratch)

// An empty asm that makes a VGPR value opaque to the SLP vectoriser (no instruction is emitted).
#define PASTA_KEEP_SCALAR(x) asm("" : "+v"(x))
// every component of a 16-byte vector counts as used: a partly used LDS read stays one ds_read_b128
#define PASTA_KEEP_WHOLE(q) asm("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))

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

// The same without a branch: the store loops of the kernels run sixty-four of these per lane, and with three or four wave-uniform
// scalar branches per element (the compiler does not unswitch them) they cost the dominant kernel 5 % -- profiles/r4_ab_epilogues.txt.
// "No activation" is expressed by the caller as b = 0, slope = 1, relu = false, gain = 1, clamp_on = false (EpiAct below): the value then
// passes unchanged (NaN and infinities included).  relu selects an exact 0 for v <= 0 and for NaN, as conv_epilogue does.
struct EpiAct { float slope, gain, clamp; bool relu, clamp_on, on; };
__device__ __forceinline__ EpiAct conv_epi_act(int act, float alpha, float gain, float clamp, bool fused) {
    EpiAct e;
    e.on = act != 0 && fused;
    e.slope = (e.on && act == 3) ? alpha : 1.f;
    e.relu = e.on && act == 2;
    e.gain = e.on ? gain : 1.f;
    e.clamp_on = e.on && clamp >= 0.f;
    e.clamp = clamp;
    return e;
}
__device__ __forceinline__ float conv_epilogue_u(float v, float b, const EpiAct& e) {
    v += b;
    const float neg = e.relu ? 0.f : v * e.slope;              // selects on wave-uniform values, not branches
    v = v > 0.f ? v : neg;
    v *= e.gain;
    const bool in = (v > -e.clamp && v < e.clamp) || !e.clamp_on;
    return in ? v : (v >= 0.f ? e.clamp : -e.clamp);
}
// The store loops are instantiated per case and chosen ONCE per workgroup.  SPEC kernels (the default arithmetic, 16-bit storage, the fp32
// tiles: what the training and inference steps run) get an instance per (whole tile of output rows?, no activation | activation | activation
// + clamp) -- with the activation's wave-uniform tests resolved at compile time the loop is 1.5 ms per training step faster than the
// branch-free folded form above (profiles/r4_ab_epilogues.txt) --; the other arithmetics keep ONE loop with the row test per element and
// the folded form (compile time: an instance is sixty-four unrolled stores, and those arithmetics are side lines).  Round 4 folded every
// kernel because the family was one translation unit that took 12 minutes with the instances; since round 5 the family is thirteen units
// compiled in parallel (conv_launch.h).  f(full, EpiCase<MODE>{}): full is std::bool_constant for SPEC kernels, bool otherwise.
template <int MODE> struct EpiCase {};          // 0: folded, branch-free (conv_epilogue_u); 1: no activation; 2: activation; 3: activation and clamp
template <int MODE>
__device__ __forceinline__ float conv_epilogue_c(float v, float b, const EpiAct& e, EpiCase<MODE>) {
    if constexpr (MODE == 0) return conv_epilogue_u(v, b, e);
    else if constexpr (MODE == 1) return v;
    else {
        v += b;
        const float neg = e.relu ? 0.f : v * e.slope;          // a select on a wave-uniform value, not a branch (relu: an exact 0 for v <= 0 and for NaN)
        v = v > 0.f ? v : neg;
        v *= e.gain;
        if constexpr (MODE == 3) v = (v > -e.clamp && v < e.clamp) ? v : (v >= 0.f ? e.clamp : -e.clamp);
        return v;
    }
}
#ifndef PASTA_EPI_CASES
#define PASTA_EPI_CASES 1       // build switch for same-box A/B (tools/ab_lib.sh): 0 = round 4's final form, an instance per whole-tile answer with the folded activation
#endif
template <bool SPEC, class F>
__device__ __forceinline__ void conv_epilogue_dispatch(bool full, const EpiAct& e, F&& f) {
    if constexpr (SPEC && !PASTA_EPI_CASES) {
        if (full) f(std::true_type{}, EpiCase<0>{}); else f(std::false_type{}, EpiCase<0>{});
    } else if constexpr (SPEC) {
        if (full) {
            if (!e.on) f(std::true_type{}, EpiCase<1>{}); else if (!e.clamp_on) f(std::true_type{}, EpiCase<2>{}); else f(std::true_type{}, EpiCase<3>{});
        } else {
            if (!e.on) f(std::false_type{}, EpiCase<1>{}); else if (!e.clamp_on) f(std::false_type{}, EpiCase<2>{}); else f(std::false_type{}, EpiCase<3>{});
        }
    } else {
        f(full, EpiCase<0>{});
    }
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

//------------------------------------------------------------------------------------
// PASTA_MATH_F16X3: fp32-equivalent products from THREE fp16 matrix-core products (v_mfma_f32_32x32x16_f16).
//
// An fp32 operand v is scaled by a power of two S chosen from the largest magnitude of its tensor (amax * S in [2^13, 2^14):
// exact, no overflow) and written as   v S = h + r,  h = fp16(v S)  (round to nearest, 11 significand bits),  r exact in fp32,
// |r| <= 2^-11 |v S|.  The residual is kept as  l' = fp16(2^11 r)  ("pre-scaled low piece": the same magnitude as h, so it is a
// NORMAL fp16 number wherever h is, and |r - 2^-11 l'| <= 2^-23 |v S| thanks to its own sign).  h h, h l' and l' h are
// exact in fp32 (11 x 11 bits), so with a third piece  h'' = h 2^-11  (an exponent shift) on the OTHER operand
//     a b S_a S_b = h_a h_b + h''_a l'_b + l_a h_b          (+ l_a l_b, dropped: <= 2^-22 |a b|)
// is three MFMAs into ONE fp32 accumulator set; the epilogue multiplies by 1 / (S_a S_b) (exact).  Representation error
// per operand <= 2^-23 relative (rms 4e-8; fp32's own rounding is 2^-24), i.e. the products are fp32-class, and the fp32
// accumulation error (measured rms 3 - 8e-7 at K = 2304) dominates as it does for an fp32 FMA chain.
// Pieces by index: 0 = h, 1 = low piece, 2 = h''.
//   forward-type kernels: A = weights (packed once per launch: h, l = fp16(r) unscaled, h'' = h 2^-11), B = activations staged as
//       (h, l'): TWO pieces, two thirds of the split-bf16 staging and LDS traffic.  Products (A,B): (0,0) (2,1) (1,0).
//       Round 4, measured and dropped (profiles/r4_ab_hpp_in_registers.txt, same-box A/B of two source trees under rocprofv3):
//       forming h'' in registers from the h fragment (four v_pk_mul_f16 per fragment) instead of packing, staging and reading it
//       -- a third of the weight stream, of its LDS stores and of its fragment reads gone: the eight-wave 2-D tile 290.9 -> 290.1 us
//       per launch (nothing), the four-wave 64 x 256 tile 294 -> 329 us (11 % SLOWER), the base kernel 121 -> 125 us: eight more
//       VALU instructions per step and wave next to the MFMAs cost what the LDS bytes saved, and more where a wave also stages
//       three units per chunk.
//       The weight scale is PER OUTPUT ROW of the GEMM (round 4): the packing kernel takes the largest magnitude of each
//       output channel's weights (after wscale and, for per-sample weights, after the modulation) itself, so no |max| of w is
//       passed in or cached anywhere, and a row's elements keep full precision down to 2^-16 of THEIR ROW's largest (below
//       that the unscaled l leaves fp16's normal range and an element keeps 11..22 bits: absolute error <= 2^-28 of the row's
//       amax per term).  The epilogue multiplies row o by 1 / (S_x S_w[o]) (p.w_rowinv).  Activations keep full precision
//       down to 2^-28 of their tensor's largest element.
//   weight-gradient kernels: both operands are activations and symmetric, so the exponent-shift trick buys nothing there (h'' of
//       EITHER operand leaves fp16's normal range below 2^-16 of its tensor's largest element, whichever way the low pieces are
//       scaled): each operand is (h, l = fp16(r)), TWO pieces, products (S,L): (0,0) (0,1) (1,0) -- full precision for elements
//       within 2^-17 of their tensor's largest, absolute error <= 2^-39 amax per element below that.
// Scales: the tensor's |max| arrives as 256 partial maxima (pasta_tensor_amax: one pass at HBM rate, non-finite elements
// skipped so that an inf / NaN stays local); every wave reduces them itself (one 16-byte load per lane) -- no finalising
// launch, no atomics, no host round trip.
constexpr int NP_F16X3 = 4;             // pseudo piece count of the template parameter NP: fp16 pieces, three products
constexpr int AMAX_PARTS = 256;         // partial maxima per tensor

template <int NP> struct Arith {
    static constexpr bool f16x3 = NP == NP_F16X3;
    static constexpr int npa = f16x3 ? 3 : NP;          // A pieces in LDS
    static constexpr int npb = f16x3 ? 2 : NP;          // B pieces in LDS (forward-type kernels)
    static constexpr int npw = f16x3 ? 2 : NP;          // pieces of either operand in the weight-gradient kernels
};
// is the product (A piece pa) x (B piece pb) part of the arithmetic?  forward-type kernels
template <int NP> __host__ __device__ constexpr bool mm_on(int pa, int pb) {
    return NP == NP_F16X3 ? ((pa == 0 && pb == 0) || (pa == 2 && pb == 1) || (pa == 1 && pb == 0)) : (pa + pb < NP);
}
// ... weight-gradient kernels (S piece pa, L piece pb)
template <int NP> __host__ __device__ constexpr bool mmw_on(int pa, int pb) {
    return NP == NP_F16X3 ? (pa + pb < 2) : (pa + pb < NP);
}
// the matrix-core instruction of an arithmetic / storage type
template <int IO, int NP> __device__ __forceinline__ f32x16 mfma16(bf16x8_t a, bf16x8_t b, f32x16 acc) {
    if constexpr (IO == IO_F16 || NP == NP_F16X3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// S (power of two) and 1 / S for a tensor whose largest finite magnitude is amax: amax S in [2^13, 2^14); amax == 0 or
// subnormal: S = 2^120.  Exponent fields stay in [7, 247], so S, 1 / S and every v S are normal or exactly zero.
__host__ __device__ __forceinline__ void scale_from_amax(float amax, float& S, float& invS) {
    const uint32_t eb = __builtin_bit_cast(uint32_t, amax) >> 23;          // amax >= 0
    int field = 267 - (int)eb;                                             // 127 + 14 - (eb - 126)
    field = field < 7 ? 7 : field > 247 ? 247 : field;
    S = __builtin_bit_cast(float, (uint32_t)field << 23);
    invS = __builtin_bit_cast(float, (uint32_t)(254 - field) << 23);
}
// the largest of the AMAX_PARTS partial maxima at `parts`, computed redundantly by every wave (wave-uniform result)
__device__ __forceinline__ float amax_of_parts(const float* __restrict__ parts) {
    const float4 v = ((const float4*)parts)[threadIdx.x & 63];
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two scaled fp32 values -> h (packed fp16 pair) and the pre-scaled low piece l' = fp16(2^11 (v - h))
__device__ __forceinline__ void f16_split2(float v0, float v1, uint32_t& h, uint32_t& lp) {
    const f16x2_t hh = __builtin_convertvector(f32x2_t{v0, v1}, f16x2_t);
    h = __builtin_bit_cast(uint32_t, hh);
    float r0 = v0 - (float)hh[0], r1 = v1 - (float)hh[1];
    PASTA_KEEP_SCALAR(r0);
    lp = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{r0 * 2048.f, r1 * 2048.f}, f16x2_t));
}
// ... and with the low piece left unscaled: l = fp16(v - h) (weight-gradient kernels)
__device__ __forceinline__ void f16_split2_direct(float v0, float v1, uint32_t& h, uint32_t& l) {
    const f16x2_t hh = __builtin_convertvector(f32x2_t{v0, v1}, f16x2_t);
    h = __builtin_bit_cast(uint32_t, hh);
    float r0 = v0 - (float)hh[0], r1 = v1 - (float)hh[1];
    PASTA_KEEP_SCALAR(r0);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}
// AMAX_PARTS partial |max| of a contiguous tensor (amax.hip)
int tensor_amax(const void* x, int64_t numel, int dtype, float* parts, hipStream_t s);

// accumulator -> output value: demodulation scale and noise as mod_bias_act_kernel rounds them (fma(acc, d, noise * strength)),
// residual, then bias / activation / gain / clamp.  d = 1 and nz = 0 leave the accumulator bit-identical.
__device__ __forceinline__ float conv_scale_noise(float v, const float* osb, int o, float nz) {
    return osb ? fmaf(v, osb[o], nz) : v + nz;
}

// Parameters of the forward-type kernels (conv2d, conv_transpose2d and both input gradients).
// One packing job of pack_weights_f16x3_(pair_)kernel: where the packed rows and their scales go, and how the weight tensor is read
struct PackJob {
    void* wp; float* rowinv;
    int G, Ig, Og, Ig_pad, Og_pad, kh, kw, transposed, flip, pack_xcd_rows;
    float wscale;
};

struct ConvFwdParams {
    const float* x; const float* wp; float* y;         // x, y (and res): elements of type `io` behind these pointers
    const float* iscale; const float* oscale;
    const float* x_amax;                                // PASTA_MATH_F16X3: AMAX_PARTS partial |max| of x
    const float* x2; const float* x2_amax; int C1;      // conv1x1_f16x3_kernel: channels [C1, Cin) live in a second tensor x2 ([N, Cin - C1, H, W]); null = one tensor
    const float* w_rowinv;                              // PASTA_MATH_F16X3: [G][Og_pad] 1 / S_w of every packed weight row (pack_weights_f16x3_kernel)
    float* y_amax;                                      // optional (fused epilogue): zeroed partial |max| slots of y (common.h, amax_commit)
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
    // packed-K mode of conv_fwd_bf16x6_kernel (few input channels): "channel" k of the K loop is (input channel, tap) and lives
    // koff[k] bytes behind the pixel's base address in a zero-padded copy of the input; null = off
    const unsigned* koff;
    int xcd_order;                        // conv_fwd_rows2d_bf16x6_kernel, eight-wave tile: consecutive pixel tiles on ONE XCD (PASTA_XCD_ORDER=0: off)
    int x_pieces;                         // conv3x3s2_f16x3_kernel: x is PASTA_LAYOUT_PIECES16 (pieces.hip), p.x_amax the producer's bound row
};

// Tile choice.  O_pad multiple returned so that the caller can pack weights accordingly.
enum FwdTile { T128x128 = 0, T64x256 = 1, T32x256 = 2, T64x64 = 3 };

static int fwd_tile_bm(FwdTile t) { return t == T128x128 ? 128 : t == T32x256 ? 32 : 64; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Parameters of the weight-gradient kernels: dW[tap][a][b] = sum_pix S[a][pix] * L[b][pix * st + tap offset].
struct WgradParams {
    const float* S; const float* L; float* slab;        // S, L: elements of type `io`; slab: fp32
    const float* s_amax; const float* l_amax;           // PASTA_MATH_F16X3: AMAX_PARTS partial |max| of S and of L
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
    int xcd_order;          // split weight-gradient kernels: workgroup order (0: K slice fastest; 1: see wgrad_decode)
    int l_pieces;           // conv_wgrad3x3s2_pieces_kernel: L is PASTA_LAYOUT_PIECES16 (pieces.hip), l_amax the producer's bound row
};

}  // namespace pasta
