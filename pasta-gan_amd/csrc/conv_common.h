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

// Parameters of the forward-type kernels (conv2d, conv_transpose2d and both input gradients).
struct ConvFwdParams {
    const float* x; const float* wp; float* y;
    const float* iscale; const float* oscale;
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
    const float* S; const float* L; float* slab;
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
