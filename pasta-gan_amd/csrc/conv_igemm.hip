// Dense convolution family for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
// conv2d, conv_transpose2d and the weight gradient of either, NCHW fp32, exact-f32 products.
//
// Stands where the reference hands its convolutions to ATen/cuDNN
// (torch_utils/ops/conv2d_gradfix.py:38,43 forward; :125-128 input gradient through the
// transposed operator; :140-148 weight gradient).  Everything here is an implicit GEMM:
//
//   forward-type kernel   C[o][pix] = sum_{tap,i} Wp[tap][i][o] * X[i][pix + tap offset]
//       rows    = output channels of one group          (MFMA "A" operand = packed weights)
//       columns = a lattice of output pixels            (MFMA "B" operand = gathered activations)
//     conv2d is one lattice (all output pixels, input step = stride); conv_transpose2d with
//     stride u is u*u lattices (one per output parity class) so no multiply ever meets a
//     stuffed zero.  Optional per-(n,channel) input and output scales carry the StyleGAN2
//     modulation / demodulation (training/networks.py:74, 77-79).
//
//   weight-gradient kernel  dW[tap][a][b] = sum_pix S[a][pix] * L[b][pix*stride + tap offset]
//       S = the smaller-resolution tensor (dy for conv2d, x for conv_transpose2d), L the other.
//     K (= pixels) is split across workgroups; partial slabs are summed in a fixed order by a
//     second kernel that also writes PyTorch's [.., .., kh, kw] layout (bitwise reproducible).
//
// C/D fragment map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
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
// Weight packing: PyTorch layout -> [G][kh*kw][I_pad][O_pad] (O contiguous), zero padded so
// the GEMM's A-operand staging needs no bounds checks.

__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int G, int Ig,
                                                           int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                                                           int flip, float wscale) {
    const int64_t total = (int64_t)G * kh * kw * Ig_pad * Og_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int o = (int)(idx % Og_pad);
        int64_t r = idx / Og_pad;
        int i = (int)(r % Ig_pad); r /= Ig_pad;
        int t = (int)(r % (kh * kw));
        int g = (int)(r / (kh * kw));
        float v = 0.f;
        if (i < Ig && o < Og) {
            int ty = t / kw, tx = t - ty * kw;
            if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
            int64_t src = transposed ? (((int64_t)(g * Ig + i) * Og + o) * kh + ty) * kw + tx
                                     : (((int64_t)(g * Og + o) * Ig + i) * kh + ty) * kw + tx;
            v = w[src] * wscale;
        }
        wp[idx] = v;
    }
}

//------------------------------------------------------------------------------------
// Forward-type implicit GEMM.

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

template <int BM, int BN, int WMT, int WNT, int KC, int OCC = 1>   // OCC = minimum waves per SIMD asked of the register allocator
__global__ __launch_bounds__(256, OCC) void conv_fwd_kernel(ConvFwdParams p) {
    constexpr int WAVES_N = BN / (32 * WNT);
    static_assert((BM / (32 * WMT)) * WAVES_N == 4, "four waves per workgroup");
    constexpr int RSTEP = 256 / BN > 0 ? 256 / BN : 1;      // k-rows covered by one pass of the workgroup (B tile)
    constexpr int BPT = KC * BN / 256;                       // B elements per thread per chunk
    constexpr int A4_PER_ROW = BM / 4;
    constexpr int APT = (KC * A4_PER_ROW + 255) / 256;       // float4 A loads per thread per chunk
    static_assert(BN <= 256 && BPT >= 1, "tile/thread mapping");

    __shared__ float As[2][KC][BM];
    __shared__ float Bs[2][KC][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    const int64_t npix = (int64_t)p.N * p.P * p.Q;
    const int64_t pix_blk = (int64_t)blockIdx.x * BN;
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int chunks_all = p.T * NC;
    const int c_first = (int)((int64_t)chunks_all * ks / p.ksplit);
    const int nchunks = (int)((int64_t)chunks_all * (ks + 1) / p.ksplit) - c_first;

    // ---- B staging: this thread's pixel column is fixed for the whole K loop.
    const int bcol = tid % BN, brow0 = tid / BN;
    const int64_t mypix = pix_blk + bcol;
    const bool pix_ok = mypix < npix;
    int n_in = 0, py = 0, px = 0;
    if (pix_ok) {
        n_in = (int)(mypix / (p.P * p.Q));
        int rem = (int)(mypix - (int64_t)n_in * p.P * p.Q);
        py = rem / p.Q; px = rem - py * p.Q;
    }
    const float* xb = p.x + ((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW;
    const float* isb = p.iscale ? p.iscale + (int64_t)n_in * p.Cin + (int64_t)g * p.Ig : nullptr;
    const int iy_base = py * p.isy, ix_base = px * p.isx;

    // ---- A staging.
    const float* wb = p.wp + (int64_t)g * p.KK * p.Ig_pad * p.Og_pad + o_blk;

    float  breg[BPT];
    float4 areg[APT];
    unsigned bmask = 0;         // bit j: breg[j] is a real element (inside the image, channel < Ig)

    // Loader state: tap index and channel offset of the NEXT chunk to fetch, plus the per-tap
    // quantities derived from them (recomputed only when the tap changes: T times, not per chunk).
    int ld_t = c_first / NC, ld_c0 = (c_first - ld_t * NC) * KC;
    bool ld_ok = false;
    const float* ld_xp = xb;
    const float* ld_wt = wb;
    auto set_tap = [&](int t) {
        const int iy = iy_base + p.tap_dy[t], ix = ix_base + p.tap_dx[t];
        ld_ok = pix_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        ld_xp = ld_ok ? xb + iy * p.W + ix : xb;       // always readable; out-of-image taps are zeroed by the select below
        ld_wt = wb + (int64_t)p.tap_slab[t] * p.Ig_pad * p.Og_pad;
    };
    if (ld_t < p.T) set_tap(ld_t);
    // Unconditional loads from a clamped channel index: no branch and no wait sits between a load and the MFMAs that
    // cover its latency.
    auto load_chunk = [&]() {
        const int last = p.Ig - 1;
        bmask = 0;
#pragma unroll
        for (int j = 0; j < BPT; j++) {
            const int c = ld_c0 + brow0 + j * RSTEP;
            if (ld_ok && c < p.Ig) bmask |= 1u << j;         // validity is known now; the select waits until the store
        }
        if (isb) {
#pragma unroll
            for (int j = 0; j < BPT; j++) {
                const int c = ld_c0 + brow0 + j * RSTEP, cs = c < last ? c : last;
                breg[j] = ld_xp[(int64_t)cs * HW] * isb[cs];
            }
        } else {
#pragma unroll
            for (int j = 0; j < BPT; j++) {
                const int c = ld_c0 + brow0 + j * RSTEP, cs = c < last ? c : last;
                breg[j] = ld_xp[(int64_t)cs * HW];
            }
        }
        const float* wt = ld_wt + (int64_t)ld_c0 * p.Og_pad;
#pragma unroll
        for (int j = 0; j < APT; j++) {
            const int e = tid + j * 256, row = e / A4_PER_ROW, c4 = e - row * A4_PER_ROW;
            if (row < KC) areg[j] = *(const float4*)(wt + (int64_t)row * p.Og_pad + c4 * 4);
        }
        ld_c0 += KC;
        if (ld_c0 >= p.Ig_pad) {
            ld_c0 = 0;
            if (++ld_t < p.T) set_tap(ld_t);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BPT; j++) Bs[buf][brow0 + j * RSTEP][bcol] = (bmask >> j & 1u) ? breg[j] : 0.f;
#pragma unroll
        for (int j = 0; j < APT; j++) {
            const int e = tid + j * 256, row = e / A4_PER_ROW, c4 = e - row * A4_PER_ROW;
            if (row < KC) *(float4*)&As[buf][row][c4 * 4] = areg[j];
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    if (nchunks > 0) {          // an empty K slice (more slices than chunks) contributes zeros
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    const int kl = lane >> 5, jl = lane & 31;
    for (int ch = 0; ch < nchunks; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk();
#pragma unroll
        for (int kk = 0; kk < KC / 2; kk++) {
            float af[WMT], bf[WNT];
#pragma unroll
            for (int a = 0; a < WMT; a++) af[a] = As[buf][kk * 2 + kl][(wm * WMT + a) * 32 + jl];
#pragma unroll
            for (int b = 0; b < WNT; b++) bf[b] = Bs[buf][kk * 2 + kl][(wn * WNT + b) * 32 + jl];
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int b = 0; b < WNT; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: registers -> NCHW, rows = channels, lanes = consecutive pixels.
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        if (pix >= npix) continue;
        const int n = (int)(pix / (p.P * p.Q));
        const int rem = (int)(pix - (int64_t)n * p.P * p.Q);
        const int pp = rem / p.Q, qq = rem - pp * p.Q;
        float* yb = (p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW : p.y) +
                    ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + (p.oy0 + pp * p.osy) * p.OW + p.ox0 + qq * p.osx;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                if (o < p.Og) {
                    float v = acc[a][b][r];
                    if (osb) v *= osb[o];
                    if (p.act && p.ksplit == 1) v = conv_epilogue(v, p.bias ? p.bias[g * p.Og + o] : 0.f, p.act, p.alpha, p.gain, p.clamp);
                    yb[(int64_t)o * OHW] = v;
                }
            }
    }
}

template <int BM, int BN, int WMT, int WNT, int KC, int OCC = 1>
static void launch_fwd(const ConvFwdParams& p, hipStream_t s) {
    const int64_t npix = (int64_t)p.N * p.P * p.Q;
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    dim3 grid((unsigned)ceil_div64(npix, BN), q.o_tiles * q.ksplit, p.G);
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WMT, WNT, KC, OCC>), grid, dim3(256), 0, s, q);
}

// Tile choice.  O_pad multiple returned so that the caller can pack weights accordingly.
enum FwdTile { T128x128 = 0, T64x256 = 1, T32x256 = 2, T64x64 = 3 };

static int fwd_tile_bm(FwdTile t) { return t == T128x128 ? 128 : t == T32x256 ? 32 : 64; }

//------------------------------------------------------------------------------------
// Forward-type implicit GEMM on the bf16 matrix cores with fp32-equivalent products ("split-bf16").
//
// Every fp32 operand v is written as v = v1 + v2 + v3 with v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2):
// three bf16 pieces of 8 significand bits each carry the 24 bits of an fp32 significand, and a product of two bf16
// values is exact in fp32.  a*b is evaluated as a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (the three dropped terms are
// below 2^-24 |ab|), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs replace eight fp32 MFMAs of a
// quarter of the rate each: 2.67x the fp32-MFMA throughput at fp32 accuracy.
// Tile 128 x 128, K chunks of 16 channels of one tap; weights are split once by the packing kernel, activations by
// the staging code (after the optional modulation scale).  Layouts in LDS (per piece and per k-half of 8):
// [piece][half][row or pixel][8 bf16] so that a fragment is one conflict-free 16-byte read.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// An empty asm that makes a VGPR value opaque to the SLP vectoriser (no instruction is emitted).
#define PASTA_KEEP_SCALAR(x) asm("" : "+v"(x))

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)v;
    float r = v - (float)a;
    b = (__bf16)r;
    r -= (float)b;
    c = (__bf16)r;
}

// [g][tap][chunk of 16 channels][piece 3][half 2][O_pad][8]
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int G, int Ig,
                                                                int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                                                                int flip, float wscale) {
    const int64_t total = (int64_t)G * kh * kw * Ig_pad * Og_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx & 7);
        int64_t r = idx >> 3;
        const int o = (int)(r % Og_pad); r /= Og_pad;
        const int half = (int)(r & 1); r >>= 1;
        const int cc = (int)(r % (Ig_pad / 16)); r /= (Ig_pad / 16);
        const int t = (int)(r % (kh * kw));
        const int g = (int)(r / (kh * kw));
        const int i = cc * 16 + half * 8 + j;
        float v = 0.f;
        if (i < Ig && o < Og) {
            int ty = t / kw, tx = t - ty * kw;
            if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
            const int64_t src = transposed ? (((int64_t)(g * Ig + i) * Og + o) * kh + ty) * kw + tx
                                           : (((int64_t)(g * Og + o) * Ig + i) * kh + ty) * kw + tx;
            v = w[src] * wscale;
        }
        __bf16 p1, p2, p3;
        split3(v, p1, p2, p3);
        const int64_t chunk = (((int64_t)g * kh * kw + t) * (Ig_pad / 16) + cc) * 6 * Og_pad * 8;
        const int64_t within = ((int64_t)half * Og_pad + o) * 8 + j;
        wp[chunk + within] = p1;
        wp[chunk + 2 * Og_pad * 8 + within] = p2;
        wp[chunk + 4 * Og_pad * 8 + within] = p3;
    }
}

template <int BM, int BN, int OCC>      // (128, 128): waves 2 x 2;  (64, 256): waves 1 x 4; each wave 64 rows x 64 pixels
__global__ __launch_bounds__(256, OCC) void conv_fwd_bf16x6_kernel(ConvFwdParams p) {
    constexpr int WMT = 2, WNT = 2, KC = 16;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "four waves per workgroup");
    constexpr int ASEG = BM * 8, BSEG = BN * 8;         // bf16 elements of one (piece, half) segment
    constexpr int AUNITS = 6 * BM;                      // sixteen-byte units of the A chunk
    constexpr int APT = (AUNITS + 255) / 256;           // per thread: 3 (BM 128) or 2 (BM 64, second one guarded)
    constexpr int BPT = BN * 2 / 256;                   // (pixel, k-half) pairs per thread: 1 or 2
    // A buffers are rounded up to APT * 256 units: every thread copies APT units without a guard (see load_chunk)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][APT * 256 * 8];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][6 * BSEG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    // Workgroup -> (lattice class, pixel tile), class-major: the grid holds gridDim.x / ncls tiles for every class.
    int cls = 0;
    unsigned tile_x = blockIdx.x;
    if (p.ncls > 1) {
        const unsigned per = gridDim.x / (unsigned)p.ncls;
        cls = (int)(blockIdx.x / per);
        tile_x = blockIdx.x - (unsigned)cls * per;
    }
    const int P = p.cls[cls].P, Q = p.cls[cls].Q, oy0 = p.cls[cls].oy0, ox0 = p.cls[cls].ox0, T = p.cls[cls].T, tap0 = p.cls[cls].tap0;
    const int64_t npix = (int64_t)p.N * P * Q;
    const int64_t pix_blk = (int64_t)tile_x * BN;
    if (pix_blk >= npix) return;                   // grid is sized for the largest class
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int chunks_all = T * NC;
    const int c_first = (int)((int64_t)chunks_all * ks / p.ksplit);
    const int nchunks = (int)((int64_t)chunks_all * (ks + 1) / p.ksplit) - c_first;

    // B staging: this thread's pixel column (fixed for the whole K loop) and its k-halves:
    //   BN 128: one half, tid >> 7 (uniform per wave);  BN 256: both halves of pixel tid.
    const int bcol = tid & (BN - 1);
    const int64_t mypix = pix_blk + bcol;
    const bool pix_ok = mypix < npix;
    int n_in = 0, py = 0, px = 0;
    if (pix_ok) {
        n_in = (int)(mypix / (P * Q));
        const int rem = (int)(mypix - (int64_t)n_in * P * Q);
        py = rem / Q; px = rem - py * Q;
    }
    // Activation addressing: byte offset = (per-thread pixel part, VGPR) + (per-wave channel part, SGPR); the host only
    // selects this kernel for tensors below 2^30 elements, so 32-bit byte offsets suffice.
    const unsigned xb_off = (unsigned)(((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW) * 4u;
    const char* const xbytes = (const char*)p.x;
    const int iy_base = py * p.isy, ix_base = px * p.isx;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int half0 = BPT == 1 ? __builtin_amdgcn_readfirstlane(tid >> 7) : 0;

    // Two register sets for the activations: while chunk c is multiplied out of LDS, chunk c+1 (already in registers)
    // is split into bf16 pieces and written to the other LDS buffer between the MFMAs, and chunk c+2 is being fetched.
    // The weights of chunk c+1 (already split, L2-resident: every workgroup reads the same ones) are fetched at the
    // start of step c and copied to LDS at its end.
    // The loop body is free of data-dependent control flow around its memory operations: every step issues the same
    // loads and stores (past the end of the K range they re-read valid addresses and the activations are zeroed), so
    // that the s_waitcnt counters the compiler derives let a fetch stay in flight for a whole step.
    struct Stage { float b[8 * BPT]; int nvalid[BPT]; };
    Stage st0, st1;
    float4 areg0, areg1, areg2;         // APT of them are used (scalars: an array here is not kept in registers)
    int ld_t = c_first / NC, ld_cc = c_first - ld_t * NC, ld_left = nchunks;
    bool ld_ok = false;
    unsigned ld_pix = xb_off;          // byte offset of this thread's tap pixel in channel 0 (a readable address also when the tap is outside)
    const __bf16* a_wt = wb;           // weight fetch position: tap slab and chunk within it
    int a_t = ld_t, a_cc = ld_cc;
    auto set_tap = [&](int t_in) {
        const int t = __builtin_amdgcn_readfirstlane(tap0 + t_in);   // the tap tables are read with scalar loads
        const int iy = iy_base + p.tap_dy[t], ix = ix_base + p.tap_dx[t];
        ld_ok = pix_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        ld_pix = ld_ok ? xb_off + (unsigned)(iy * p.W + ix) * 4u : xb_off;
    };
    if (ld_t >= T) { ld_t = T - 1; ld_cc = 0; a_t = ld_t; a_cc = 0; }      // empty K slice: nothing is accumulated, addresses stay valid
    set_tap(ld_t);
    a_wt = wb + (int64_t)p.tap_slab[tap0 + a_t] * NC * 6 * p.Og_pad * 8;
    auto load_chunk = [&](Stage& st) {
        const int cc = __builtin_amdgcn_readfirstlane(ld_cc);
        const int last = p.Ig - 1;
        const bool real = ld_left > 0;              // chunks past the end of this K slice contribute zeros
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int c0 = cc * KC + (half0 + i) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned coff = (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * 4u;    // scalar
                st.b[8 * i + j] = *(const float*)(xbytes + (ld_pix + coff));
            }
            st.nvalid[i] = (ld_ok && real) ? p.Ig - c0 : 0;     // elements j < nvalid are real
        }
        --ld_left;
        if (++ld_cc >= NC) {
            ld_cc = 0;
            if (ld_t + 1 < T) set_tap(++ld_t);
        }
    };
    auto load_a = [&]() {
        const __bf16* wt = a_wt + (int64_t)a_cc * 6 * p.Og_pad * 8;
        // 6 (piece, half) segments of BM sixteen-byte units each; this thread copies units tid, tid + 256, ...
        // (BM 64: the last 128 threads repeat unit AUNITS - 1 into the padding of the LDS buffer)
        auto unit = [&](int j) {
            int e = tid + 256 * j;
            if (256 * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;
            const int seg = e / BM, within = e - seg * BM;
            return *(const float4*)(wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8);
        };
        areg0 = unit(0);
        areg1 = unit(1);
        if (APT > 2) areg2 = unit(2);
        if (++a_cc >= NC) {
            a_cc = 0;
            if (a_t + 1 < T) a_wt = wb + (int64_t)p.tap_slab[__builtin_amdgcn_readfirstlane(tap0 + ++a_t)] * NC * 6 * p.Og_pad * 8;
        }
    };
    uint32_t q1[BPT][4], q2[BPT][4], q3[BPT][4];          // 8 bf16 per piece, packed two per dword
    // Two elements at a time: v_cvt_pk_bf16_f32 yields the packed pair, the residuals come from its halves.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_pair = [&](const Stage& st, int i, int j) {
        float v0 = st.b[8 * i + 2 * j], v1 = st.b[8 * i + 2 * j + 1];
        if (st.nvalid[i] < 8) {                  // border pixel or channel tail
            v0 = 2 * j < st.nvalid[i] ? v0 : 0.f;
            v1 = 2 * j + 1 < st.nvalid[i] ? v1 : 0.f;
        }
        f32x2 v = {v0, v1};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        q1[i][j] = w;
        // the two residual subtractions stay scalar: packed f32 VALU next to MFMAs costs more than it saves
        // (MI355X_MICROARCH.md, cycle table), and the empty asm keeps the SLP vectoriser from pairing them
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        q2[i][j] = w;
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        q3[i][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            __bf16* bd = &Bs[buf][((half0 + i) * BN + bcol) * 8];
            *(uint4*)(bd) = make_uint4(q1[i][0], q1[i][1], q1[i][2], q1[i][3]);
            *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[i][0], q2[i][1], q2[i][2], q2[i][3]);
            *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[i][0], q3[i][1], q3[i][2], q3[i][3]);
        }
    };
    auto store_a = [&](int buf) {
        *(float4*)&As[buf][tid * 8] = areg0;
        *(float4*)&As[buf][(tid + 256) * 8] = areg1;
        if (APT > 2) *(float4*)&As[buf][(tid + 512) * 8] = areg2;
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int hl = lane >> 5, jl = lane & 31;
    // Fragments of one chunk: [tile][piece], read in the order the MFMA groups consume them.
    struct Frag { bf16x8 a[WMT][3], b[WNT][3]; };
    auto read_frag = [&](Frag& f, int buf) {
#define PASTA_LDA(PC) _Pragma("unroll") for (int a = 0; a < WMT; a++) f.a[a][PC] = *(const bf16x8*)&As[buf][(((PC) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8];
#define PASTA_LDB(PC) _Pragma("unroll") for (int b = 0; b < WNT; b++) f.b[b][PC] = *(const bf16x8*)&Bs[buf][(((PC) * 2 + hl) * BN + (wn * WNT + b) * 32 + jl) * 8];
        PASTA_LDA(2) PASTA_LDB(0) PASTA_LDA(0) PASTA_LDB(2) PASTA_LDA(1) PASTA_LDB(1)
#undef PASTA_LDA
#undef PASTA_LDB
    };
    // One K chunk: 24 MFMAs in six groups of four; the staging work for the next chunk is slotted between the groups.
    auto step = [&](int buf, Stage& cur_next, Stage& fetch_into) {
        load_a();                   // weights of the next chunk first: they are waited for with the activation fetch still in flight
        load_chunk(fetch_into);
        Frag f;
        read_frag(f, buf);
#define PASTA_MM(PA, PB)                                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[a][PA], f.b[b][PB], acc[a][b], 0, 0, 0);
#define PASTA_SPLIT(J) _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(cur_next, i, J);
        // smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1
        PASTA_MM(2, 0)
        PASTA_SPLIT(0)
        PASTA_MM(0, 2)
        PASTA_SPLIT(1)
        PASTA_MM(1, 1)
        PASTA_SPLIT(2)
        PASTA_MM(1, 0)
        PASTA_SPLIT(3)
        PASTA_MM(0, 1)
        store_b(buf ^ 1); store_a(buf ^ 1);
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        __syncthreads();
    };

    load_a();
    load_chunk(st0);
#pragma unroll
    for (int i = 0; i < BPT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) split_pair(st0, i, j);
    store_b(0); store_a(0);
    load_chunk(st0);                            // chunk 1 waits in registers
    __syncthreads();
    // two chunks per trip so that the register sets swap roles without copies; an odd count runs one all-zero chunk
    for (int ch = 0; ch < nchunks; ch += 2) {
        step(0, st0, st1);                      // st0 holds chunk ch+1, chunk ch+2 is fetched into st1
        step(1, st1, st0);
    }

    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        if (pix >= npix) continue;
        const int n = (int)(pix / (P * Q));
        const int rem = (int)(pix - (int64_t)n * P * Q);
        const int pp = rem / Q, qq = rem - pp * Q;
        float* yb = (p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW : p.y) +
                    ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + (oy0 + pp * p.osy) * p.OW + ox0 + qq * p.osx;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                if (o < p.Og) {
                    float v = acc[a][b][r];
                    if (osb) v *= osb[o];
                    if (p.act && p.ksplit == 1) v = conv_epilogue(v, p.bias ? p.bias[g * p.Og + o] : 0.f, p.act, p.alpha, p.gain, p.clamp);
                    yb[(int64_t)o * OHW] = v;
                }
            }
    }
}


//------------------------------------------------------------------------------------
// Row-reuse variant of conv_fwd_bf16x6_kernel for stride-1 lattices whose taps form kh rows of three horizontally
// adjacent offsets (every 3x3 stride-1 convolution and its input gradient: 85 % of the forward-type FLOPs of the step).
// A pixel tile is R = BN / SEG row segments of SEG = min(Q, BN) consecutive pixels.  The activations of one input row
// (kernel row dy) and one 16-channel chunk are fetched, split into bf16 pieces and stored to LDS ONCE, with one halo
// pixel on either side of every segment ([piece][k-half][slot][8 bf16], slot = pixel + 2 * segment + 1), and the three
// horizontal taps read their B fragments from that image at slot offsets 0, 1, 2: per 72 MFMAs one activation fetch
// and split instead of three.  The weights are fetched per tap as in the base kernel.
// K loop: "stages" (dy, chunk) of three steps (the taps of the row).  Step 0 of a stage issues the loads of the next
// stage (halo pixels first: one wave, the waves take turns), step 1 splits and stores its main pixels, step 2 its halo
// pixels; the B
// image is double-buffered per stage, the A image per step.  Control flow around memory operations is static as in
// the base kernel.
template <int BM, int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_fwd_rows_bf16x6_kernel(ConvFwdParams p) {
    constexpr int WMT = 2, WNT = 2, KC = 16;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "four waves per workgroup");
    constexpr int AUNITS = 6 * BM;
    constexpr int APT = (AUNITS + 255) / 256;
    constexpr int BPT = BN * 2 / 256;                   // (pixel, k-half) pairs per thread: 1 or 2
    constexpr int SLOTS = BN + 16;                      // up to 8 segments with two halo slots each
    constexpr int ABUF = APT * 256 * 8, BSEG = SLOTS * 8, BBUF = 6 * BSEG;      // bf16 elements
    extern __shared__ __attribute__((aligned(16))) __bf16 rows_smem[];
    __bf16* const As = rows_smem;                       // [2][ABUF]
    __bf16* const Bs = rows_smem + 2 * ABUF;            // [2][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    const int P = p.cls[0].P, Q = p.cls[0].Q, oy0 = p.cls[0].oy0, ox0 = p.cls[0].ox0, T = p.cls[0].T;
    const int KH = T / 3;
    const int64_t pix_blk = (int64_t)blockIdx.x * BN;   // the host guarantees full tiles inside one image
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int stages_all = KH * NC;
    const int s_first = (int)((int64_t)stages_all * ks / p.ksplit);
    const int nstages = (int)((int64_t)stages_all * (ks + 1) / p.ksplit) - s_first;
    const int seg_log2 = 31 - __builtin_clz(Q < BN ? Q : BN);
    const int SEG = 1 << seg_log2, R = BN >> seg_log2;
    const int d0 = p.rows_d0;

    // main pixel of this thread (fixed): tile pixel bcol -> image (n, py, px); its slot keeps one halo slot per segment free
    const int bcol = tid & (BN - 1);
    const int n_in = (int)((pix_blk + bcol) / (P * Q));
    const int rem_in = (int)(pix_blk + bcol - (int64_t)n_in * P * Q);
    const int py = rem_in / Q, px = rem_in - py * Q;
    const int m_slot = bcol + 2 * (bcol >> seg_log2) + 1;
    const int m_cx = px + d0 + 1;
    const unsigned xb_off = (unsigned)(((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW) * 4u;
    // halo pixels (one wave per stage, in turn): BN 128: lane = segment * 4 + side * 2 + k-half; BN 256: lane = segment * 2 + side, both halves.
    // Lanes beyond the last segment repeat it (identical data to the identical slot).
    int h_r = BPT == 1 ? lane >> 2 : lane >> 1;
    h_r = h_r < R ? h_r : R - 1;
    const int h_side = BPT == 1 ? (lane >> 1) & 1 : lane & 1;
    const int h_half = BPT == 1 ? lane & 1 : 0;
    const int64_t h_pixel = pix_blk + ((int64_t)h_r << seg_log2);
    const int h_n = (int)(h_pixel / (P * Q));
    const int h_rem = (int)(h_pixel - (int64_t)h_n * P * Q);
    const int h_py = h_rem / Q, h_qs = h_rem - h_py * Q;
    const int h_slot = h_r * (SEG + 2) + (h_side ? SEG + 1 : 0);
    const int h_cx = h_side ? h_qs + SEG + d0 + 1 : h_qs + d0;
    const unsigned hb_off = (unsigned)(((int64_t)h_n * p.Cin + (int64_t)g * p.Ig) * HW) * 4u;

    const char* const xbytes = (const char*)p.x;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int half0 = BPT == 1 ? __builtin_amdgcn_readfirstlane(tid >> 7) : 0;
    const int64_t a_chunk = (int64_t)6 * p.Og_pad * 8;             // bf16 elements of one packed 16-channel chunk

    // ---- fetch state of the activations: the stage (kernel row b_dy, chunk b_cc) that the next load_b() fetches
    int b_dy = s_first / NC, b_cc = s_first - b_dy * NC, b_left = nstages;
    if (b_dy >= KH) { b_dy = KH - 1; b_cc = 0; }
    bool m_ok = false, h_ok = false;
    unsigned m_pix = xb_off, h_pix = hb_off;
    auto set_row = [&](int dyi) {
        const int dy = p.tap_dy[__builtin_amdgcn_readfirstlane(3 * dyi)];
        const int iy = py + dy, hy = h_py + dy;
        m_ok = (unsigned)iy < (unsigned)p.H && (unsigned)m_cx < (unsigned)p.W;
        m_pix = m_ok ? xb_off + (unsigned)(iy * p.W + m_cx) * 4u : xb_off;
        h_ok = (unsigned)hy < (unsigned)p.H && (unsigned)h_cx < (unsigned)p.W;
        h_pix = h_ok ? hb_off + (unsigned)(hy * p.W + h_cx) * 4u : hb_off;
    };
    set_row(b_dy);
    float mb[8 * BPT], hb[8 * BPT];
    int m_nvalid[BPT], h_nvalid[BPT];
    int h_owner = 0;                                 // the wave that stages the halo pixels of the stage in flight
    auto load_b = [&]() {
        const int cc = __builtin_amdgcn_readfirstlane(b_cc);
        const int last = p.Ig - 1;
        const bool real = b_left > 0;
        h_owner = b_left & 3;
        if (wave == h_owner) {                       // oldest loads of the step: every later wait covers them
#pragma unroll
            for (int i = 0; i < BPT; i++) {
                const int c0 = cc * KC + (h_half + i) * 8;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    hb[8 * i + j] = *(const float*)(xbytes + (h_pix + (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * 4u));
                h_nvalid[i] = (h_ok && real) ? p.Ig - c0 : 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int c0 = cc * KC + (half0 + i) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned coff = (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * 4u;    // scalar
                mb[8 * i + j] = *(const float*)(xbytes + (m_pix + coff));
            }
            m_nvalid[i] = (m_ok && real) ? p.Ig - c0 : 0;
        }
        --b_left;
        if (++b_cc >= NC) {
            b_cc = 0;
            if (b_dy + 1 < KH) set_row(++b_dy);
        }
    };
    // ---- fetch state of the weights: one step ahead of the multiplication
    int a_dy = b_dy, a_cc = b_cc;
    const __bf16* a_w0 = wb; const __bf16* a_w1 = wb; const __bf16* a_w2 = wb;
    auto set_a_row = [&](int dyi) {
        const int t = __builtin_amdgcn_readfirstlane(3 * dyi);
        a_w0 = wb + (int64_t)p.tap_slab[t] * NC * a_chunk;
        a_w1 = wb + (int64_t)p.tap_slab[t + 1] * NC * a_chunk;
        a_w2 = wb + (int64_t)p.tap_slab[t + 2] * NC * a_chunk;
    };
    set_a_row(a_dy);
    float4 areg0, areg1, areg2;
    auto load_a = [&](int tap_i) {                   // tap_i is a compile-time constant at every call
        const __bf16* wt = (tap_i == 0 ? a_w0 : tap_i == 1 ? a_w1 : a_w2) + (int64_t)a_cc * a_chunk;
        auto unit = [&](int j) {
            int e = tid + 256 * j;
            if (256 * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;
            const int seg = e / BM, within = e - seg * BM;
            return *(const float4*)(wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8);
        };
        areg0 = unit(0);
        areg1 = unit(1);
        if (APT > 2) areg2 = unit(2);
    };
    auto next_a_stage = [&]() {
        if (++a_cc >= NC) {
            a_cc = 0;
            if (a_dy + 1 < KH) set_a_row(++a_dy);
        }
    };
    auto store_a = [&](int buf) {
        __bf16* d = As + buf * ABUF;
        *(float4*)&d[tid * 8] = areg0;
        *(float4*)&d[(tid + 256) * 8] = areg1;
        if (APT > 2) *(float4*)&d[(tid + 512) * 8] = areg2;
    };

    uint32_t q1[BPT][4], q2[BPT][4], q3[BPT][4];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_pair = [&](const float* b, const int* nvalid, int i, int j) {
        float v0 = b[8 * i + 2 * j], v1 = b[8 * i + 2 * j + 1];
        if (nvalid[i] < 8) {
            v0 = 2 * j < nvalid[i] ? v0 : 0.f;
            v1 = 2 * j + 1 < nvalid[i] ? v1 : 0.f;
        }
        f32x2 v = {v0, v1};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        q1[i][j] = w;
        // the two residual subtractions stay scalar: packed f32 VALU next to MFMAs costs more than it saves
        // (MI355X_MICROARCH.md, cycle table), and the empty asm keeps the SLP vectoriser from pairing them
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        q2[i][j] = w;
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        q3[i][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    };
    auto store_q = [&](int buf, int slot, int hbase) {
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            __bf16* bd = Bs + buf * BBUF + ((hbase + i) * SLOTS + slot) * 8;
            *(uint4*)(bd) = make_uint4(q1[i][0], q1[i][1], q1[i][2], q1[i][3]);
            *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[i][0], q2[i][1], q2[i][2], q2[i][3]);
            *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[i][0], q3[i][1], q3[i][2], q3[i][3]);
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int hl = lane >> 5, jl = lane & 31;
    int fslot[WNT];                                  // slot of this lane's pixel of B fragment b, for tap offset 0
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int t = (wn * WNT + b) * 32 + jl;
        fslot[b] = t + 2 * (t >> seg_log2);
    }
    struct Frag { bf16x8 a[WMT][3], b[WNT][3]; };
    auto read_frag = [&](Frag& f, int abuf, int bbuf, int off) {
        const __bf16* A_ = As + abuf * ABUF;
        const __bf16* B_ = Bs + bbuf * BBUF;
#define PASTA_LDA(PC) _Pragma("unroll") for (int a = 0; a < WMT; a++) f.a[a][PC] = *(const bf16x8*)&A_[(((PC) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8];
#define PASTA_LDB(PC) _Pragma("unroll") for (int b = 0; b < WNT; b++) f.b[b][PC] = *(const bf16x8*)&B_[(((PC) * 2 + hl) * SLOTS + fslot[b] + off) * 8];
        PASTA_LDA(2) PASTA_LDB(0) PASTA_LDA(0) PASTA_LDB(2) PASTA_LDA(1) PASTA_LDB(1)
#undef PASTA_LDA
#undef PASTA_LDB
    };
    // One tap = one step: 24 MFMAs in six groups; TAP (0, 1, 2: position in the kernel row), ABUF_ and BBUF_ are literals.
    auto step = [&](const int TAP, const int abuf, const int bbuf) {
        if (TAP == 2) next_a_stage();
        if (TAP == 0) {
            // halo loads (wave 0) are issued inside load_b ahead of everything else of this step
            load_b();
            load_a(1);
        } else {
            load_a(TAP == 1 ? 2 : 0);
        }
        Frag f;
        read_frag(f, abuf, bbuf, p.rows_rev ? 2 - TAP : TAP);
#define PASTA_MM(PA, PB)                                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[a][PA], f.b[b][PB], acc[a][b], 0, 0, 0);
#define PASTA_SPLIT(J)                                                                                         \
        if (TAP == 1) { _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(mb, m_nvalid, i, J); }        \
        if (TAP == 2 && wave == h_owner) { _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(hb, h_nvalid, i, J); }
        PASTA_MM(2, 0)
        PASTA_SPLIT(0)
        PASTA_MM(0, 2)
        PASTA_SPLIT(1)
        PASTA_MM(1, 1)
        PASTA_SPLIT(2)
        PASTA_MM(1, 0)
        PASTA_SPLIT(3)
        PASTA_MM(0, 1)
        if (TAP == 1) store_q(bbuf ^ 1, m_slot, half0);
        if (TAP == 2 && wave == h_owner) store_q(bbuf ^ 1, h_slot, h_half);
        store_a(abuf ^ 1);
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        __syncthreads();
    };

    // prologue: stage 0 of this K slice entirely, and the weights of its first tap
    load_b();
    load_a(0);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < BPT; i++) split_pair(mb, m_nvalid, i, j);
    store_q(0, m_slot, half0);
    if (wave == h_owner) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < BPT; i++) split_pair(hb, h_nvalid, i, j);
        store_q(0, h_slot, h_half);
    }
    store_a(0);
    __syncthreads();
    // two stages (six steps) per trip: the B image alternates per stage, the A image per step; an odd stage count runs
    // one all-zero stage
    for (int s = 0; s < nstages; s += 2) {
        step(0, 0, 0); step(1, 1, 0); step(2, 0, 0);
        step(0, 1, 1); step(1, 0, 1); step(2, 1, 1);
    }

    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        const int n = (int)(pix / (P * Q));
        const int rem = (int)(pix - (int64_t)n * P * Q);
        const int pp = rem / Q, qq = rem - pp * Q;
        float* yb = (p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW : p.y) +
                    ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + (oy0 + pp * p.osy) * p.OW + ox0 + qq * p.osx;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                if (o < p.Og) {
                    float v = acc[a][b][r];
                    if (osb) v *= osb[o];
                    if (p.act && p.ksplit == 1) v = conv_epilogue(v, p.bias ? p.bias[g * p.Og + o] : 0.f, p.act, p.alpha, p.gain, p.clamp);
                    yb[(int64_t)o * OHW] = v;
                }
            }
    }
}

// Pixel tiles of the row-reuse kernel: full tiles of BN pixels made of whole row segments inside one image.
static bool rows_tile_ok(int P, int Q, int BN) {
    const int seg = Q < BN ? Q : BN;
    return Q % 32 == 0 && (seg & (seg - 1)) == 0 && BN % seg == 0 && Q % seg == 0 && ((int64_t)P * Q) % BN == 0;
}

template <int BM, int BN>
static void launch_fwd_bf16x6(const ConvFwdParams& p, hipStream_t s) {
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    int64_t tiles = 0;
    for (int c = 0; c < p.ncls; c++) {
        const int64_t t = ceil_div64((int64_t)p.N * p.cls[c].P * p.cls[c].Q, BN);
        if (t > tiles) tiles = t;
    }
    tiles *= p.ncls;
    dim3 grid((unsigned)tiles, q.o_tiles * q.ksplit, p.G);
    if (p.rows && p.ncls == 1) {
        // row-reuse kernel: full tiles made of whole row segments inside one image
        if (rows_tile_ok(p.cls[0].P, p.cls[0].Q, BN)) {
            constexpr int APT = (6 * BM + 255) / 256;
            constexpr size_t lds = (size_t)(2 * APT * 256 * 8 + 2 * 6 * (BN + 16) * 8) * sizeof(__bf16);
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)conv_fwd_rows_bf16x6_kernel<BM, BN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_set = true;
            }
            hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2>), grid, dim3(256), lds, s, q);
            return;
        }
    }
    hipLaunchKernelGGL((conv_fwd_bf16x6_kernel<BM, BN, 3>), grid, dim3(256), 0, s, q);
}

// Do the T taps at table positions [0, T) form rows of three horizontally adjacent offsets (ascending or descending)?
static bool detect_tap_rows(ConvFwdParams& p, int T) {
    p.rows = 0;
    if (T % 3 != 0 || p.isx != 1 || p.isy != 1 || p.osx != 1 || p.osy != 1) return false;
    const int step = p.tap_dx[1] - p.tap_dx[0];
    if (step != 1 && step != -1) return false;
    const int d0 = step == 1 ? p.tap_dx[0] : p.tap_dx[2];
    for (int j = 0; j < T; j += 3)
        for (int i = 0; i < 3; i++)
            if (p.tap_dy[j + i] != p.tap_dy[j] || p.tap_dx[j + i] != p.tap_dx[0] + i * step) return false;
    p.rows = 1; p.rows_d0 = d0; p.rows_rev = step == -1 ? 1 : 0;
    return true;
}

constexpr int FWD_KC = 8;
// Packed input-channel padding: a multiple of the KC of the kernel instance that will run.
static int fwd_ipad(int Ig, FwdTile t) { return (Ig <= 4 && t == T64x256) ? 4 : Ig <= 8 ? 8 : 16; }

static void dispatch_fwd(FwdTile t, const ConvFwdParams& p, hipStream_t s) {
    switch (t) {
        case T128x128:
            if (p.bf16x6) { launch_fwd_bf16x6<128, 128>(p, s); break; }
            launch_fwd<128, 128, 2, 2, FWD_KC, 4>(p, s);        // 4 waves/SIMD: 99-112 TFLOP/s vs 95-104 at 3
            break;
        case T64x256:
            if (p.bf16x6) { launch_fwd_bf16x6<64, 256>(p, s); break; }
            if (p.Ig_pad == 4) launch_fwd<64, 256, 2, 2, 4>(p, s);      // RGB stems: 4-channel K chunks
            else launch_fwd<64, 256, 2, 2, FWD_KC, 4>(p, s);
            break;
        case T32x256:  launch_fwd<32, 256, 1, 2, FWD_KC>(p, s); break;
        case T64x64:   launch_fwd<64, 64, 1, 1, FWD_KC>(p, s); break;
    }
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// y[n,c,:] = oscale[n,c] * sum_ks partial[ks][n,c,:]   (fixed order; split-K epilogue)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ y,
                                                            const float* __restrict__ oscale, int64_t numel, int ohw, int ksplit,
                                                            const float* __restrict__ bias, int cout, int act, float alpha, float gain,
                                                            float clamp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        int k = 0;
        for (; k + 4 <= ksplit; k += 4) {                       // four slices in flight, summed in slice order
            const float r0 = partial[(int64_t)k * numel + i], r1 = partial[(int64_t)(k + 1) * numel + i];
            const float r2 = partial[(int64_t)(k + 2) * numel + i], r3 = partial[(int64_t)(k + 3) * numel + i];
            v += r0; v += r1; v += r2; v += r3;
        }
        for (; k < ksplit; k++) v += partial[(int64_t)k * numel + i];
        const int64_t nc = i / ohw;
        if (oscale) v *= oscale[nc];
        if (act) v = conv_epilogue(v, bias ? bias[nc % cout] : 0.f, act, alpha, gain, clamp);
        y[i] = v;
    }
}

// K slices for launches that would leave most CUs idle (the 4..17 pixel layers: K = 9*512 against <= 4624 pixels).
static int64_t fwd_lattice_pixels(const pasta_conv_desc* d) {
    if (!d->transposed) return (int64_t)d->N * d->OH * d->OW;
    return (int64_t)d->N * ((d->OH + d->stride - 1) / d->stride) * ((d->OW + d->stride - 1) / d->stride);
}

// Which kernel a forward-type launch uses: the tile, the number of K slices, and whether the split-bf16 kernel may
// run (it also needs iscale == nullptr, known only at launch).
struct FwdPlan { FwdTile tile; int ksplit; int bf16x6; };

static FwdPlan plan_fwd(const pasta_conv_desc* d) {
    const int Og = d->C_out / d->groups, Ig = d->C_in / d->groups;
    const int64_t npix = fwd_lattice_pixels(d);
    const bool sb = d->math != PASTA_MATH_F32 && Ig >= 16 && (int64_t)d->N * d->C_in * d->H * d->W < (1ll << 30);
    FwdPlan f;
    if (Og <= 32) f.tile = T32x256;                                   // ToRGB / parsing heads: HBM-bound, few rows
    else if (npix <= 8192) f.tile = (sb && Og > 64) ? T128x128 : T64x64;     // 4..16 pixel layers: K is sliced to fill the chip
    else if (Og <= 64) f.tile = T64x256;
    else f.tile = T128x128;
    f.bf16x6 = sb && (f.tile == T128x128 || f.tile == T64x256);
    f.ksplit = 1;
    if (npix <= 8192 && f.tile != T32x256) {
        const int bm = fwd_tile_bm(f.tile), bn = f.tile == T64x64 ? 64 : 128;
        const int64_t blocks = ceil_div64(npix, bn) * ((Og + bm - 1) / bm) * d->groups;
        const int taps = d->transposed ? (d->kh * d->kw + d->stride * d->stride - 1) / (d->stride * d->stride) : d->kh * d->kw;
        const int64_t k_total = (int64_t)taps * round_up(Ig, 16);
        int64_t ks = (f.tile == T64x64 ? 768 : 512) / (blocks > 0 ? blocks : 1);
        if (ks > k_total / 64) ks = k_total / 64;                     // at least 64 channel-taps per slice
        if (ks > 32) ks = 32;
        f.ksplit = ks < 2 ? 1 : (int)ks;
    }
    return f;
}

// The four output parity classes of a stride-2 conv_transpose2d share one class-major grid on the split-bf16 kernel:
// four times the workgroups per launch (measured 0.410 -> 0.266 ms on 512->256 @32^2, 0.262 -> 0.239 ms on 512->512 @16^2).
static bool merged_classes(const pasta_conv_desc* d, bool bf16x6) {
    return d->transposed && bf16x6 && d->stride == 2 && d->OH >= 2 && d->OW >= 2;
}

//------------------------------------------------------------------------------------
// Weight gradient.
//   dW[g*Ag + a][b][r][s] = sum_{n,p,q} S[n, g*Ag + a, p, q] * L[n, g*Bg + b, p*st + r - pad_h, q*st + s - pad_w]
// Workgroup: 64 (a) x 64 (b) x TR*TS taps, over a slice of K = pixels.  K is walked in chunks of
// CHH x CW = 32 lattice pixels (CW a power of two <= 32 chosen from Q).

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

// TR x TS taps per workgroup; each wave owns WA x WB 32x32 tiles per tap; PIPE = prefetch the next chunk into
// registers behind the MFMAs (needs a halo of at most 128 positions)
template <int TR, int TS, int WA, int WB, int PIPE, int KP>   // PIPE: 0 = none, 1 = halo <= 128 positions; KP pixels per chunk
__global__ __launch_bounds__(256, PIPE ? 2 : 1) void conv_wgrad_kernel(WgradParams p) {
    constexpr int NT = TR * TS;
    constexpr int BA = 64 * WA, BB = 64 * WB;  // workgroup tile: 2 x 2 waves
    constexpr int SPITCH = KP + 1;             // odd pitch: column-of-channels reads hit 32 banks
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int CW = 1 << p.cw_log2, CHH = KP >> p.cw_log2;
    const int LWID = (CW - 1) * p.st + TS;     // halo width per chunk row
    const int LPITCH = LWID | 1;
    const int LROWS = CHH * TR;                // halo rows per channel: one set of TR rows per chunk row
    const int LCH = (LROWS * LPITCH) | 1;      // odd per-channel pitch
    float* Ss = smem;                          // [BA][SPITCH]
    float* Ls = smem + BA * SPITCH;            // [BB][LCH]

    // block coordinates
    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int tgs = bid % p.tap_groups_s; bid /= p.tap_groups_s;
    const int tgr = bid % p.tap_groups_r; bid /= p.tap_groups_r;
    const int bt = bid % p.b_tiles; bid /= p.b_tiles;
    const int at = bid % p.a_tiles; bid /= p.a_tiles;
    const int g = bid;
    const int r0 = tgr * TR, s0 = tgs * TS;
    const int a_blk = at * BA, b_blk = bt * BB;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;

    // ---- staging roles, fixed for the whole K loop.
    // S: this thread's pixel of the chunk and every 8th channel.
    constexpr int SROWS = 256 / KP;                        // channels covered by one pass of the workgroup
    constexpr int SPT = BA / SROWS;
    const int s_k = tid & (KP - 1), s_a0 = tid / KP;
    const int s_dr = s_k >> p.cw_log2, s_dq = s_k & (CW - 1);
    // L: one halo position and every `lgroups`-th channel (the host guarantees NPOS <= 256).
    const int NPOS = LROWS * LWID;
    int npos_pad = 64;
    while (npos_pad < NPOS) npos_pad <<= 1;
    const int lgroups = 256 / npos_pad;                    // 4, 2 or 1 channel groups
    const int l_pos = tid & (npos_pad - 1), l_cg = tid / npos_pad;
    const bool l_act = l_pos < NPOS;
    const int l_lr = l_act ? l_pos / LWID : 0, l_lc = l_pos - l_lr * LWID;
    const int l_cr = l_lr / TR, l_tr = l_lr - l_cr * TR;
    float* const l_dst = Ls + l_lr * LPITCH + l_lc;
    const bool l_tap_ok = l_act && r0 + l_tr < p.kh;
    constexpr int LPT = PIPE ? BB / 2 : 1;                 // prefetch registers per thread (PIPE needs lgroups >= 2)

    f32x16 acc[NT][WA][WB];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < WA; i++)
#pragma unroll
            for (int j = 0; j < WB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][i][j][r] = 0.f;

    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.ksplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.ksplit);
    const int kl = lane >> 5, jl = lane & 31;
    const float* const Sg = p.S + ((int64_t)g * p.Ag + a_blk) * PQ;
    const float* const Lg = p.L + ((int64_t)g * p.Bg + b_blk) * LHW;

    float sreg[SPT];
    float lreg[LPT];

    // Per-chunk source offsets of this thread's S pixel and L halo position (element offsets fit in 31 bits).
    auto s_source = [&](int ch, bool& ok) -> unsigned {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row = rb * CHH + s_dr, q = qb * CW + s_dq;
        ok = row < p.rows_total && q < p.Q;
        const int n = ok ? row / p.P : 0, pp = row - n * p.P;
        return (unsigned)(n * p.SC) * (unsigned)PQ + (unsigned)(pp * p.Q + q);
    };
    auto l_source = [&](int ch, bool& ok) -> unsigned {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row = rb * CHH + l_cr;
        const bool rok = l_tap_ok && row < p.rows_total;
        const int n = rok ? row / p.P : 0, pp = row - n * p.P;
        const int ly = pp * p.st + r0 + l_tr - p.pad_h, lx = qb * CW * p.st + s0 + l_lc - p.pad_w;
        ok = rok && (unsigned)ly < (unsigned)p.LH && (unsigned)lx < (unsigned)p.LW;
        return ok ? (unsigned)(n * p.LC) * (unsigned)LHW + (unsigned)(ly * p.LW + lx) : 0u;
    };

    const bool full_a = a_blk + BA <= p.Ag, full_b = b_blk + BB <= p.Bg;   // uniform: no per-channel bound checks
    auto fetch = [&](int ch) {            // global -> registers (PIPE only)
        bool ok;
        const unsigned so = s_source(ch, ok);
        {
            const float* sp = Sg + so + (unsigned)s_a0 * (unsigned)PQ;
            const unsigned step = (unsigned)SROWS * (unsigned)PQ;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                sreg[j] = (ok && (full_a || a_blk + s_a0 + SROWS * j < p.Ag)) ? *sp : 0.f;
                sp += step;
            }
        }
        const unsigned lo = l_source(ch, ok);
        {
            const float* lp = Lg + lo + (unsigned)l_cg * (unsigned)LHW;
            const unsigned step = (unsigned)lgroups * (unsigned)LHW;
#pragma unroll
            for (int j = 0; j < LPT; j++) {
                const int b = l_cg + lgroups * j;
                lreg[j] = (ok && b < BB && (full_b || b_blk + b < p.Bg)) ? *lp : 0.f;
                lp += step;
            }
        }
    };
    auto stash = [&]() {                  // registers -> LDS (PIPE only)
#pragma unroll
        for (int j = 0; j < SPT; j++) Ss[(s_a0 + SROWS * j) * SPITCH + s_k] = sreg[j];
        if (l_act) {
#pragma unroll
            for (int j = 0; j < LPT; j++) {
                const int b = l_cg + lgroups * j;
                if (b < BB) l_dst[b * LCH] = lreg[j];
            }
        }
    };
    auto stage_direct = [&](int ch) {     // global -> LDS without the register stage (!PIPE)
        bool ok;
        const unsigned so = s_source(ch, ok);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int a = s_a0 + SROWS * j;
            Ss[a * SPITCH + s_k] = (ok && a_blk + a < p.Ag) ? Sg[so + (unsigned)a * (unsigned)PQ] : 0.f;
        }
        const unsigned lo = l_source(ch, ok);
        if (l_act) {
#pragma unroll 8
            for (int b = l_cg; b < BB; b += lgroups)
                l_dst[b * LCH] = (ok && b_blk + b < p.Bg) ? Lg[lo + (unsigned)b * (unsigned)LHW] : 0.f;
        }
    };

    if (PIPE && c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        if (PIPE) stash(); else stage_direct(ch);
        __syncthreads();
        if (PIPE && ch + 1 < c_end) fetch(ch + 1);   // in flight behind the MFMAs below
        // ---- KP/2 k-steps of 2 pixels; per step WA A-fragments feed NT*WA*WB MFMAs
#pragma unroll 2
        for (int kk = 0; kk < KP / 2; kk++) {
            const int k = kk * 2 + kl;
            float af[WA];
#pragma unroll
            for (int i = 0; i < WA; i++) af[i] = Ss[((wa * WA + i) * 32 + jl) * SPITCH + k];
            const int cr = k >> p.cw_log2, cc = k & (CW - 1);
            const float* lb = Ls + cr * TR * LPITCH + cc * p.st;
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const float* lbj = lb + ((wb * WB + j) * 32 + jl) * LCH;
#pragma unroll
                for (int tr = 0; tr < TR; tr++)
#pragma unroll
                    for (int ts = 0; ts < TS; ts++) {
                        const float bf = lbj[tr * LPITCH + ts];
#pragma unroll
                        for (int i = 0; i < WA; i++)
                            acc[tr * TS + ts][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf, acc[tr * TS + ts][i][j], 0, 0, 0);
                    }
            }
        }
    }

    // ---- partial slab: [ksplit][G][kh*kw][Ag_pad][Bg_pad], b contiguous
    const int Ag_pad = p.a_tiles * BA, Bg_pad = p.b_tiles * BB;
    float* out = p.slab + ((int64_t)ks * p.G + g) * p.kh * p.kw * Ag_pad * Bg_pad;
#pragma unroll
    for (int tr = 0; tr < TR; tr++)
#pragma unroll
        for (int ts = 0; ts < TS; ts++) {
            if (r0 + tr >= p.kh || s0 + ts >= p.kw) continue;
            float* ot = out + (int64_t)((r0 + tr) * p.kw + s0 + ts) * Ag_pad * Bg_pad;
#pragma unroll
            for (int i = 0; i < WA; i++)
#pragma unroll
                for (int j = 0; j < WB; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int a = a_blk + (wa * WA + i) * 32 + acc_row(r, lane), b = b_blk + (wb * WB + j) * 32 + jl;
                        ot[(int64_t)a * Bg_pad + b] = acc[tr * TS + ts][i][j][r];
                    }
        }
}

// dW[(g*Ag + a)][b][ty][tx] = sum_ks slab[ks][g][t][a][b]   (tap index optionally mirrored)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ksplit,
                                                           int G, int Ag, int Bg, int Ag_pad, int Bg_pad, int kh, int kw,
                                                           int flip, float wscale) {
    const int KK = kh * kw;
    const int64_t total = (int64_t)G * KK * Ag * Bg;
    const int64_t slab_stride = (int64_t)G * KK * Ag_pad * Bg_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int b = (int)(idx % Bg);
        int64_t r = idx / Bg;
        const int a = (int)(r % Ag); r /= Ag;
        const int t = (int)(r % KK);
        const int g = (int)(r / KK);
        const float* src = slab + (((int64_t)g * KK + t) * Ag_pad + a) * Bg_pad + b;
        // sixteen slabs in flight per thread; four partial sums combined in a fixed order (bitwise reproducible)
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int k = 0;
        for (; k + 16 <= ksplit; k += 16) {
            float r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = src[(int64_t)(k + j) * slab_stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { v0 += r[j]; v1 += r[j + 1]; v2 += r[j + 2]; v3 += r[j + 3]; }
        }
        for (; k < ksplit; k++) v0 += src[(int64_t)k * slab_stride];
        const float v = ((v0 + v1) + (v2 + v3)) * wscale;
        int ty = t / kw, tx = t - ty * kw;
        if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
        dw[(((int64_t)(g * Ag + a) * Bg + b) * kh + ty) * kw + tx] = v;
    }
}

//------------------------------------------------------------------------------------
// Weight gradient when the input has very few channels (RGB / pose stems: 3 or 6 channels, up to 7x7):
// the (channel, tap) pairs become the GEMM's column index b' = (i*kh + r)*kw + s, so a 7x7x3 kernel fills
// 147 of 160 MFMA columns instead of 3 of 64.  conv2d, stride 1, groups 1 only.
//   dW[o][b'] = sum_{n,p,q} dy[n,o,p,q] * x[n, i, p + r - pad_h, q + s - pad_w]

struct WgradSmallParams {
    const float* S; const float* L; float* slab;
    int N, Ag, P, Q;        // S = dy: [N, Ag, P, Q]
    int Bg, LH, LW;         // L = x : [N, Bg, LH, LW]
    int kh, kw, pad_h, pad_w;
    int bprime, nb;         // Bg*kh*kw and its number of 32-column tiles (<= 5)
    int cw_log2, rows_total, qblocks, chunks_total, ksplit, a_tiles;
};

__global__ __launch_bounds__(256) void conv_wgrad_smallcin_kernel(WgradSmallParams p) {
    constexpr int KP = 32, SPITCH = KP + 1, MAXT = 3;     // each wave owns column tiles wb, wb+2, wb+4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int CW = 1 << p.cw_log2, CHH = KP >> p.cw_log2;
    const int HW_ = CW + p.kw - 1;           // halo width
    const int RH = CHH * p.kh;               // halo rows per channel: kh rows for each chunk row (rows may straddle images)
    float* Ss = smem;                        // [64][SPITCH]
    float* Ls = smem + 64 * SPITCH;          // [Bg][RH][HW_], then one zero word
    const int halo_elems = p.Bg * RH * HW_;

    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int a_blk = bid * 64;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;
    const int kl = lane >> 5, jl = lane & 31;

    // this lane's (channel, tap) column in each of its tiles -> LDS offset of its halo element for chunk pixel (0,0);
    // columns beyond Bg*kh*kw read the zero word
    int boff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int bp = (wb + 2 * t) * 32 + jl;
        if (bp < p.bprime) {
            const int i = bp / (p.kh * p.kw), rs = bp - i * p.kh * p.kw, r = rs / p.kw, sx = rs - r * p.kw;
            boff[t] = (i * RH + r) * HW_ + sx;
        } else boff[t] = -1;
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int s_k = tid & 31, s_a0 = tid >> 5;
    const int s_dr = s_k >> p.cw_log2, s_dq = s_k & (CW - 1);
    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.ksplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.ksplit);

    // Register prefetch: chunk ch+1 is fetched while chunk ch is multiplied out of LDS.  Every thread owns the same 8 S
    // elements (channel s_a0 + 8j, pixel s_k) and up to 4 halo slots of every chunk; the slot -> (channel, kernel row,
    // chunk row, halo column) decomposition is fixed, only the chunk origin moves.
    constexpr int HSLOTS = 4;                          // halo_elems <= 1024 (checked by the host)
    int h_off[HSLOTS], h_r[HSLOTS], h_cr[HSLOTS], h_hx[HSLOTS], h_i[HSLOTS];
#pragma unroll
    for (int j = 0; j < HSLOTS; j++) {
        int rem = tid + 256 * j;
        h_off[j] = rem < halo_elems ? rem : -1;
        h_hx[j] = rem % HW_; rem /= HW_;
        h_r[j] = rem % p.kh; rem /= p.kh;
        h_cr[j] = rem % CHH; h_i[j] = rem / CHH;
    }
    float sreg[8], hreg[HSLOTS];
    auto fetch = [&](int ch) {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row0 = rb * CHH, q0 = qb * CW;
        {   // S: 64 channels x 32 pixels
            const int row = row0 + s_dr, q = q0 + s_dq;
            const bool ok = row < p.rows_total && q < p.Q;
            const int n = ok ? row / p.P : 0, pp = row - n * p.P;
            const float* sp = p.S + ((int64_t)n * p.Ag + a_blk) * PQ + pp * p.Q + q;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int a = s_a0 + 8 * j;
                sreg[j] = (ok && a_blk + a < p.Ag) ? sp[(int64_t)a * PQ] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < HSLOTS; j++) {              // L halo, [i][cr*kh + r][hx]
            float v = 0.f;
            const int row = row0 + h_cr[j];
            if (h_off[j] >= 0 && row < p.rows_total) {
                const int n = row / p.P, pp = row - n * p.P;
                const int ly = pp + h_r[j] - p.pad_h, lx = q0 + h_hx[j] - p.pad_w;
                if ((unsigned)ly < (unsigned)p.LH && (unsigned)lx < (unsigned)p.LW)
                    v = p.L[((int64_t)n * p.Bg + h_i[j]) * LHW + ly * p.LW + lx];
            }
            hreg[j] = v;
        }
    };
    if (c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                                // the previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < 8; j++) Ss[(s_a0 + 8 * j) * SPITCH + s_k] = sreg[j];
#pragma unroll
        for (int j = 0; j < HSLOTS; j++)
            if (h_off[j] >= 0) Ls[h_off[j]] = hreg[j];
        if (tid == 0) Ls[halo_elems] = 0.f;
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll 4
        for (int kk = 0; kk < KP / 2; kk++) {
            const int k = kk * 2 + kl;
            const float af = Ss[(wa * 32 + jl) * SPITCH + k];
            const int koff = (k >> p.cw_log2) * p.kh * HW_ + (k & (CW - 1));
#pragma unroll
            for (int t = 0; t < MAXT; t++) {
                if (wb + 2 * t >= p.nb) continue;      // uniform per wave
                const float bf = Ls[boff[t] >= 0 ? boff[t] + koff : halo_elems];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
            }
        }
    }

    // slab [ksplit][a_pad][nb*32]
    const int bpad = p.nb * 32;
    float* out = p.slab + (int64_t)ks * p.a_tiles * 64 * bpad;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (wb + 2 * t >= p.nb) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = (wb + 2 * t) * 32 + jl;
            out[(int64_t)a * bpad + b] = acc[t][r];
        }
    }
}

// dw[o][b'] = sum_ks slab[ks][o][b']   (b' already in PyTorch's [i][r][s] order)
__global__ __launch_bounds__(256) void wgrad_smallcin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ksplit,
                                                                    int Ag, int bprime, int a_pad, int bpad, float wscale) {
    const int total = Ag * bprime;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int a = idx / bprime, b = idx - a * bprime;
        const float* src = slab + (int64_t)a * bpad + b;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;          // as wgrad_reduce_kernel: sixteen slabs in flight, fixed order
        const int64_t stride = (int64_t)a_pad * bpad;
        int k = 0;
        for (; k + 16 <= ksplit; k += 16) {
            float r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = src[(k + j) * stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { v0 += r[j]; v1 += r[j + 1]; v2 += r[j + 2]; v3 += r[j + 3]; }
        }
        for (; k < ksplit; k++) v0 += src[k * stride];
        dw[idx] = ((v0 + v1) + (v2 + v3)) * wscale;
    }
}

struct WgradSmallPlan { bool use; int nb, bprime, cw_log2, qblocks, chunks_total, ksplit, a_tiles, rows_total; int64_t slab_floats; size_t lds_bytes; };

static WgradSmallPlan plan_wgrad_small(const pasta_conv_desc* d) {
    WgradSmallPlan w; w.use = false;
    const int Ig = d->C_in / d->groups;
    if (d->transposed || d->groups != 1 || d->stride != 1 || d->flip || Ig > 8 || Ig * d->kh * d->kw > 160) return w;
    w.use = true;
    w.bprime = Ig * d->kh * d->kw; w.nb = (w.bprime + 31) / 32;
    int cw = 32, lg = 5;
    while (cw > 1 && cw / 2 >= d->OW) { cw /= 2; lg--; }
    const int chh = 32 / cw;
    w.cw_log2 = lg; w.rows_total = d->N * d->OH;
    w.qblocks = (d->OW + cw - 1) / cw;
    w.chunks_total = ((w.rows_total + chh - 1) / chh) * w.qblocks;
    w.a_tiles = (d->C_out + 63) / 64;
    int64_t ks = (1024 + w.a_tiles - 1) / w.a_tiles;        // four workgroups per CU: one chunk in flight each
    if (ks > w.chunks_total / 8) ks = w.chunks_total / 8;
    if (ks < 1) ks = 1;
    w.ksplit = (int)ks;
    w.slab_floats = (int64_t)w.ksplit * w.a_tiles * 64 * w.nb * 32;
    w.lds_bytes = (size_t)(64 * 33 + Ig * chh * d->kh * (cw + d->kw - 1) + 4) * sizeof(float);
    if (Ig * chh * d->kh * (cw + d->kw - 1) > 1024) w.use = false;      // four halo slots per thread in the kernel
    return w;
}

//------------------------------------------------------------------------------------
// Weight gradient of 3x3 / stride-1 / pad-1 convolutions on the bf16 matrix cores with split-bf16 products
// (same arithmetic as conv_fwd_bf16x6_kernel: three bf16 pieces per fp32 operand, six exact products, fp32
// accumulate).  K = pixels: a chunk is 32 consecutive pixels of one image row (Q % 32 == 0), two K steps of 16.
//   A operand: S pieces in LDS as [piece][a][32 px] (row pitch 40 bf16 = 80 B: conflict-free 16-byte reads)
//   B operand: L halo pieces as [piece][b][3 rows][40 px], halo column 0 = image column q0 - 4, so every global and
//              LDS access is 16-byte aligned; the window of tap column ts starts at halo column 3 + ts: it is cut out
//              of two aligned 16-byte blocks with v_alignbit (ts = 0, 2) or by register renaming (ts = 1).
// One wave owns a 32 x 32 (a, b) tile for all 9 taps (144 accumulator registers): 54 MFMAs per K step.

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_bf16x6_kernel(WgradParams p) {
    constexpr int SP = 40, LP = 40;                 // row pitches in bf16 elements (80 B)
    constexpr int S_PIECE = 64 * SP;                // one piece of the S tile
    constexpr int L_PIECE = 64 * 3 * LP;            // one piece of the L halo tile
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    __bf16* Ss = smem16;                            // [3][64][SP]
    __bf16* Ls = smem16 + 3 * S_PIECE;              // [3][64][3][LP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;

    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int bt = bid % p.b_tiles; bid /= p.b_tiles;
    const int at = bid % p.a_tiles; bid /= p.a_tiles;
    const int g = bid;
    const int a_blk = at * 64, b_blk = bt * 64;
    const int PQ = p.P * p.Q;
    const float* const Sg = p.S + ((int64_t)g * p.Ag + a_blk) * PQ;
    const float* const Lg = p.L + ((int64_t)g * p.Bg + b_blk) * PQ;       // LH == P, LW == Q for this kernel

    // staging roles (fixed): S unit = (channel a, group of 8 pixels); L units = (channel b, halo row, group of 8 columns)
    const int s_a = tid >> 2, s_grp = tid & 3;
    const bool s_ch_ok = a_blk + s_a < p.Ag;
    int l_b[4], l_row[4], l_grp[4];
    bool l_ch_ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int v = tid + 256 * j;                // 960 units
        l_b[j] = v / 15;
        const int rg = v - l_b[j] * 15;
        l_row[j] = rg / 5; l_grp[j] = rg - l_row[j] * 5;
        l_ch_ok[j] = v < 960 && b_blk + l_b[j] < p.Bg;
    }

    float4 sreg[2], lreg[4][2];
    unsigned vmask = 0;                              // validity of the 10 sixteen-byte halves held in registers
    auto fetch = [&](int ch) {
        const int row = ch / p.qblocks, qb = ch - row * p.qblocks;
        const int n = row / p.P, pp = row - n * p.P, q0 = qb * 32;
        vmask = 0;
        {
            const float* sp = Sg + (int64_t)n * p.SC * PQ + (int64_t)s_a * PQ + pp * p.Q + q0 + 8 * s_grp;
            if (s_ch_ok) {
                sreg[0] = *(const float4*)sp; sreg[1] = *(const float4*)(sp + 4);
                vmask |= 3u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ly = pp + l_row[j] - 1, lx = q0 - 4 + 8 * l_grp[j];
            const bool rok = l_ch_ok[j] && (unsigned)ly < (unsigned)p.P;
            const float* lp = Lg + (int64_t)n * p.LC * PQ + (int64_t)l_b[j] * PQ + ly * p.Q + lx;
            if (rok && lx >= 0 && lx + 4 <= p.Q) { lreg[j][0] = *(const float4*)lp; vmask |= 4u << (2 * j); }
            if (rok && lx + 4 >= 0 && lx + 8 <= p.Q) { lreg[j][1] = *(const float4*)(lp + 4); vmask |= 8u << (2 * j); }
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    // split 8 floats (two float4 halves, each possibly invalid -> 0) into three packed bf16x8 pieces and store them
    auto split_store = [&](float4 h0, float4 h1, bool ok0, bool ok1, __bf16* dst, int piece_stride) {
        const float vals[8] = {ok0 ? h0.x : 0.f, ok0 ? h0.y : 0.f, ok0 ? h0.z : 0.f, ok0 ? h0.w : 0.f,
                               ok1 ? h1.x : 0.f, ok1 ? h1.y : 0.f, ok1 ? h1.z : 0.f, ok1 ? h1.w : 0.f};
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x2 v = {vals[2 * j], vals[2 * j + 1]};
            uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q1[j] = w;
            v[0] -= __builtin_bit_cast(float, w << 16);
            v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
            w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q2[j] = w;
            v[0] -= __builtin_bit_cast(float, w << 16);
            v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
            q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        }
        *(uint4*)(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        *(uint4*)(dst + piece_stride) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        *(uint4*)(dst + 2 * piece_stride) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };
    auto stash = [&]() {
        split_store(sreg[0], sreg[1], vmask & 1u, vmask & 2u, Ss + s_a * SP + 8 * s_grp, S_PIECE);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (tid + 256 * j < 960)
                split_store(lreg[j][0], lreg[j][1], vmask & (4u << (2 * j)), vmask & (8u << (2 * j)),
                            Ls + (l_b[j] * 3 + l_row[j]) * LP + 8 * l_grp[j], L_PIECE);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.ksplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.ksplit);
    if (c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        stash();
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 af[3];
#pragma unroll
            for (int pc = 0; pc < 3; pc++) af[pc] = *(const bf16x8*)&Ss[pc * S_PIECE + (wa * 32 + jl) * SP + 16 * s + 8 * hl];
#pragma unroll
            for (int pb = 2; pb >= 0; pb--) {       // B pieces from the smallest to the largest
#pragma unroll
                for (int row = 0; row < 3; row++) {
                    const __bf16* lb = &Ls[pb * L_PIECE + ((wb * 32 + jl) * 3 + row) * LP + 16 * s + 8 * hl];
                    const uint4 b0 = *(const uint4*)lb, b1 = *(const uint4*)(lb + 8);
                    const uint32_t d[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    uint4 w[3];
                    w[0] = make_uint4(__builtin_amdgcn_alignbit(d[2], d[1], 16), __builtin_amdgcn_alignbit(d[3], d[2], 16),
                                      __builtin_amdgcn_alignbit(d[4], d[3], 16), __builtin_amdgcn_alignbit(d[5], d[4], 16));   // halo col 3
                    w[1] = make_uint4(d[2], d[3], d[4], d[5]);                                                                 // halo col 4
                    w[2] = make_uint4(__builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16),
                                      __builtin_amdgcn_alignbit(d[5], d[4], 16), __builtin_amdgcn_alignbit(d[6], d[5], 16));   // halo col 5
#pragma unroll
                    for (int ts = 0; ts < 3; ts++) {
                        const bf16x8 bw = __builtin_bit_cast(bf16x8, w[ts]);
                        const int tap = row * 3 + ts;
                        // a_pa * b_pb with pa + pb <= 2, smallest A piece first
#pragma unroll
                        for (int pa = 2 - pb; pa >= 0; pa--)
                            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa], bw, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
    }

    // partial slab: [ksplit][G][9][Ag_pad][Bg_pad], b contiguous (same layout as conv_wgrad_kernel)
    const int Ag_pad = p.a_tiles * 64, Bg_pad = p.b_tiles * 64;
    float* out = p.slab + ((int64_t)ks * p.G + g) * 9 * Ag_pad * Bg_pad;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* ot = out + (int64_t)t * Ag_pad * Bg_pad;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = b_blk + wb * 32 + jl;
            ot[(int64_t)a * Bg_pad + b] = acc[t][r];
        }
    }
}

//------------------------------------------------------------------------------------
// Stride-2 sibling of conv_wgrad3x3_bf16x6_kernel (3x3, stride 2, pad 0 or 1 on both axes; conv2d and, with the
// operand roles swapped, conv_transpose2d): dW[tap r,s][a][b] = sum_pix S[a][p][q] * L[b][2p + r - pad][2q + s - pad].
// K chunk = 16 consecutive S pixels of one row = one K step.  The L halo is three rows of 40 columns starting at
// column 2*q0 - 4, split once and stored as [piece][b][3 rows][40] like the stride-1 kernel; the operand of tap column
// s is every second halo element from 4 - pad + s on, gathered from three aligned 16-byte LDS reads with v_perm_b32
// (the bf16 pairs of a dword are halo columns 2i, 2i + 1: a window of even or of odd columns is the low or the high
// halves of eight consecutive dwords).  L rows are not 16-byte aligned in general (257-pixel planes), so the halo is
// fetched with dword loads.
template <int PW>
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3s2_bf16x6_kernel(WgradParams p) {
    constexpr int SP = 16, LP = 40;                 // row pitches in bf16 elements
    constexpr int S_PIECE = 64 * SP;
    constexpr int L_PIECE = 64 * 3 * LP;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    __bf16* Ss = smem16;                            // [3][64][SP]
    __bf16* Ls = smem16 + 3 * S_PIECE;              // [3][64][3][LP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;

    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int bt = bid % p.b_tiles; bid /= p.b_tiles;
    const int at = bid % p.a_tiles; bid /= p.a_tiles;
    const int g = bid;
    const int a_blk = at * 64, b_blk = bt * 64;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;
    const float* const Sg = p.S + ((int64_t)g * p.Ag + a_blk) * PQ;
    const float* const Lg = p.L + ((int64_t)g * p.Bg + b_blk) * LHW;

    // staging roles (fixed): S unit = (channel a, group of 8 pixels), threads 0..127; L units = (channel b, halo row,
    // group of 8 columns), 960 of them
    const int s_a = tid >> 1, s_grp = tid & 1;
    const bool s_on = tid < 128 && a_blk + s_a < p.Ag;
    int l_b[4], l_row[4], l_grp[4];
    bool l_ch_ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int v = tid + 256 * j;
        l_b[j] = v / 15;
        const int rg = v - l_b[j] * 15;
        l_row[j] = rg / 5; l_grp[j] = rg - l_row[j] * 5;
        l_ch_ok[j] = v < 960 && b_blk + l_b[j] < p.Bg;
    }

    float4 sreg[2];
    float lreg[4][8];
    unsigned lmask = 0;                              // validity bit of each of the 32 halo elements held in registers
    bool s_ok = false;
    auto fetch = [&](int ch) {
        const int row = ch / p.qblocks, qb = ch - row * p.qblocks;
        const int n = row / p.P, pp = row - n * p.P, q0 = qb * 16;
        s_ok = s_on;
        if (s_on) {
            const float* sp = Sg + (int64_t)n * p.SC * PQ + (int64_t)s_a * PQ + pp * p.Q + q0 + 8 * s_grp;
            sreg[0] = *(const float4*)sp; sreg[1] = *(const float4*)(sp + 4);
        }
        lmask = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ly = 2 * pp + l_row[j] - p.pad_h, lx = 2 * q0 - 4 + 8 * l_grp[j];
            const bool rok = l_ch_ok[j] && (unsigned)ly < (unsigned)p.LH;
            const float* lp = Lg + (int64_t)n * p.LC * LHW + (int64_t)l_b[j] * LHW + ly * p.LW + lx;
            int first = lx < 0 ? -lx : 0, last = p.LW - lx < 8 ? p.LW - lx : 8;
            if (!rok || last < 0) last = 0;
            if (first > last) first = last;
            const unsigned m = ((1u << last) - 1u) & ~((1u << first) - 1u);
            if (m == 0xffu) {
#pragma unroll
                for (int e = 0; e < 8; e++) lreg[j][e] = lp[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if ((m >> e) & 1u) lreg[j][e] = lp[e];
            }
            lmask |= m << (8 * j);
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_store = [&](const float* vals, __bf16* dst, int piece_stride) {
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x2 v = {vals[2 * j], vals[2 * j + 1]};
            uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q1[j] = w;
            v[0] -= __builtin_bit_cast(float, w << 16);
            v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
            w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q2[j] = w;
            v[0] -= __builtin_bit_cast(float, w << 16);
            v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
            q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        }
        *(uint4*)(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        *(uint4*)(dst + piece_stride) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        *(uint4*)(dst + 2 * piece_stride) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };
    auto stash = [&]() {
        if (tid < 128) {
            const float sv[8] = {s_ok ? sreg[0].x : 0.f, s_ok ? sreg[0].y : 0.f, s_ok ? sreg[0].z : 0.f, s_ok ? sreg[0].w : 0.f,
                                 s_ok ? sreg[1].x : 0.f, s_ok ? sreg[1].y : 0.f, s_ok ? sreg[1].z : 0.f, s_ok ? sreg[1].w : 0.f};
            split_store(sv, Ss + s_a * SP + 8 * s_grp, S_PIECE);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (tid + 256 * j < 960) {
                float lv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) lv[e] = ((lmask >> (8 * j + e)) & 1u) ? lreg[j][e] : 0.f;
                split_store(lv, Ls + (l_b[j] * 3 + l_row[j]) * LP + 8 * l_grp[j], L_PIECE);
            }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.ksplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.ksplit);
    if (c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        stash();
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
        bf16x8 af[3];
#pragma unroll
        for (int pc = 0; pc < 3; pc++) af[pc] = *(const bf16x8*)&Ss[pc * S_PIECE + (wa * 32 + jl) * SP + 8 * hl];
#pragma unroll
        for (int pb = 2; pb >= 0; pb--) {           // B pieces from the smallest to the largest
#pragma unroll
            for (int row = 0; row < 3; row++) {
                const __bf16* lb = &Ls[pb * L_PIECE + ((wb * 32 + jl) * 3 + row) * LP + 16 * hl];
                const uint4 b0 = *(const uint4*)lb, b1 = *(const uint4*)(lb + 8), b2 = *(const uint4*)(lb + 16);
                const uint32_t d[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                for (int ts = 0; ts < 3; ts++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int c0 = 4 - PW + ts, d0 = c0 >> 1;
                    const uint32_t sel = (c0 & 1) ? 0x07060302u : 0x05040100u;
                    const uint4 w = make_uint4(__builtin_amdgcn_perm(d[d0 + 1], d[d0], sel), __builtin_amdgcn_perm(d[d0 + 3], d[d0 + 2], sel),
                                               __builtin_amdgcn_perm(d[d0 + 5], d[d0 + 4], sel), __builtin_amdgcn_perm(d[d0 + 7], d[d0 + 6], sel));
                    const bf16x8 bw = __builtin_bit_cast(bf16x8, w);
                    const int tap = row * 3 + ts;
#pragma unroll
                    for (int pa = 2 - pb; pa >= 0; pa--)
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa], bw, acc[tap], 0, 0, 0);
                }
            }
        }
    }

    const int Ag_pad = p.a_tiles * 64, Bg_pad = p.b_tiles * 64;
    float* out = p.slab + ((int64_t)ks * p.G + g) * 9 * Ag_pad * Bg_pad;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* ot = out + (int64_t)t * Ag_pad * Bg_pad;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = b_blk + wb * 32 + jl;
            ot[(int64_t)a * Bg_pad + b] = acc[t][r];
        }
    }
}

struct WgradPlan {
    int TR, TS, WA, WB, pipe, npos, kp, bf16x6, tgr, tgs, a_tiles, b_tiles, cw_log2, qblocks, chunks_total, ksplit, rows_total;
    int64_t slab_floats; size_t lds_bytes;
};

static WgradPlan plan_wgrad(int N, int P, int Q, int G, int Ag, int Bg, int kh, int kw, int st) {
    WgradPlan w;
    if (kh == 3 && kw == 3) { w.TR = 3; w.TS = 3; }
    else if (kw == 7) { w.TR = 1; w.TS = 7; }
    else if (kw == 4) { w.TR = 1; w.TS = 4; }
    else { w.TR = 1; w.TS = 1; }
    // single-tap kernels carry 16 accumulator registers per tile: give each wave 2 x 2 tiles when both
    // channel counts fill a 128-wide workgroup tile
    w.WA = w.WB = (w.TR * w.TS == 1 && Ag > 64 && Bg > 64) ? 2 : 1;
    const int BA = 64 * w.WA, BB = 64 * w.WB;
    w.tgr = (kh + w.TR - 1) / w.TR; w.tgs = (kw + w.TS - 1) / w.TS;
    w.a_tiles = (Ag + BA - 1) / BA; w.b_tiles = (Bg + BB - 1) / BB;
    // chunk = KP lattice pixels (CHH rows x CW columns, CW a power of two covering Q when Q is small); halve the
    // chunk when the L halo of a 32-pixel chunk is too wide for the register-prefetch pipeline (stride 2)
    int kp = 32;
    for (;;) {
        int cw = kp, lg = kp == 32 ? 5 : 4;
        while (cw > 1 && cw / 2 >= Q) { cw /= 2; lg--; }
        const int chh = kp / cw;
        const int lwid = (cw - 1) * st + w.TS;
        w.cw_log2 = lg; w.kp = kp; w.npos = chh * w.TR * lwid;
        if (w.npos <= 128 || kp == 16) break;
        kp = 16;
    }
    const int cw = 1 << w.cw_log2, chh = w.kp >> w.cw_log2;
    w.rows_total = N * P;
    w.qblocks = (Q + cw - 1) / cw;
    w.chunks_total = ((w.rows_total + chh - 1) / chh) * w.qblocks;
    const int64_t base_blocks = (int64_t)G * w.a_tiles * w.b_tiles * w.tgr * w.tgs;
    int64_t ks = (512 + base_blocks / 2) / base_blocks;  // one full wave of workgroups at 2 per CU (register-limited)
    if (ks > w.chunks_total / 8) ks = w.chunks_total / 8; // at least eight chunks per slice
    if (ks < 1) ks = 1;
    if (ks > 1024) ks = 1024;
    w.ksplit = (int)ks;
    w.slab_floats = (int64_t)w.ksplit * G * kh * kw * w.a_tiles * BA * w.b_tiles * BB;
    const int lwid = (cw - 1) * st + w.TS, lpitch = lwid | 1, lch = (chh * w.TR * lpitch) | 1;
    w.lds_bytes = (size_t)(BA * (w.kp + 1) + BB * lch) * sizeof(float);
    w.pipe = w.npos <= 128 ? 1 : 0;
    return w;
}

//------------------------------------------------------------------------------------
// Descriptor validation shared by the entry points.

static int check_desc(const pasta_conv_desc* d, const char* who) {
    PASTA_CHECK(d, "%s: null descriptor", who);
    PASTA_CHECK(d->N >= 1 && d->C_in >= 1 && d->H >= 1 && d->W >= 1 && d->C_out >= 1 && d->OH >= 1 && d->OW >= 1,
                "%s: empty tensor in descriptor", who);
    PASTA_CHECK(d->kh >= 1 && d->kw >= 1 && d->kh * d->kw <= MAX_TAPS, "%s: kernel %dx%d unsupported (max %d taps)", who, d->kh, d->kw, MAX_TAPS);
    PASTA_CHECK(d->stride >= 1 && d->stride <= 4, "%s: stride %d unsupported", who, d->stride);
    PASTA_CHECK(d->pad_h >= 0 && d->pad_w >= 0, "%s: negative padding", who);
    PASTA_CHECK(d->math >= PASTA_MATH_DEFAULT && d->math <= PASTA_MATH_BF16X6, "%s: unknown math mode %d", who, d->math);
    PASTA_CHECK(d->groups >= 1 && d->C_in % d->groups == 0 && d->C_out % d->groups == 0, "%s: channels not divisible by groups=%d", who, d->groups);
    if (!d->transposed) {
        const int oh = (d->H + 2 * d->pad_h - d->kh) / d->stride + 1, ow = (d->W + 2 * d->pad_w - d->kw) / d->stride + 1;
        PASTA_CHECK(d->H + 2 * d->pad_h >= d->kh && d->W + 2 * d->pad_w >= d->kw && oh == d->OH && ow == d->OW,
                    "%s: conv2d output is %dx%d, descriptor says %dx%d", who, oh, ow, d->OH, d->OW);
    } else {
        const int oh = (d->H - 1) * d->stride - 2 * d->pad_h + d->kh, ow = (d->W - 1) * d->stride - 2 * d->pad_w + d->kw;
        PASTA_CHECK(d->OH >= oh && d->OH < oh + d->stride && d->OW >= ow && d->OW < ow + d->stride,
                    "%s: conv_transpose2d output %dx%d not in [%d,%d)x[%d,%d)", who, d->OH, d->OW, oh, oh + d->stride, ow, ow + d->stride);
    }
    PASTA_CHECK((int64_t)d->N * d->C_in * d->H * d->W <= INT32_MAX && (int64_t)d->N * d->C_out * d->OH * d->OW <= INT32_MAX,
                "%s: tensor too large", who);
    return 0;
}

}  // namespace pasta

//------------------------------------------------------------------------------------
// C ABI.

extern "C" int64_t pasta_conv2d_workspace(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_workspace")) return -1;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    const FwdPlan f = plan_fwd(d);
    const FwdTile t = f.tile;
    const int ks = f.ksplit;
    // packed weights: fp32 (4 B) or three bf16 pieces (6 B) per element; sized for the larger, in floats
    const int64_t pack = ((int64_t)d->groups * d->kh * d->kw * round_up(Ig, fwd_ipad(Ig, t)) * round_up(Og, fwd_tile_bm(t)) * 3 + 1) / 2;
    const int64_t partial = ks > 1 ? (int64_t)ks * d->N * d->C_out * d->OH * d->OW : 0;
    return (round_up((int)pack, 4) + partial) * (int64_t)sizeof(float);
}

extern "C" int pasta_conv2d_tile(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_tile")) return -1;
    return (int)plan_fwd(d).tile;
}

extern "C" int pasta_conv2d_plan(const pasta_conv_desc* d, int has_iscale, int* tile, int* ksplit, int* math, int* launches, int* kernel) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d_plan")) return e;
    const FwdPlan f = plan_fwd(d);
    const bool sb = f.bf16x6 && !has_iscale;
    if (tile) *tile = (int)f.tile;
    if (ksplit) *ksplit = f.ksplit;
    if (math) *math = sb ? PASTA_MATH_BF16X6 : PASTA_MATH_F32;
    if (launches) *launches = !d->transposed ? 1 : merged_classes(d, sb) ? 1 : (d->stride < d->OH ? d->stride : d->OH) * (d->stride < d->OW ? d->stride : d->OW);
    if (kernel) {
        // the lattice of a stride-1 launch is the output plane itself, its taps kh rows of kw adjacent offsets
        const bool rows = sb && d->stride == 1 && d->kw == 3 && rows_tile_ok(d->OH, d->OW, f.tile == T128x128 ? 128 : 256);
        *kernel = !sb ? 0 : rows ? 2 : 1;
    }
    return 0;
}

extern "C" int pasta_conv2d(const float* x, const float* w, float* y, const float* iscale, const float* oscale,
                            const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream) {
    return pasta_conv2d_ex(x, w, y, iscale, oscale, nullptr, d, workspace, workspace_bytes, stream);
}

extern "C" int pasta_conv2d_ex(const float* x, const float* w, float* y, const float* iscale, const float* oscale,
                               const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d")) return e;
    PASTA_CHECK(!ep || (ep->act >= 1 && ep->act <= 3), "conv2d: fused epilogue supports act 1..3 (linear, relu, lrelu), got %d", ep ? ep->act : 0);
    PASTA_CHECK(x && w && y, "conv2d: null pointer");
    const int64_t need = pasta_conv2d_workspace(d);
    PASTA_CHECK(workspace && workspace_bytes >= need, "conv2d: workspace of %lld bytes needed, %lld given", (long long)need, (long long)workspace_bytes);
    PASTA_CHECK(((uintptr_t)workspace & 15) == 0, "conv2d: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;

    ConvFwdParams p;
    p.x = x; p.y = y; p.wp = (const float*)workspace; p.iscale = iscale; p.oscale = oscale;
    p.N = d->N; p.Cin = d->C_in; p.H = d->H; p.W = d->W;
    p.Cout = d->C_out; p.OH = d->OH; p.OW = d->OW;
    p.G = d->groups; p.Ig = d->C_in / d->groups; p.Og = d->C_out / d->groups;
    const FwdPlan plan = plan_fwd(d);
    const FwdTile tile = plan.tile;
    p.Ig_pad = round_up(p.Ig, fwd_ipad(p.Ig, tile)); p.Og_pad = round_up(p.Og, fwd_tile_bm(tile));
    p.KK = d->kh * d->kw;
    p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
    p.alpha = ep ? ep->alpha : 0.f; p.gain = ep ? ep->gain : 1.f; p.clamp = ep ? ep->clamp : -1.f;
    p.ksplit = plan.ksplit;
    p.o_tiles = 1;
    p.partial = (float*)workspace + round_up((int)(((int64_t)p.G * p.KK * p.Ig_pad * p.Og_pad * 3 + 1) / 2), 4);
    p.bf16x6 = (plan.bf16x6 && !iscale) ? 1 : 0;
    p.rows = 0; p.rows_d0 = 0; p.rows_rev = 0;

    const float wscale = d->wscale == 0.f ? 1.f : d->wscale;
    {   // pack weights (times wscale)
        const int64_t total = (int64_t)p.G * p.KK * p.Ig_pad * p.Og_pad;
        int64_t blocks = ceil_div64(total, 256);
        if (blocks > 4096) blocks = 4096;
        if (p.bf16x6)
            hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, (__bf16*)workspace, p.G, p.Ig, p.Og,
                               p.Ig_pad, p.Og_pad, d->kh, d->kw, d->transposed, d->flip, wscale);
        else
            hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, (float*)workspace, p.G, p.Ig, p.Og,
                               p.Ig_pad, p.Og_pad, d->kh, d->kw, d->transposed, d->flip, wscale);
    }

    if (!d->transposed) {
        p.P = d->OH; p.Q = d->OW; p.oy0 = 0; p.ox0 = 0; p.osy = 1; p.osx = 1; p.isy = d->stride; p.isx = d->stride;
        p.T = p.KK;
        for (int r = 0; r < d->kh; r++)
            for (int c = 0; c < d->kw; c++) {
                const int t = r * d->kw + c;
                p.tap_dy[t] = r - d->pad_h; p.tap_dx[t] = c - d->pad_w; p.tap_slab[t] = t;
            }
        p.ncls = 1; p.cls[0] = {p.P, p.Q, 0, 0, p.T, 0};
        detect_tap_rows(p, p.T);
        dispatch_fwd(tile, p, s);
    } else {
        // output row oy = iy*u - pad + r.  For parity class a (oy = a + u*pp): taps r with (a + pad - r) % u == 0,
        // input row = pp + (a + pad - r)/u.
        const int u = d->stride;
        p.osy = u; p.osx = u; p.isy = 1; p.isx = 1;
        const bool merged = merged_classes(d, p.bf16x6 != 0);
        int ntap = 0;
        p.ncls = 0;
        for (int a = 0; a < u && a < d->OH; a++)
            for (int b = 0; b < u && b < d->OW; b++) {
                if (!merged) ntap = 0;
                const int tap0 = ntap;
                for (int r = 0; r < d->kh; r++) {
                    if (posmod(a + d->pad_h - r, u) != 0) continue;
                    for (int c = 0; c < d->kw; c++) {
                        if (posmod(b + d->pad_w - c, u) != 0) continue;
                        p.tap_dy[ntap] = floordiv(a + d->pad_h - r, u);
                        p.tap_dx[ntap] = floordiv(b + d->pad_w - c, u);
                        p.tap_slab[ntap] = r * d->kw + c;
                        ntap++;
                    }
                }
                if (ntap == tap0) {   // no tap reaches this class: the outputs are zero
                    return fail("conv_transpose2d: kernel %dx%d smaller than stride %d leaves empty output classes (unsupported)", d->kh, d->kw, u);
                }
                p.P = (d->OH - a + u - 1) / u; p.Q = (d->OW - b + u - 1) / u;
                p.oy0 = a; p.ox0 = b; p.T = ntap - tap0;
                if (merged) {
                    p.cls[p.ncls++] = {p.P, p.Q, a, b, p.T, tap0};
                } else {
                    p.ncls = 1; p.cls[0] = {p.P, p.Q, a, b, p.T, 0};
                    detect_tap_rows(p, p.T);
                    dispatch_fwd(tile, p, s);
                }
            }
        if (merged) dispatch_fwd(tile, p, s);
    }
    if (p.ksplit > 1) {
        const int64_t numel = (int64_t)d->N * d->C_out * d->OH * d->OW;
        int64_t blocks = ceil_div64(numel, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)p.partial, y, oscale, numel,
                           d->OH * d->OW, p.ksplit, p.bias, d->C_out, p.act, p.alpha, p.gain, p.clamp);
    }
    return launch_status("conv2d");
}

namespace pasta {
// The split-bf16 weight-gradient kernel covers 3x3, stride 1, pad 1, rows of a multiple of 32 pixels.
static bool wgrad_bf16x6(const pasta_conv_desc* d, const WgradPlan& w) {
    const int P = d->transposed ? d->H : d->OH, Q = d->transposed ? d->W : d->OW;
    const int LH = d->transposed ? d->OH : d->H, LW = d->transposed ? d->OW : d->W;
    return d->math != PASTA_MATH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           Q % 32 == 0 && LH == P && LW == Q && w.kp == 32 && w.cw_log2 == 5;
}
// ... and its stride-2 sibling: 3x3, stride 2, equal pads of 0 or 1, rows of a multiple of 16 pixels.
static bool wgrad_s2_bf16x6(const pasta_conv_desc* d, const WgradPlan& w) {
    const int Q = d->transposed ? d->W : d->OW;
    return d->math != PASTA_MATH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_h == d->pad_w && d->pad_h <= 1 &&
           Q % 16 == 0 && w.kp == 16 && w.cw_log2 == 4;
}
}  // namespace pasta

extern "C" int pasta_conv2d_wgrad_plan(const pasta_conv_desc* d, int* kernel) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d_wgrad_plan")) return e;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    int k = 0;
    if (plan_wgrad_small(d).use) k = 1;
    else {
        const WgradPlan w = d->transposed ? plan_wgrad(d->N, d->H, d->W, d->groups, Ig, Og, d->kh, d->kw, d->stride)
                                          : plan_wgrad(d->N, d->OH, d->OW, d->groups, Og, Ig, d->kh, d->kw, d->stride);
        if (wgrad_bf16x6(d, w)) k = 2;
        else if (wgrad_s2_bf16x6(d, w)) k = 3;
    }
    if (kernel) *kernel = k;
    return 0;
}

extern "C" int64_t pasta_conv2d_wgrad_workspace(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_wgrad_workspace")) return -1;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    const WgradSmallPlan ws = plan_wgrad_small(d);
    if (ws.use) return ws.slab_floats * (int64_t)sizeof(float);
    const WgradPlan w = d->transposed ? plan_wgrad(d->N, d->H, d->W, d->groups, Ig, Og, d->kh, d->kw, d->stride)
                                      : plan_wgrad(d->N, d->OH, d->OW, d->groups, Og, Ig, d->kh, d->kw, d->stride);
    return w.slab_floats * (int64_t)sizeof(float);
}

extern "C" int pasta_conv2d_wgrad(const float* x, const float* dy, float* dw, const pasta_conv_desc* d, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d_wgrad")) return e;
    PASTA_CHECK(x && dy && dw, "conv2d_wgrad: null pointer");
    const int64_t need = pasta_conv2d_wgrad_workspace(d);
    PASTA_CHECK(workspace && workspace_bytes >= need, "conv2d_wgrad: workspace of %lld bytes needed, %lld given", (long long)need, (long long)workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;

    const WgradSmallPlan ws = plan_wgrad_small(d);
    if (ws.use) {
        WgradSmallParams q;
        q.S = dy; q.L = x; q.slab = (float*)workspace;
        q.N = d->N; q.Ag = d->C_out; q.P = d->OH; q.Q = d->OW; q.Bg = Ig; q.LH = d->H; q.LW = d->W;
        q.kh = d->kh; q.kw = d->kw; q.pad_h = d->pad_h; q.pad_w = d->pad_w;
        q.bprime = ws.bprime; q.nb = ws.nb; q.cw_log2 = ws.cw_log2; q.rows_total = ws.rows_total; q.qblocks = ws.qblocks;
        q.chunks_total = ws.chunks_total; q.ksplit = ws.ksplit; q.a_tiles = ws.a_tiles;
        PASTA_CHECK(ws.lds_bytes <= 64 * 1024, "conv2d_wgrad: small-cin LDS footprint %zu too large", ws.lds_bytes);
        hipLaunchKernelGGL(conv_wgrad_smallcin_kernel, dim3((unsigned)(ws.a_tiles * ws.ksplit)), dim3(256), ws.lds_bytes, s, q);
        const int total = d->C_out * ws.bprime;
        hipLaunchKernelGGL(wgrad_smallcin_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)workspace, dw,
                           ws.ksplit, d->C_out, ws.bprime, ws.a_tiles * 64, ws.nb * 32, d->wscale == 0.f ? 1.f : d->wscale);
        return launch_status("conv2d_wgrad(small-cin)");
    }

    WgradParams p;
    p.slab = (float*)workspace;
    p.G = d->groups; p.kh = d->kh; p.kw = d->kw; p.st = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.N = d->N;
    if (!d->transposed) {   // dw[o][i]: S = dy, L = x
        p.S = dy; p.SC = d->C_out; p.P = d->OH; p.Q = d->OW; p.Ag = Og;
        p.L = x;  p.LC = d->C_in;  p.LH = d->H; p.LW = d->W; p.Bg = Ig;
    } else {                // dw[i][o]: S = x, L = dy
        p.S = x;  p.SC = d->C_in;  p.P = d->H; p.Q = d->W; p.Ag = Ig;
        p.L = dy; p.LC = d->C_out; p.LH = d->OH; p.LW = d->OW; p.Bg = Og;
    }
    const WgradPlan w = plan_wgrad(p.N, p.P, p.Q, p.G, p.Ag, p.Bg, p.kh, p.kw, p.st);
    p.cw_log2 = w.cw_log2; p.rows_total = w.rows_total; p.qblocks = w.qblocks; p.chunks_total = w.chunks_total;
    p.ksplit = w.ksplit; p.a_tiles = w.a_tiles; p.b_tiles = w.b_tiles; p.tap_groups_r = w.tgr; p.tap_groups_s = w.tgs;
    PASTA_CHECK(w.lds_bytes <= 160 * 1024, "conv2d_wgrad: LDS footprint %zu too large", w.lds_bytes);
    PASTA_CHECK(w.npos <= 256, "conv2d_wgrad: halo of %d positions per chunk is not supported", w.npos);

    const int64_t blocks = (int64_t)p.G * w.a_tiles * w.b_tiles * w.tgr * w.tgs * w.ksplit;
    PASTA_CHECK(blocks <= INT32_MAX, "conv2d_wgrad: grid too large");
#define PASTA_WGRAD1(TR_, TS_, WA_, WB_, PIPE_, KP_)                                                                      \
    do {                                                                                                                  \
        if (w.lds_bytes > 64 * 1024)                                                                                      \
            PASTA_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_kernel<TR_, TS_, WA_, WB_, PIPE_, KP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w.lds_bytes)); \
        hipLaunchKernelGGL((conv_wgrad_kernel<TR_, TS_, WA_, WB_, PIPE_, KP_>), dim3((unsigned)blocks), dim3(256), w.lds_bytes, s, p); \
    } while (0)
#define PASTA_WGRAD(TR_, TS_, WA_, WB_)                                                                                   \
    do {                                                                                                                  \
        if (w.kp == 16) { if (w.pipe) PASTA_WGRAD1(TR_, TS_, WA_, WB_, 1, 16); else PASTA_WGRAD1(TR_, TS_, WA_, WB_, 0, 16); } \
        else            { if (w.pipe) PASTA_WGRAD1(TR_, TS_, WA_, WB_, 1, 32); else PASTA_WGRAD1(TR_, TS_, WA_, WB_, 0, 32); } \
    } while (0)
    if (wgrad_bf16x6(d, w)) {
        const size_t lds = (size_t)(3 * 64 * 40 + 3 * 64 * 3 * 40) * 2;
        hipLaunchKernelGGL(conv_wgrad3x3_bf16x6_kernel, dim3((unsigned)blocks), dim3(256), lds, s, p);
    }
    else if (wgrad_s2_bf16x6(d, w)) {
        const size_t lds = (size_t)(3 * 64 * 16 + 3 * 64 * 3 * 40) * 2;
        if (d->pad_w == 1) hipLaunchKernelGGL(conv_wgrad3x3s2_bf16x6_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, s, p);
        else               hipLaunchKernelGGL(conv_wgrad3x3s2_bf16x6_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, s, p);
    }
    else if (w.TR == 3 && w.TS == 3) PASTA_WGRAD(3, 3, 1, 1);
    else if (w.TS == 7) PASTA_WGRAD(1, 7, 1, 1);
    else if (w.TS == 4) PASTA_WGRAD(1, 4, 1, 1);
    else if (w.WA == 2) PASTA_WGRAD(1, 1, 2, 2);
    else PASTA_WGRAD(1, 1, 1, 1);
#undef PASTA_WGRAD
#undef PASTA_WGRAD1
    {
        const int64_t total = (int64_t)p.G * p.kh * p.kw * p.Ag * p.Bg;
        int64_t rb = ceil_div64(total, 256);
        if (rb > 8192) rb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const float*)workspace, dw, w.ksplit, p.G,
                           p.Ag, p.Bg, w.a_tiles * 64 * w.WA, w.b_tiles * 64 * w.WB, p.kh, p.kw, d->flip, d->wscale == 0.f ? 1.f : d->wscale);
    }
    return launch_status("conv2d_wgrad");
}
