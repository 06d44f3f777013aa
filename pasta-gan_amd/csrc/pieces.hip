// Producer-written operand pieces (round 5): the 4x4 low-pass in front of a stride-2 convolution (conv2d_resample.py:119-122) writes its
// output ONCE, as the matrix-core operand of the three-product fp16 arithmetic (conv_common.h, PASTA_MATH_F16X3), instead of as an fp32 NCHW
// tensor that each consumer launch -- the stride-2 forward convolution and its weight gradient -- fetches with eight channel-strided dword
// loads per staging unit and splits with ~20 VALU instructions per unit, launch after launch.
//
// Layout PASTA_LAYOUT_PIECES16 of a logical [N, C, H, W] tensor (C a multiple of 8):
//     unit (n, c / 8, y, piece, x) = 16 bytes: eight fp16 values, one per channel 8 (c / 8) .. + 7:  piece 0: h = fp16(v S),  piece 1: l' = fp16(2^11 (v S - h))
//     units in the order [N][C / 8][H][2][W]: a row of one piece is 16 W contiguous bytes -- what a wave stores (producer) or fetches (consumers)
//     with one sixteen-byte access per lane is one contiguous run -- and every unit is 16-byte aligned whatever W is (the fp32 planes of 257
//     columns are not even 16-byte aligned from row to row).  (First version: h | l' interleaved per pixel, 32-byte units -- every store
//     instruction of a wave then filled half of each 32-byte sector it touched, and the blur ran no faster than the fp32 one.)
// 4 bytes per logical element, like fp32.  S is the power of two that scale_from_amax() takes from a BOUND of the output's magnitude that
// is known before the blur starts: the 256 partial maxima of the blur's INPUT (the producer row its writer left, or one scan) times
// gain * sum |f| (1 for the normalised low-pass of the networks).  The kernel leaves that bound as a 256-float row of its own (y_amax), and
// the consumer, handed the row, derives the same S (a maximum commutes with the multiplication by a positive constant, bit for bit).
// Because S is a power of two, v S, h and l' scale exactly with it: the consumer's result does not depend on WHICH admissible S was used
// (tests/test_pieces_gpu.py halves and doubles it), as long as nothing overflows (S too large) or leaves fp16's normal range (S too small).
//
// Kernel: one workgroup = 8 channels x (8 rows x 64 columns) of outputs.  The 8 x 11 x 67 input footprint is staged in LDS with row-coalesced
// loads, a thread computes two rows of one column for the eight channels (sixteen 16-tap sums, the taps in the order of upfirdn2d_tile_kernel:
// the fp32 value in front of the split is bit-identical to pasta_upfirdn2d's) and stores four 16-byte units (two rows x two pieces): a wave writes 1 KB runs.
// Planes of 64 k + 1 columns (257, 129, 65: every live shape) are covered by k tile columns whose last thread column computes the extra one.
#include "conv_common.h"

namespace pasta {

struct BlurPiecesParams {
    const float* x; const float* f; void* pieces; const float* parts; float* bound;
    int N, C, H, W, OH, OW, padx0, pady0, flip;
    float gain;
};

template <int TOW, int TOH>
__global__ __launch_bounds__(256) void blur_pieces_kernel(BlurPiecesParams p, int tiles_x, int tiles_y, int rem_x) {
    static_assert(TOW == 64 && TOH == 8, "thread map: 64 columns x 4 row pairs");
    constexpr int TIW = TOW + 4, TIH = TOH + 3, LDW = TIW | 1;         // one more column for the remainder column of the last tile
    __shared__ f32x2_t sx[4][TIH][LDW];                // [channel pair][row][column]: the pair (2 q, 2 q + 1) is one 8-byte read and one packed operand
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    int b = blockIdx.x;
    const int tile_x = b % tiles_x; b /= tiles_x;
    const int tile_y = b % tiles_y; b /= tiles_y;
    const int C8 = p.C >> 3;
    const int c8 = b % C8, n = b / C8;

    float g[4][4];
    float fsum = 0.f;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < 4; c++) { g[a][c] = p.f[(p.flip ? a : 3 - a) * 4 + (p.flip ? c : 3 - c)]; fsum += fabsf(p.f[a * 4 + c]); }
    const float bmul = fsum * fabsf(p.gain);          // |y| <= bmul max |x|
    float sc, isc;
    scale_from_amax(amax_of_parts(p.parts) * bmul, sc, isc);
    if (blockIdx.x == 0) p.bound[tid] = p.parts[tid] * bmul;         // the row the consumers take the same S from

    const int ix0 = tile_x * TOW - p.padx0, iy0 = tile_y * TOH - p.pady0;
    const float* const xp = p.x + ((int64_t)n * p.C + (int64_t)c8 * 8) * p.H * p.W;
    // Staging: wave w fetches channels 2 w and 2 w + 1 -- row and channel of every load are wave-uniform (scalar address arithmetic, no division
    // per slot: the first version decoded 24 flat slots per thread with two divisions each and spent more instructions there than on the taps),
    // lane = column; the four columns beyond the 64th by the first four lanes.  Every load of the thread in flight before the first LDS store.
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x2_t stage[TIH], extra[TIH];
#pragma unroll
    for (int row = 0; row < TIH; row++) {
        const int iy = iy0 + row, ix = ix0 + lane;
        const bool rok = (unsigned)iy < (unsigned)p.H;
        const bool in = rok && (unsigned)ix < (unsigned)p.W, in2 = rok && lane < TIW - 64 && (unsigned)(ix + 64) < (unsigned)p.W;
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const float* const rp = xp + ((int64_t)(2 * wave + c2) * p.H + (rok ? iy : 0)) * p.W;
            const float v = rp[ix < 0 ? 0 : ix < p.W ? ix : p.W - 1];      // always a valid address, no branch around the load; zeroed below
            stage[row][c2] = in ? v : 0.f;
            extra[row][c2] = in2 ? rp[ix + 64] : 0.f;
        }
    }
#pragma unroll
    for (int row = 0; row < TIH; row++) {
        sx[wave][row][lane] = stage[row];
        if (lane < TIW - 64) sx[wave][row][64 + lane] = extra[row];
    }
    __syncthreads();

    char* const out = (char*)p.pieces + (((int64_t)n * C8 + c8) * p.OH) * (int64_t)p.OW * 32;        // rows of this octet: [OH][2 pieces][OW] units of 16 bytes
    auto one = [&](int yy, int xx) {                    // output (yy, xx) relative to the tile origin, eight channels -> the h unit and the l' unit
        const int oy = tile_y * TOH + yy, ox = tile_x * TOW + xx;
        if (oy >= p.OH || ox >= p.OW) return;
        uint32_t h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x2_t a = {0.f, 0.f};                     // channels 2 q, 2 q + 1: sixteen packed FMAs, the taps in pasta_upfirdn2d's order
#pragma unroll
            for (int jy = 0; jy < 4; jy++)
#pragma unroll
                for (int jx = 0; jx < 4; jx++) a = __builtin_elementwise_fma(f32x2_t{g[jy][jx], g[jy][jx]}, sx[q][yy + jy][xx + jx], a);
            a *= p.gain;
            f16_split2(a[0] * sc, a[1] * sc, h[q], l[q]);
        }
        u32x4* const d = (u32x4*)(out + ((int64_t)oy * 2 * p.OW + ox) * 16);
        d[0] = u32x4{h[0], h[1], h[2], h[3]};
        d[p.OW] = u32x4{l[0], l[1], l[2], l[3]};
    };
    one(2 * ty, tx);
    one(2 * ty + 1, tx);
    if (rem_x && tile_x == tiles_x - 1 && tx == TOW - 1) {
        one(2 * ty, TOW);
        one(2 * ty + 1, TOW);
    }
}

// v = (h + 2^-11 l') / S per element, back to fp32 NCHW (tests, diagnostics; the operand's 22 bits, not the blur's 24)
__global__ __launch_bounds__(256) void pieces_unpack_kernel(const void* pieces, const float* parts, float* y, int N, int C, int H, int W) {
    float sc, isc;
    scale_from_amax(amax_of_parts(parts), sc, isc);
    const int64_t units = (int64_t)N * (C >> 3) * H * W;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int64_t x = u % W, r = u / W;
        const u32x4 hq = ((const u32x4*)pieces)[2 * r * W + x], lq = ((const u32x4*)pieces)[(2 * r + 1) * W + x];      // r = (n, octet, row)
        const f16x8 h = __builtin_bit_cast(f16x8, hq), l = __builtin_bit_cast(f16x8, lq);
        const int64_t yy = r % H, r2 = r / H;
        const int64_t c8 = r2 % (C >> 3), n = r2 / (C >> 3);
#pragma unroll
        for (int j = 0; j < 8; j++)
            y[((n * C + c8 * 8 + j) * H + yy) * W + x] = ((float)h[j] + (float)l[j] * (1.f / 2048.f)) * isc;
    }
}

// The reverse: an fp32 NCHW tensor whose partial maxima are known, split as the consuming kernels split it (f16_split2 under the same power-of-two
// scale: the pieces a kernel would have formed in its staging, bit for bit).  For tests and measurements of the pieces-reading kernels, and for a
// caller whose producer is not one of this library's kernels.
__global__ __launch_bounds__(256) void pieces_pack_kernel(const float* x, const float* parts, void* pieces, int N, int C, int H, int W) {
    float sc, isc;
    scale_from_amax(amax_of_parts(parts), sc, isc);
    const int64_t units = (int64_t)N * (C >> 3) * H * W;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int64_t xx = u % W, r = u / W;
        const int64_t yy = r % H, r2 = r / H;
        const int64_t c8 = r2 % (C >> 3), n = r2 / (C >> 3);
        u32x4 hq, lq;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v0 = x[((n * C + c8 * 8 + 2 * j) * H + yy) * W + xx], v1 = x[((n * C + c8 * 8 + 2 * j + 1) * H + yy) * W + xx];
            uint32_t a, b;
            f16_split2(v0 * sc, v1 * sc, a, b);
            hq[j] = a; lq[j] = b;
        }
        ((u32x4*)pieces)[2 * r * W + xx] = hq;
        ((u32x4*)pieces)[(2 * r + 1) * W + xx] = lq;
    }
}

}  // namespace pasta

extern "C" int64_t pasta_pieces_bytes(int N, int C, int H, int W) {
    if (N < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return -1;
    return (int64_t)N * (C >> 3) * H * W * 32;
}

extern "C" int pasta_blur_pieces(const float* x, const float* f, void* pieces, const float* x_amax, float* y_amax, int N, int C, int H, int W,
                                 int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream) {
    using namespace pasta;
    PASTA_CHECK(x && f && pieces && x_amax && y_amax, "blur_pieces: null pointer (the partial maxima of x are required: they fix the operand scale)");
    PASTA_CHECK(N >= 1 && C >= 8 && (C & 7) == 0 && H >= 1 && W >= 1, "blur_pieces: [%d, %d, %d, %d]: the channel count must be a multiple of 8", N, C, H, W);
    BlurPiecesParams p;
    p.x = x; p.f = f; p.pieces = pieces; p.parts = x_amax; p.bound = y_amax;
    p.N = N; p.C = C; p.H = H; p.W = W;
    p.OW = W + padx0 + padx1 - 3; p.OH = H + pady0 + pady1 - 3;
    PASTA_CHECK(p.OW >= 1 && p.OH >= 1, "blur_pieces: output must be at least 1x1");
    PASTA_CHECK((int64_t)N * C * H * W <= INT32_MAX && (int64_t)N * C * p.OH * p.OW <= INT32_MAX, "blur_pieces: tensor too large");
    p.padx0 = padx0; p.pady0 = pady0; p.flip = flip ? 1 : 0; p.gain = gain;
    constexpr int TOW = 64, TOH = 8;
    int tiles_x = (p.OW + TOW - 1) / TOW;
    const int rem_x = (p.OW > TOW && p.OW % TOW == 1) ? 1 : 0;
    tiles_x -= rem_x;
    const int tiles_y = (p.OH + TOH - 1) / TOH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * (C >> 3) * N;
    PASTA_CHECK(blocks <= INT32_MAX, "blur_pieces: grid too large");
    hipLaunchKernelGGL((blur_pieces_kernel<TOW, TOH>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, tiles_x, tiles_y, rem_x);
    return launch_status("blur_pieces");
}

extern "C" int pasta_pieces_unpack(const void* pieces, const float* x_amax, float* y, int N, int C, int H, int W, void* stream) {
    using namespace pasta;
    PASTA_CHECK(pieces && x_amax && y, "pieces_unpack: null pointer");
    PASTA_CHECK(N >= 1 && C >= 8 && (C & 7) == 0 && H >= 1 && W >= 1, "pieces_unpack: [%d, %d, %d, %d]: the channel count must be a multiple of 8", N, C, H, W);
    const int64_t units = (int64_t)N * (C >> 3) * H * W;
    int64_t blocks = ceil_div64(units, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pieces_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pieces, x_amax, y, N, C, H, W);
    return launch_status("pieces_unpack");
}

extern "C" int pasta_pieces_pack(const float* x, const float* x_amax, void* pieces, int N, int C, int H, int W, void* stream) {
    using namespace pasta;
    PASTA_CHECK(x && x_amax && pieces, "pieces_pack: null pointer (the partial maxima of x fix the operand scale)");
    PASTA_CHECK(N >= 1 && C >= 8 && (C & 7) == 0 && H >= 1 && W >= 1, "pieces_pack: [%d, %d, %d, %d]: the channel count must be a multiple of 8", N, C, H, W);
    const int64_t units = (int64_t)N * (C >> 3) * H * W;
    int64_t blocks = ceil_div64(units, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pieces_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_amax, pieces, N, C, H, W);
    return launch_status("pieces_pack");
}
