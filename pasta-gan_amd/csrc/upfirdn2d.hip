// upfirdn2d for gfx950: pad -> zero-stuff upsample -> 2-D FIR -> decimate.
//
// Semantics follow the reference specification torch_utils/ops/upfirdn2d.py:169-208
// (output size: torch_utils/ops/upfirdn2d.cpp:32-33).  In one dimension:
//     y[o] = gain * sum_{t=0}^{F-1} g[t] * Z[o*D + t - pad0]
//     Z[u] = x[u/U] if u % U == 0 and 0 <= u/U < W, else 0
//     g[t] = f[F-1-t]   (flip == 0, true convolution)   or   f[t]  (flip == 1)
//
// Three kernels:
//   * tile kernel  -- NCHW-contiguous planes, compile-time (U, D, F, phase):
//     a 256-thread workgroup stages the input footprint of one output tile in LDS
//     with row-coalesced loads, every thread produces an MX x MY micro-tile whose
//     tap positions are all compile-time constants, filter taps live in SGPRs.
//     HBM traffic = read each input once (+ halo) and write each output once.
//     Planes of k * tile + 1 columns / rows (257, 129, 65, 33: every blur in front of a stride-2 convolution) are
//     covered by k tiles whose last thread column / row computes the extra output (REM).
//   * small-plane kernel -- the 4..16-pixel layers: several whole planes per workgroup through LDS.
//   * generic kernel -- any strides / factors / filter size / dtype, one output per
//     thread, no LDS (correctness net for every configuration the API admits).
// Measured and dropped (round 2, [16,64,256,256] -> 257 x 257, 176 us with the tile kernel): an XCD-aware tile order
// that keeps the tiles of one tile row on one XCD (195 us, and 15 % slower on every aligned shape), and a full-width
// strip kernel whose wave stores are 256-byte aligned in the output address (235 us: sixteen LDS reads per output and no
// load / compute overlap cost more than the aligned stores return).  What the odd plane width costs is in the stores:
// the same filter into 256-column rows takes 136 us, an odd number of rows costs nothing.
// Round 3, measured and dropped again: full-width strips of 16 rows whose loads AND stores are 16-byte packs aligned to the address
// (a strip of whole rows is one contiguous block on both sides, whatever the row length), (row, column) of a pack's elements by
// reciprocal multiplication, four outputs of a row sharing a 4 x 7 LDS window: 199 us (257 -> 256) and 234 us (256 -> 257) against
// 148 / 209 for the tile kernel on the same box -- the address arithmetic per element and two barriers per 4112 outputs cost more than
// the partial cache lines.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace pasta {

struct UpfirdnParams {
    const void*  x;
    const float* f;
    void*        y;
    int inW, inH, C, N;
    int outW, outH;
    int64_t isx, isy, isc, isn;   // input element strides
    int64_t osx, osy, osc, osn;   // output element strides
    int fw, fh;
    int upx, upy, downx, downy;
    int padx0, pady0;
    int flip;
    float gain;
    float* y_amax;                // optional: PASTA_AMAX_PARTS zeroed floats that receive the largest finite |y| (common.h)
    const void* y_add;            // optional: a tensor laid out like y, added to the result on its way out (ABI 18: another consumer's gradient
                                  // when this launch is the backward of a tensor with several consumers)
};

//------------------------------------------------------------------------------------
// Generic kernel.

template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(UpfirdnParams p) {
    typedef typename acc_of<T>::type A;
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int ox = (int)(idx % p.outW);
        int64_t r = idx / p.outW;
        int oy = (int)(r % p.outH);
        r /= p.outH;
        int c = (int)(r % p.C);
        int n = (int)(r / p.C);

        const int ux0 = ox * p.downx - p.padx0;   // upsampled coordinate of tap 0
        const int uy0 = oy * p.downy - p.pady0;
        const int tx0 = posmod(-ux0, p.upx);
        const int ty0 = posmod(-uy0, p.upy);
        const T* xp = (const T*)p.x + n * p.isn + c * p.isc;

        A v = 0;
        for (int ty = ty0; ty < p.fh; ty += p.upy) {
            int iy = (uy0 + ty) / p.upy;   // exact division
            if (uy0 + ty < 0) continue;
            if (iy >= p.inH) break;
            int gy = p.flip ? ty : p.fh - 1 - ty;
            for (int tx = tx0; tx < p.fw; tx += p.upx) {
                int ix = (ux0 + tx) / p.upx;
                if (ux0 + tx < 0) continue;
                if (ix >= p.inW) break;
                int gx = p.flip ? tx : p.fw - 1 - tx;
                v += ld<T>(xp + iy * p.isy + ix * p.isx) * (A)p.f[gy * p.fw + gx];
            }
        }
        v *= (A)p.gain;
        if (p.y_add) v += (A)ld<T>((const T*)p.y_add + n * p.osn + c * p.osc + oy * p.osy + ox * p.osx);
        st<T>((T*)p.y + n * p.osn + c * p.osc + oy * p.osy + ox * p.osx, v);
        if (p.y_amax) amax_take(am, (float)v);
    }
    amax_commit(am, aslot);
}

//------------------------------------------------------------------------------------
// Tile kernel.

__host__ __device__ constexpr int c_posmod(int a, int b) { return ((a % b) + b) % b; }
// first tap index hit by output a of a micro-tile (phase PH = pad0 mod U)
__host__ __device__ constexpr int c_tap0(int a, int D, int U, int PH) { return c_posmod(PH - a * D, U); }
// input offset (relative to the micro-tile base) read by tap 0 of output a
__host__ __device__ constexpr int c_in0(int a, int D, int U, int PH) { return (a * D + c_tap0(a, D, U, PH) - PH) / U; }
// number of taps of output a
__host__ __device__ constexpr int c_ntaps(int a, int D, int U, int PH, int F) { return (F - c_tap0(a, D, U, PH) + U - 1) / U; }

// REM = 1 (1:1 kernels only): an output plane of k * TOW + 1 columns (k * TOH + 1 rows) is covered by k tiles per row
// (column); the last tile's last thread column (row) computes the one extra output column (row).  Every blur behind a
// pad-2 convolution has such planes (257, 129, 65, 33): without this a fifth 64-column tile runs 1/64 used.
// PL (round 5) = planes per work item: 2 (fp32, an even number of planes) = two consecutive planes at once, interleaved in LDS as float2, so that
// a tap is ONE 8-byte read feeding ONE v_pk_fma_f32 for the two planes.  The round's ablation of the blur in front of the stride-2 convolutions
// (pieces.hip; profiles/r5_down_path_microbench.txt) showed these kernels bound by instruction issue, not by memory: with one plane per item the
// compiler packs pairs of OUTPUTS into v_pk_fma_f32 and spends two v_mov per packed operand (85 v_mov for 32 packed FMAs in the 1:1 4 x 4 instance).
typedef float up_f32x2 __attribute__((ext_vector_type(2)));
template <int PL> struct UpVec { typedef float type; };
template <> struct UpVec<2> { typedef up_f32x2 type; };
__device__ __forceinline__ float up_fma(float g, float x, float v) { return fmaf(g, x, v); }
__device__ __forceinline__ up_f32x2 up_fma(float g, up_f32x2 x, up_f32x2 v) { return __builtin_elementwise_fma(up_f32x2{g, g}, x, v); }
__device__ __forceinline__ float up_lane(float v, int) { return v; }
__device__ __forceinline__ float up_lane(up_f32x2 v, int k) { return v[k]; }

template <class T, int UX, int UY, int DX, int DY, int FW, int FH, int PHX, int PHY,
          int MX, int MY, int BX, int BY, int REM, int PL = 1>
__global__ __launch_bounds__(BX* BY) void upfirdn2d_tile_kernel(UpfirdnParams p, int tiles_x, int tiles_y, int nitems, int rem_x, int rem_y) {
    static_assert(MX % UX == 0 && MY % UY == 0, "micro-tile must cover whole phases");
    static_assert((BX * BY) % 64 == 0, "whole wavefronts");
    static_assert(REM == 0 || (UX == 1 && UY == 1 && DX == 1 && DY == 1), "remainder strips exist for the 1:1 kernels only");
    static_assert(PL == 1 || PL == 2, "one plane, or a pair of planes, per work item");
    typedef typename UpVec<PL>::type V;
    constexpr int NT = BX * BY;
    constexpr int TOW = BX * MX, TOH = BY * MY;
    constexpr int TIW = ((TOW + REM - 1) * DX + FW - 1) / UX + 2;
    constexpr int TIH = ((TOH + REM - 1) * DY + FH - 1) / UY + 2;
    constexpr int LDW = TIW | 1;   // odd row pitch keeps strided column reads off one bank
    constexpr int NLOAD = (TIH * TIW + NT - 1) / NT;
    __shared__ V sx[TIH * LDW];

    const int tid = threadIdx.x;
    const int tx = tid % BX, ty = tid / BX;

    // filter taps, flipped as requested; uniform -> scalar registers
    float g[FH][FW];
#pragma unroll
    for (int a = 0; a < FH; a++)
#pragma unroll
        for (int b = 0; b < FW; b++)
            g[a][b] = p.f[(p.flip ? a : FH - 1 - a) * FW + (p.flip ? b : FW - 1 - b)];

    // this thread's slots of the input tile (fixed): slot e = tid + j*NT -> (row, column)
    int slot_row[NLOAD], slot_col[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; j++) {
        const int e = tid + j * NT;
        slot_row[j] = e / TIW; slot_col[j] = e - slot_row[j] * TIW;
    }

    // A work item is one output tile of one plane; the workgroup walks items blockIdx.x, +gridDim.x, ... and
    // fetches the next item's input tile into registers while it computes the current one from LDS.
    V stage[NLOAD];
    auto fetch = [&](int item) {
        const int tile_x = item % tiles_x, r = item / tiles_x;
        const int tile_y = r % tiles_y, plane = (r / tiles_y) * PL;       // the item's first plane
        const int ix0 = (tile_x * TOW * DX - p.padx0 + PHX) / UX;      // exact divisions by construction
        const int iy0 = (tile_y * TOH * DY - p.pady0 + PHY) / UY;
        const T* xp = (const T*)p.x + (int64_t)plane * p.inH * p.inW;
#pragma unroll
        for (int j = 0; j < NLOAD; j++) {
            const int iy = iy0 + slot_row[j], ix = ix0 + slot_col[j];
            const bool in = slot_row[j] < TIH && iy >= 0 && iy < p.inH && ix >= 0 && ix < p.inW;
            if constexpr (PL == 2) {
                up_f32x2 v = {0.f, 0.f};
                if (in) { v[0] = ld<T>(xp + (int64_t)iy * p.inW + ix); v[1] = ld<T>(xp + (int64_t)(p.inH + iy) * p.inW + ix); }
                stage[j] = v;
            } else {
                float v = 0.f;
                if (in) v = ld<T>(xp + (int64_t)iy * p.inW + ix);
                stage[j] = v;
            }
        }
    };

    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    int item = blockIdx.x;
    if (item < nitems) fetch(item);
    while (item < nitems) {
        __syncthreads();   // the previous item's LDS reads are done
#pragma unroll
        for (int j = 0; j < NLOAD; j++)
            if (slot_row[j] < TIH) sx[slot_row[j] * LDW + slot_col[j]] = stage[j];
        __syncthreads();
        const int next = item + gridDim.x;
        if (next < nitems) fetch(next);

        const int tile_x = item % tiles_x, r_ = item / tiles_x;
        const int tile_y = r_ % tiles_y, plane = (r_ / tiles_y) * PL;
        const int64_t OHW = (int64_t)p.outH * p.outW;
        T* yp = (T*)p.y + (int64_t)plane * OHW;
        const int rx = tx * (MX * DX / UX), ry = ty * (MY * DY / UY);
        V acc[MY][MX];
#pragma unroll
        for (int b = 0; b < MY; b++)
#pragma unroll
            for (int a = 0; a < MX; a++) {
                V v = V(0.f);
#pragma unroll
                for (int jy = 0; jy < c_ntaps(b, DY, UY, PHY, FH); jy++)
#pragma unroll
                    for (int jx = 0; jx < c_ntaps(a, DX, UX, PHX, FW); jx++)
                        v = up_fma(g[c_tap0(b, DY, UY, PHY) + jy * UY][c_tap0(a, DX, UX, PHX) + jx * UX],
                                   sx[(ry + c_in0(b, DY, UY, PHY) + jy) * LDW + rx + c_in0(a, DX, UX, PHX) + jx], v);
                acc[b][a] = v * p.gain;
            }
        const int ox = tile_x * TOW + tx * MX, oy = tile_y * TOH + ty * MY;
        const T* ap = p.y_add ? (const T*)p.y_add + (int64_t)plane * OHW : nullptr;
        if (ap) {                                       // the addend of this thread's outputs, fetched in one go in front of the stores
#pragma unroll
            for (int b = 0; b < MY; b++)
#pragma unroll
                for (int a = 0; a < MX; a++)
                    if (oy + b < p.outH && ox + a < p.outW) {
                        if constexpr (PL == 2) { acc[b][a][0] += ld<T>(ap + (int64_t)(oy + b) * p.outW + ox + a); acc[b][a][1] += ld<T>(ap + OHW + (int64_t)(oy + b) * p.outW + ox + a); }
                        else acc[b][a] += ld<T>(ap + (int64_t)(oy + b) * p.outW + ox + a);
                    }
        }
#pragma unroll
        for (int b = 0; b < MY; b++) {
            if (oy + b >= p.outH) break;
#pragma unroll
            for (int a = 0; a < MX; a++)
                if (ox + a < p.outW) {
#pragma unroll
                    for (int k = 0; k < PL; k++) {
                        const float v = up_lane(acc[b][a], k);
                        st<T>(yp + k * OHW + (int64_t)(oy + b) * p.outW + ox + a, v);
                        if (p.y_amax) amax_take(am, v);
                    }
                }
        }
        if constexpr (REM == 1) {
            // one extra output column / row / corner of the plane, by the last thread column / row of the last tile
            const bool ex = rem_x && tile_x == tiles_x - 1 && tx == BX - 1;
            const bool ey = rem_y && tile_y == tiles_y - 1 && ty == BY - 1;
            auto one = [&](int yy, int xx) {          // output (yy, xx) relative to the tile origin
                // the OTHER dimension may end inside this tile (non-square planes: 20 x 33 on 32 x 32 tiles)
                if (tile_y * TOH + yy >= p.outH || tile_x * TOW + xx >= p.outW) return;
                V v = V(0.f);
#pragma unroll
                for (int jy = 0; jy < FH; jy++)
#pragma unroll
                    for (int jx = 0; jx < FW; jx++) v = up_fma(g[jy][jx], sx[(yy + jy) * LDW + xx + jx], v);
                v *= p.gain;
#pragma unroll
                for (int k = 0; k < PL; k++) {
                    float w = up_lane(v, k);
                    if (ap) w += ld<T>(ap + k * OHW + (int64_t)(tile_y * TOH + yy) * p.outW + tile_x * TOW + xx);
                    st<T>(yp + k * OHW + (int64_t)(tile_y * TOH + yy) * p.outW + tile_x * TOW + xx, w);
                    if (p.y_amax) amax_take(am, w);
                }
            };
            if (ex) {
#pragma unroll
                for (int b = 0; b < MY; b++) one(ty * MY + b, TOW);
            }
            if (ey) {
#pragma unroll
                for (int a = 0; a < MX; a++) one(TOH, tx * MX + a);
            }
            if (ex && ey) one(TOH, TOW);
        }
        item = next;
    }
    __shared__ uint32_t amred[NT / 64];
    amax_commit_block<NT>(am, aslot, amred);          // one commit per workgroup
}

//------------------------------------------------------------------------------------
// Small-plane kernel: the 4..16-pixel layers.  A plane is a few hundred bytes, so a workgroup takes PL consecutive
// planes of the NCHW tensor: one contiguous, fully coalesced read of PL * inH * inW elements into LDS, every thread
// then produces outputs of those planes from LDS (any factors / filter size, taps from a scalar table), and the PL
// output planes leave as one contiguous write.
constexpr int SMALL_LDS_FLOATS = 8192;      // 32 KB of input planes per workgroup

// U, D, F > 0: compile-time factors and a square F x F filter (the live classes: 4-tap blur, x2 up, /2 down) -- the taps sit
// in registers and the tap loops unroll; U = 0: any parameters (taps from a table in memory).  Output index -> (plane, row,
// column) by multiplication with host-computed reciprocals (exact below 2^20 / divisor outputs per pass).
template <class T, int U, int D, int F>
__global__ __launch_bounds__(256) void upfirdn2d_small_kernel(UpfirdnParams p, int planes, int PL, unsigned magic_sz, unsigned magic_w) {
    __shared__ float sx[SMALL_LDS_FLOATS];
    const int in_sz = p.inH * p.inW, out_sz = p.outH * p.outW;
    float g[U > 0 ? F : 1][U > 0 ? F : 1];
    if constexpr (U > 0) {
#pragma unroll
        for (int a = 0; a < F; a++)
#pragma unroll
            for (int b = 0; b < F; b++) g[a][b] = p.f[(p.flip ? a : F - 1 - a) * F + (p.flip ? b : F - 1 - b)];
    }
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    for (int p0 = blockIdx.x * PL; p0 < planes; p0 += gridDim.x * PL) {
        const int np = planes - p0 < PL ? planes - p0 : PL;
        const T* xp = (const T*)p.x + (int64_t)p0 * in_sz;
        __syncthreads();
        for (int e = threadIdx.x; e < np * in_sz; e += 256) sx[e] = ld<T>(xp + e);
        __syncthreads();
        T* yp = (T*)p.y + (int64_t)p0 * out_sz;
        for (int e = threadIdx.x; e < np * out_sz; e += 256) {
            const int pl = (int)__umulhi((unsigned)e, magic_sz), r = e - pl * out_sz;
            const int oy = (int)__umulhi((unsigned)r, magic_w), ox = r - oy * p.outW;
            const float* sp = sx + pl * in_sz;
            float v = 0.f;
            if constexpr (U > 0) {
                const int ux0 = ox * D - p.padx0, uy0 = oy * D - p.pady0;
#pragma unroll
                for (int ty = 0; ty < F; ty++) {
                    const int uy = uy0 + ty;
                    const int iy = U == 1 ? uy : uy >> 1;
                    const bool yok = (U == 1 || (uy & 1) == 0) && (unsigned)iy < (unsigned)p.inH;     // U in {1, 2}
#pragma unroll
                    for (int tx = 0; tx < F; tx++) {
                        const int ux = ux0 + tx;
                        const int ix = U == 1 ? ux : ux >> 1;
                        const bool ok = yok && (U == 1 || (ux & 1) == 0) && (unsigned)ix < (unsigned)p.inW;
                        v = fmaf(ok ? sp[iy * p.inW + ix] : 0.f, g[ty][tx], v);
                    }
                }
                v *= p.gain;
                if (p.y_add) v += ld<T>((const T*)p.y_add + (int64_t)p0 * out_sz + e);
                st<T>(yp + e, v);
                if (p.y_amax) amax_take(am, v);
            } else {
                const int ux0 = ox * p.downx - p.padx0, uy0 = oy * p.downy - p.pady0;
                for (int ty = posmod(-uy0, p.upy); ty < p.fh; ty += p.upy) {
                    const int uy = uy0 + ty;
                    if (uy < 0) continue;
                    const int iy = uy / p.upy;
                    if (iy >= p.inH) break;
                    const int gy = p.flip ? ty : p.fh - 1 - ty;
                    for (int tx = posmod(-ux0, p.upx); tx < p.fw; tx += p.upx) {
                        const int ux = ux0 + tx;
                        if (ux < 0) continue;
                        const int ix = ux / p.upx;
                        if (ix >= p.inW) break;
                        const int gx = p.flip ? tx : p.fw - 1 - tx;
                        v = fmaf(sp[iy * p.inW + ix], p.f[gy * p.fw + gx], v);
                    }
                }
                v *= p.gain;
                if (p.y_add) v += ld<T>((const T*)p.y_add + (int64_t)p0 * out_sz + e);
                st<T>(yp + e, v);
                if (p.y_amax) amax_take(am, v);
            }
        }
    }
    amax_commit(am, aslot);
}

// ceil(2^32 / d): umulhi(e, magic) == e / d for e * d < 2^32
static unsigned recip_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <class T>
static bool try_small(const UpfirdnParams& p, hipStream_t s) {
    const int in_sz = p.inH * p.inW;
    const int64_t out_sz = (int64_t)p.outH * p.outW;
    const int64_t planes = (int64_t)p.N * p.C;
    if (in_sz > 1024 || out_sz > 4096 || planes > INT32_MAX) return false;
    int PL = SMALL_LDS_FLOATS / in_sz;
    // enough workgroups to fill the chip before planes are stacked deeper
    while (PL > 1 && planes / PL < 512) PL >>= 1;
    if ((int64_t)PL * out_sz * out_sz >= (1ll << 32) || out_sz == 1 || p.outW == 1) return false;      // reciprocal range (d = 1 has no 32-bit reciprocal)
    const int64_t groups = (planes + PL - 1) / PL;
    const dim3 grid((unsigned)(groups < 4096 ? groups : 4096));
    const unsigned msz = recip_magic((int)out_sz), mw = recip_magic(p.outW);
    const bool sq4 = p.fw == 4 && p.fh == 4 && p.upx == p.upy && p.downx == p.downy;
#define PASTA_SMALL(U_, D_, F_) hipLaunchKernelGGL((upfirdn2d_small_kernel<T, U_, D_, F_>), grid, dim3(256), 0, s, p, (int)planes, PL, msz, mw)
    if (sq4 && p.upx == 1 && p.downx == 1) PASTA_SMALL(1, 1, 4);
    else if (sq4 && p.upx == 2 && p.downx == 1) PASTA_SMALL(2, 1, 4);
    else if (sq4 && p.upx == 1 && p.downx == 2) PASTA_SMALL(1, 2, 4);
    else PASTA_SMALL(0, 0, 0);
#undef PASTA_SMALL
    return true;
}

template <class T, int UX, int UY, int DX, int DY, int FW, int FH, int PHX, int PHY, int MX, int MY, int BX, int BY>
static void launch_tile(const UpfirdnParams& p, hipStream_t s) {
    constexpr int TOW = BX * MX, TOH = BY * MY;
    int tiles_x = (p.outW + TOW - 1) / TOW, tiles_y = (p.outH + TOH - 1) / TOH;
    int rem_x = 0, rem_y = 0;
    if constexpr (UX == 1 && UY == 1 && DX == 1 && DY == 1) {
        rem_x = (p.outW > TOW && p.outW % TOW == 1) ? 1 : 0;
        rem_y = (p.outH > TOH && p.outH % TOH == 1) ? 1 : 0;
        tiles_x -= rem_x; tiles_y -= rem_y;
    }
    const int64_t planes = (int64_t)p.N * p.C;
    static const bool pairs_on = !(getenv("PASTA_UPFIRDN_PAIRS") && getenv("PASTA_UPFIRDN_PAIRS")[0] == '0');       // A/B switch
    // fp32, an even number of planes: two planes per work item (packed FMAs on plane pairs), as long as the items still fill the chip
    constexpr bool PAIRS_T = !std::is_same<T, double>::value;        // fp32 and the 16-bit storage types (fp32 arithmetic inside); fp64 keeps one plane
    const bool pair = PAIRS_T && pairs_on && (planes & 1) == 0 && (int64_t)tiles_x * tiles_y * (planes / 2) >= 2048;      // (the RGB up-sampling, 768 pair items, ran 2x slower on pairs: measured)
    const int64_t nitems = (int64_t)tiles_x * tiles_y * (pair ? planes / 2 : planes);
    const int64_t grid = nitems < 256 * 16 ? nitems : 256 * 16;     // up to 16 resident-or-queued workgroups per CU
#define PASTA_TILE(REM_, PL_, RX_, RY_) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, UX, UY, DX, DY, FW, FH, PHX, PHY, MX, MY, BX, BY, REM_, PL_>), \
                                                          dim3((unsigned)grid), dim3(BX * BY), 0, s, p, tiles_x, tiles_y, (int)nitems, RX_, RY_)
    if constexpr (UX == 1 && UY == 1 && DX == 1 && DY == 1) {
        if (rem_x || rem_y) {
            if constexpr (PAIRS_T) { if (pair) { PASTA_TILE(1, 2, rem_x, rem_y); return; } }
            PASTA_TILE(1, 1, rem_x, rem_y);
            return;
        }
    }
    if constexpr (PAIRS_T) { if (pair) { PASTA_TILE(0, 2, 0, 0); return; } }
    PASTA_TILE(0, 1, 0, 0);
#undef PASTA_TILE
}

// Phase dispatch (runtime pad0 mod U -> template constant).
template <class T, int UX, int UY, int DX, int DY, int FW, int FH, int MX, int MY, int BX, int BY>
static void launch_phase(const UpfirdnParams& p, hipStream_t s) {
    const int phx = posmod(p.padx0, UX), phy = posmod(p.pady0, UY);
    if constexpr (UX == 1 && UY == 1) {
        launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 0, MX, MY, BX, BY>(p, s);
    } else if constexpr (UX == 2 && UY == 1) {
        if (phx == 0) launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 0, MX, MY, BX, BY>(p, s);
        else          launch_tile<T, UX, UY, DX, DY, FW, FH, 1, 0, MX, MY, BX, BY>(p, s);
    } else if constexpr (UX == 1 && UY == 2) {
        if (phy == 0) launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 0, MX, MY, BX, BY>(p, s);
        else          launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 1, MX, MY, BX, BY>(p, s);
    } else {
        static_assert(UX == 2 && UY == 2, "unsupported phase set");
        if (phx == 0 && phy == 0)      launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 0, MX, MY, BX, BY>(p, s);
        else if (phx == 1 && phy == 0) launch_tile<T, UX, UY, DX, DY, FW, FH, 1, 0, MX, MY, BX, BY>(p, s);
        else if (phx == 0 && phy == 1) launch_tile<T, UX, UY, DX, DY, FW, FH, 0, 1, MX, MY, BX, BY>(p, s);
        else                           launch_tile<T, UX, UY, DX, DY, FW, FH, 1, 1, MX, MY, BX, BY>(p, s);
    }
}

// Shapes on the PASTA-GAN path: 4x4 taps at 1:1, x2 up, /2 down
// (and the mirrored gradients), separable 12-tap x2 up / /2 down in x or y (ADA).
// Returns true when a tile kernel was launched.
template <class T>
static bool try_tile(const UpfirdnParams& p, hipStream_t s) {
    // Planes narrower than half a wavefront (the 4..16 pixel layers): several planes per workgroup (try_small)
    if (p.outW < 24) return false;
#define PASTA_UPF(UX, UY, DX, DY, FW_, FH_, MX, MY)                                                           \
    if (p.upx == UX && p.upy == UY && p.downx == DX && p.downy == DY && p.fw == FW_ && p.fh == FH_) {        \
        if (p.outW <= 40 * MX) launch_phase<T, UX, UY, DX, DY, FW_, FH_, MX, MY, 32, 8>(p, s);               \
        else                   launch_phase<T, UX, UY, DX, DY, FW_, FH_, MX, MY, 64, 4>(p, s);               \
        return true;                                                                                          \
    }
    PASTA_UPF(1, 1, 1, 1, 4, 4, 1, 4)
    PASTA_UPF(2, 2, 1, 1, 4, 4, 2, 4)
    PASTA_UPF(1, 1, 2, 2, 4, 4, 1, 2)
    PASTA_UPF(1, 1, 1, 1, 3, 3, 1, 4)
    PASTA_UPF(2, 1, 1, 1, 12, 1, 2, 2)
    PASTA_UPF(1, 2, 1, 1, 1, 12, 1, 2)
    PASTA_UPF(1, 1, 2, 1, 12, 1, 1, 2)
    PASTA_UPF(1, 1, 1, 2, 1, 12, 1, 2)
#undef PASTA_UPF
    return false;
}

template <class T>
static int run(const UpfirdnParams& p, bool dense_nchw, hipStream_t s) {
    if (!(dense_nchw && (try_tile<T>(p, s) || try_small<T>(p, s)))) {
        const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
        int64_t blocks = ceil_div64(total, 256);
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL((upfirdn2d_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
    return launch_status("upfirdn2d");
}

}  // namespace pasta

extern "C" int pasta_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                               const int32_t in_size[4], const int64_t in_stride[4],
                               const int32_t f_size[2],
                               const int32_t out_size[4], const int64_t out_stride[4],
                               int upx, int upy, int downx, int downy,
                               int padx0, int padx1, int pady0, int pady1,
                               int flip, float gain, void* stream, float* y_amax, const void* y_add) {
    using namespace pasta;
    PASTA_CHECK(x && f && y, "upfirdn2d: null pointer");
    PASTA_CHECK(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");
    PASTA_CHECK(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1");
    PASTA_CHECK(f_size[0] >= 1 && f_size[1] >= 1, "upfirdn2d: f must be at least 1x1");
    for (int i = 0; i < 4; i++) PASTA_CHECK(in_size[i] >= 1, "upfirdn2d: empty input dimension %d", i);

    UpfirdnParams p;
    p.x = x; p.f = f; p.y = y;
    p.N = in_size[0]; p.C = in_size[1]; p.inH = in_size[2]; p.inW = in_size[3];
    p.fh = f_size[0]; p.fw = f_size[1];
    p.outW = (p.inW * upx + padx0 + padx1 - p.fw + downx) / downx;
    p.outH = (p.inH * upy + pady0 + pady1 - p.fh + downy) / downy;
    PASTA_CHECK(p.inW * upx + padx0 + padx1 - p.fw >= 0 && p.inH * upy + pady0 + pady1 - p.fh >= 0 && p.outW >= 1 && p.outH >= 1,
                "upfirdn2d: output must be at least 1x1");
    PASTA_CHECK(out_size[0] == p.N && out_size[1] == p.C && out_size[2] == p.outH && out_size[3] == p.outW,
                "upfirdn2d: output buffer is [%d,%d,%d,%d], expected [%d,%d,%d,%d]",
                out_size[0], out_size[1], out_size[2], out_size[3], p.N, p.C, p.outH, p.outW);
    PASTA_CHECK((int64_t)p.N * p.C * p.inH * p.inW <= INT32_MAX, "upfirdn2d: x is too large");
    PASTA_CHECK((int64_t)p.N * p.C * p.outH * p.outW <= INT32_MAX, "upfirdn2d: output is too large");
    p.isn = in_stride[0]; p.isc = in_stride[1]; p.isy = in_stride[2]; p.isx = in_stride[3];
    p.osn = out_stride[0]; p.osc = out_stride[1]; p.osy = out_stride[2]; p.osx = out_stride[3];
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy;
    p.padx0 = padx0; p.pady0 = pady0; p.flip = flip ? 1 : 0; p.gain = gain; p.y_amax = y_amax; p.y_add = y_add;

    const bool dense_nchw =
        p.isx == 1 && p.isy == p.inW && p.isc == (int64_t)p.inH * p.inW && (p.N == 1 || p.isn == p.isc * p.C) &&
        p.osx == 1 && p.osy == p.outW && p.osc == (int64_t)p.outH * p.outW && (p.N == 1 || p.osn == p.osc * p.C);

    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case PASTA_F32: return run<float>(p, dense_nchw, s);
        case PASTA_F16: return run<__half>(p, dense_nchw, s);
        case PASTA_BF16: return run<__bf16>(p, dense_nchw, s);
        case PASTA_F64: return run<double>(p, false, s);       // generic kernel only (fp64 accumulation)
        default: return fail("upfirdn2d: unsupported dtype code %d", dtype);
    }
}
