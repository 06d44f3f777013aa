// Dense convolution family for gfx950: conv2d, conv_transpose2d and the weight gradient of either, NCHW fp32, on the
// matrix cores -- split-bf16 with fp32-equivalent products (v_mfma_f32_32x32x16_bf16, default) or fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  This file holds the launch plans, the descriptor checks and the C entry points.  The kernels
// live in the headers included below as templates, which this file never instantiates (it includes them for their host-side
// shape predicates): every kernel family is compiled in a translation unit of its own, conv_tu_*.hip, and reached through
// the functions of conv_launch.h -- thirteen units built in parallel (round 5; one unit of 161 kernels took 3 - 12 minutes):
//   conv_common.h              parameter blocks, tile enumeration, epilogue
//   conv_fwd_f32.h             fp32-MFMA forward-type kernel, weight packing        -> conv_tu_pack_f32.hip
//   conv_fwd_bf16x6.h          split forward-type kernels (base and row-reuse)      -> conv_tu_fwd_base_{128,64}.hip, conv_tu_fwd_rows_{128,64}.hip
//   conv_fwd_rows2d_bf16x6.h   2-D pixel tiles                                      -> conv_tu_rows2d_{wide,128_r4,128_r2,64_r8}.hip
//   conv_fwd_1x1.h, conv_fwd_s2.h  pointwise and stride-2 kernels                   -> conv_tu_fwd_small.hip
//   conv_wgrad_f32.h           fp32-MFMA weight gradients, few-channel kernels, slab reductions -> conv_tu_wgrad_f32.hip
//   conv_wgrad_bf16x6.h        split weight-gradient kernels                        -> conv_tu_wgrad_{3x3,3x3s2,1x1}.hip
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
#include "conv_launch.h"
#include "conv_fwd_f32.h"
#include "conv_fwd_bf16x6.h"
#include "conv_fwd_rows2d_bf16x6.h"
#include "conv_fwd_1x1.h"
#include "conv_fwd_s2.h"
#include "conv_fwd_fewch.h"
#include "conv_wgrad_f32.h"
#include "conv_wgrad_bf16x6.h"

namespace pasta {

// Packed input-channel padding: a multiple of the KC of the kernel instance that will run.
static int fwd_ipad(int Ig, FwdTile t) { return (Ig <= 4 && t == T64x256) ? 4 : Ig <= 8 ? 8 : 16; }

// The plain six-product fp32 launch of a 3x3 stride-1 lattice on 2-D tiles, if the plane divides into them.
static bool try_fwd_rows2d(bool tile128, const ConvFwdParams& p, hipStream_t s) {
    if (!p.rows || p.ncls != 1 || p.cls[0].T != 9 || !p.bf16x6) return false;        // every arithmetic and storage type (an input scale implies fp32 storage, six products)
    int ymin = p.tap_dy[0], ymax = p.tap_dy[0], xmin = p.tap_dx[0], xmax = p.tap_dx[0];
    for (int t = 1; t < 9; t++) {
        ymin = p.tap_dy[t] < ymin ? p.tap_dy[t] : ymin; ymax = p.tap_dy[t] > ymax ? p.tap_dy[t] : ymax;
        xmin = p.tap_dx[t] < xmin ? p.tap_dx[t] : xmin; xmax = p.tap_dx[t] > xmax ? p.tap_dx[t] : xmax;
    }
    if (ymax - ymin != 2 || xmax - xmin != 2 || xmin != p.rows_d0) return false;
    ConvFwdParams q = p;
    q.rows_y0 = ymin;
    if (tile128) {
        // eight waves on 128 x 256 (conv_tu_rows2d_wide.hip): fp32 storage, fp32-equivalent products; an input scale under the three-product arithmetic only
        if (rows2d_wide(p.cls[0].P, p.cls[0].Q) && (p.bf16x6 == 3 || p.bf16x6 == NP_F16X3) && p.io == IO_F32 && (!p.iscale || p.bf16x6 == NP_F16X3)) {
            if (q.x_pieces) tu_rows2d_wide_pieces(q, s); else tu_rows2d_wide(q, s);
            return true;
        }
        const int R = rows2d_rows(p.cls[0].P, p.cls[0].Q);
        if (R == 4) { tu_rows2d_128_r4(q, s); return true; }
        if (R == 2) { tu_rows2d_128_r2(q, s); return true; }
    } else {
        // 64 x 256 tile: 8 rows x 32 columns (B image 10 x 34 slots, 65 KB double-buffered + 16 KB of weights: two workgroups per CU, just)
        if (rows2d_rows256(p.cls[0].P, p.cls[0].Q)) { tu_rows2d_64_r8(q, s); return true; }
    }
    return false;
}


// The split forward-type kernels on a 128 x 128 or 64 x 256 tile: the row-reuse kernel where the lattice is made of whole row segments, else the base kernel.
static void launch_fwd_bf16x6(bool tile128, const ConvFwdParams& p, hipStream_t s) {
    const int BM = tile128 ? 128 : 64, BN = tile128 ? 128 : 256;
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    int64_t tiles = 0;
    for (int c = 0; c < p.ncls; c++) {
        const int64_t t = ceil_div64((int64_t)p.N * p.cls[c].P * p.cls[c].Q, BN);
        if (t > tiles) tiles = t;
    }
    tiles *= p.ncls;
    const dim3 grid((unsigned)tiles, q.o_tiles * q.ksplit, p.G);
    if (tile128) { if (!tu_fwd_rows_128(q, grid, s)) tu_fwd_base_128(q, grid, s); }
    else         { if (!tu_fwd_rows_64(q, grid, s)) tu_fwd_base_64(q, grid, s); }
}

static void dispatch_fwd(FwdTile t, const ConvFwdParams& p, hipStream_t s) {
    if (p.bf16x6 && (t == T128x128 || t == T64x256)) {
        if (!try_fwd_rows2d(t == T128x128, p, s)) launch_fwd_bf16x6(t == T128x128, p, s);
        return;
    }
    tu_fwd_f32(t, p, s);
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// y[n,c,:] = oscale[n,c] * sum_ks partial[ks][n,c,:]   (fixed order; split-K epilogue)
template <int IO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, void* __restrict__ y,
                                                            const float* __restrict__ oscale, int64_t numel, int ohw, int ksplit,
                                                            const float* __restrict__ bias, int cout, int act, float alpha, float gain,
                                                            float clamp, const void* __restrict__ res, const float* __restrict__ noise,
                                                            const float* __restrict__ noise_strength, int noise_ps, float* __restrict__ y_amax) {
    const float nstr = noise ? noise_strength[0] : 0.f;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(y_amax);
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
        const float nz = noise ? noise[(noise_ps ? (nc / cout) * (int64_t)ohw : 0) + (i - nc * ohw)] * nstr : 0.f;
        v = conv_scale_noise(v, oscale ? oscale + nc : nullptr, 0, nz);
        if (res) v += io_ld1<IO>((const char*)res + i * io_size<IO>::value);
        if (act) v = conv_epilogue(v, bias ? bias[nc % cout] : 0.f, act, alpha, gain, clamp);
        io_st<IO>(y, i, v);
        if (y_amax) amax_take(am, v);
    }
    amax_commit(am, aslot);
}

// K slices for launches that would leave most CUs idle (the 4..17 pixel layers: K = 9*512 against <= 4624 pixels).
static int64_t fwd_lattice_pixels(const pasta_conv_desc* d) {
    if (!d->transposed) return (int64_t)d->N * d->OH * d->OW;
    return (int64_t)d->N * ((d->OH + d->stride - 1) / d->stride) * ((d->OW + d->stride - 1) / d->stride);
}

// Which kernel a forward-type launch uses: the tile, the number of K slices, and whether the split-bf16 kernel may
// run (it also needs iscale == nullptr, known only at launch).
// bf16 pieces per operand of the split-bf16 kernels for a math mode
// (PASTA_MATH_F16X3: the pseudo count NP_F16X3 -- fp16 pieces, three products; conv_common.h)
static int math_pieces(int math) { return math == PASTA_MATH_BF16 ? 1 : math == PASTA_MATH_BF16X3 ? 2 : math == PASTA_MATH_BF16X6 ? 3 : NP_F16X3; }    // PASTA_MATH_DEFAULT = PASTA_MATH_F16X3
static bool fp32_equivalent(int pieces) { return pieces == 3 || pieces == NP_F16X3; }

// Leading floats of every convolution workspace: the partial |max| of the two operands (PASTA_MATH_F16X3)
constexpr int WS_AMAX_FLOATS = 2 * AMAX_PARTS;

// parts[i] *= max |v|: the bound of |x * iscale| from the bound of |x| (one workgroup; iscale is [N, C_in])
__global__ __launch_bounds__(256) void amax_times_kernel(const float* __restrict__ parts_in, const float* __restrict__ v, int n, float* __restrict__ parts_out) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float a = fabsf(v[i]); m = (a < __builtin_inff() && a > m) ? a : m; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    parts_out[threadIdx.x] = parts_in[threadIdx.x] * m;
}

struct FwdPlan { FwdTile tile; int ksplit; int bf16x6; int packed; };     // packed: the few-input-channel mode (conv_fwd_bf16x6_kernel, KT)

// Do the split-bf16 kernels of this launch take the input scale (modulation) in their staging code?
static bool isc_in_staging(const pasta_conv_desc* d) { return d->io_dtype == PASTA_F32 && fp32_equivalent(math_pieces(d->math)); }

static FwdPlan plan_fwd(const pasta_conv_desc* d) {
    const int Og = d->C_out / d->groups, Ig = d->C_in / d->groups;
    const int64_t npix = fwd_lattice_pixels(d);
    const bool sb = (d->math != PASTA_MATH_F32 || d->io_dtype != PASTA_F32) && Ig >= 16 && (int64_t)d->N * d->C_in * d->H * d->W < (1ll << 30);
    // fewer than 16 input channels into more than 32 output channels over a large plane with at least 64 (channel, tap) pairs -- the 7x7
    // RGB stems: 0.394 -> 0.234 ms.  Below (3x3: 27 pairs, 1x1: 3) the output store is what the launch costs and the fp32 kernel's
    // epilogue is the faster one: 0.113 -> 0.141 ms and 0.205 -> 0.366 ms when forced (profiles/r3_ab_packed_k.txt)
    static const bool packed_on = !(getenv("PASTA_PACKED_K") && getenv("PASTA_PACKED_K")[0] == '0');
    const bool few = packed_on && !d->transposed && d->groups == 1 && Ig < 16 && Og > 32 && npix > 8192 && d->io_dtype == PASTA_F32 &&
                     d->math != PASTA_MATH_F32 && fp32_equivalent(math_pieces(d->math)) && Ig * d->kh * d->kw >= 64 && Ig * d->kh * d->kw <= 1024 &&
                     (int64_t)d->N * d->C_in * (d->H + 2 * d->pad_h) * (d->W + 2 * d->pad_w) < (1ll << 28);
    FwdPlan f;
    f.packed = few;
    // ToRGB / parsing heads (<= 16 output channels): HBM-bound, few rows, fp32 MFMA.  17..32 output channels (the 512^2 block of the
    // 512 generator) take the 64-row split-bf16 tile half empty: 80 (fp32 storage) / 175 (16-bit) TFLOP/s effective against 55 on the
    // fp32 tile, and 16-bit tensors are not converted for the launch.
    if (Og <= 32 && !(sb && Og > 16 && npix > 8192)) f.tile = T32x256;
    else if (npix <= 8192) f.tile = (sb && Og > 64) ? T128x128 : T64x64;     // 4..16 pixel layers: K is sliced to fill the chip
    else if (Og <= 64) f.tile = T64x256;
    else f.tile = T128x128;
    f.bf16x6 = (sb || few) && (f.tile == T128x128 || f.tile == T64x256);
    f.packed = f.packed && f.bf16x6;
    f.ksplit = 1;
    if (npix <= 8192 && f.tile != T32x256) {
        const int bm = fwd_tile_bm(f.tile), bn = f.tile == T64x64 ? 64 : 128;
        int64_t blocks = ceil_div64(npix, bn) * ((Og + bm - 1) / bm) * d->groups;
        // conv_transpose2d: a parity class has between 1 and ceil(k/u)^2 of the taps; the slices are sized for the
        // smallest class, and on the split-bf16 kernel the u*u classes share the grid (merged_classes)
        const int taps = d->transposed ? 1 : d->kh * d->kw;
        if (d->transposed && f.bf16x6 && d->stride == 2) blocks *= 4;
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

struct WgradPlan {
    int TR, TS, WA, WB, pipe, npos, kp, bf16x6, tgr, tgs, a_tiles, b_tiles, cw_log2, qblocks, chunks_total, ksplit, rows_total;
    int64_t slab_floats; size_t lds_bytes;
};

// ks_multiple > 1 (pasta_conv2d_wgrad_modulated: the batch size): the number of K slices is rounded UP to a multiple of it, so that no slice
// straddles two samples (the chunks are numbered sample-major and N divides their count where the caller checked)
// 3x3 stride-1 pad-1 weight gradients over planes of 16-pixel rows under a split arithmetic (round 5): the split kernel's chunk is 32 consecutive
// pixels of a row, so these ran on the fp32-MFMA kernel (75 - 99 TFLOP/s: 157 peak).  A 16-pixel row is taken as a 32-pixel chunk whose second half
// is zero (the S loads of the missing pixels are masked; the L halo's validity bits already zero the columns past the row): half of the MFMAs
// multiply zeros, and the launch still runs twice as fast.  8-pixel rows (a quarter filled) stay where they are.  PASTA_WGRAD_WIDE16=0: off.
static bool wgrad_wide16(const pasta_conv_desc* d) {
    static const bool enabled = !(getenv("PASTA_WGRAD_WIDE16") && getenv("PASTA_WGRAD_WIDE16")[0] == '0');
    const int P = d->transposed ? d->H : d->OH, Q = d->transposed ? d->W : d->OW;
    const int LH = d->transposed ? d->OH : d->H, LW = d->transposed ? d->OW : d->W;
    return enabled && (d->math != PASTA_MATH_F32 || d->io_dtype != PASTA_F32) && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           Q == 16 && LH == P && LW == Q && !d->x_layout;
}

// wide16 (round 5): 16-pixel rows as HALF-FILLED 32-pixel chunks of the split 3x3 stride-1 kernel (wgrad_wide16 below) instead of two-row chunks
// of the fp32 kernel
static WgradPlan plan_wgrad(int N, int P, int Q, int G, int Ag, int Bg, int kh, int kw, int st, int ks_multiple = 1, bool wide16 = false) {
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
        while (cw > 1 && cw / 2 >= Q && !wide16) { cw /= 2; lg--; }
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
    if (ks_multiple > 1) ks = (ks + ks_multiple - 1) / ks_multiple * ks_multiple;
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
    PASTA_CHECK(d->math >= PASTA_MATH_DEFAULT && d->math <= PASTA_MATH_F16X3, "%s: unknown math mode %d", who, d->math);
    PASTA_CHECK(d->io_dtype == PASTA_F32 || d->io_dtype == PASTA_F16 || d->io_dtype == PASTA_BF16, "%s: io_dtype %d is not PASTA_F32 / PASTA_F16 / PASTA_BF16", who, d->io_dtype);
    PASTA_CHECK(d->groups >= 1 && d->C_in % d->groups == 0 && d->C_out % d->groups == 0, "%s: channels not divisible by groups=%d", who, d->groups);
    PASTA_CHECK(d->x_layout == PASTA_LAYOUT_NCHW || d->x_layout == PASTA_LAYOUT_PIECES16, "%s: unknown x_layout %d", who, d->x_layout);
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

namespace pasta {
// floats of the zero-padded input copy of the packed-K mode (0: the convolution has no padding, the input itself serves)
static int64_t packed_input_floats(const pasta_conv_desc* d) {
    if (d->pad_h == 0 && d->pad_w == 0) return 0;
    return (int64_t)d->N * d->C_in * (d->H + 2 * d->pad_h) * (d->W + 2 * d->pad_w);
}
}

namespace pasta { static bool t2_launch_ok(const pasta_conv_desc* d, int pieces, int ksplit, int launch_flags); }

extern "C" int64_t pasta_conv2d_workspace(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_workspace")) return -1;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    const FwdPlan f = plan_fwd(d);
    const FwdTile t = f.tile;
    const int ks = f.ksplit;
    // packed weights: fp32 (4 B) or three bf16 pieces (6 B) per element; sized for the larger, in floats
    int64_t pack = ((int64_t)d->groups * d->kh * d->kw * round_up(Ig, fwd_ipad(Ig, t)) * round_up(Og, fwd_tile_bm(t)) * 3 + 1) / 2;
    const int64_t partial = ks > 1 ? (int64_t)ks * d->N * d->C_out * d->OH * d->OW : 0;
    int64_t extra = 0;
    if (f.packed) {         // [O][C_in kh kw] packed as a 1x1 weight, the offset table, the zero-padded copy of the input
        const int64_t kp = round_up(Ig * d->kh * d->kw, 16);
        const int64_t pk = (kp * round_up(Og, fwd_tile_bm(t)) * 3 + 1) / 2;
        pack = pack > pk ? pack : pk;
        extra = kp + packed_input_floats(d);
    }
    if (t2_launch_ok(d, math_pieces(d->math), ks, 0)) extra = (int64_t)d->N * d->C_in * d->H;      // conv_fwd_t2.h: the input's last column, gathered
    const int64_t rowinv = (int64_t)d->groups * round_up(Og, fwd_tile_bm(t));      // PASTA_MATH_F16X3: 1 / S_w per packed weight row
    return (WS_AMAX_FLOATS + rowinv + round_up((int)pack, 4) + partial + round_up((int)extra, 4)) * (int64_t)sizeof(float);
}

extern "C" int pasta_conv2d_tile(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_tile")) return -1;
    return (int)plan_fwd(d).tile;
}

namespace pasta {
// Does conv3x3s2_f16x3_kernel take this launch with x as the producer wrote it (PASTA_LAYOUT_PIECES16)?  The conditions of conv3x3s2_ok (conv_fwd_s2.h)
// on the descriptor, plus whole channel octets and the blur's pad 0.
static bool pieces_fwd_ok(const pasta_conv_desc* d, int launch_flags) {
    const FwdPlan f = plan_fwd(d);
    // (round 5) ... or the eight-wave 2-D tile kernel (plan kernel 7: 3x3 stride-1 convolution or input gradient, >= 128 output channels, planes of
    // 8 rows x 32 columns), plain launches of the three-product arithmetic
    if (d->stride == 1 && d->kh == 3 && d->kw == 3 && f.bf16x6 && !f.packed && f.tile == T128x128 && math_pieces(d->math) == NP_F16X3 && d->io_dtype == PASTA_F32 &&
        d->groups == 1 && !(launch_flags & (PASTA_PLAN_ISCALE | PASTA_PLAN_MODULATED)) && d->C_in >= 16 && (d->C_in & 7) == 0 && !d->x2 &&
        rows2d_rows(d->OH, d->OW) > 0 && rows2d_wide(d->OH, d->OW))
        return true;
    return !d->transposed && f.bf16x6 && !f.packed && math_pieces(d->math) == NP_F16X3 && d->io_dtype == PASTA_F32 && d->groups == 1 && d->kh == 3 && d->kw == 3 &&
           d->stride == 2 && d->pad_h == 0 && d->pad_w == 0 && !(launch_flags & (PASTA_PLAN_ISCALE | PASTA_PLAN_OSCALE | PASTA_PLAN_MODULATED)) && f.ksplit == 1 &&
           d->C_in >= 16 && (d->C_in & 7) == 0 && d->C_out > 32 && !d->x2 && conv3x3s2_shape_ok(d->OH, d->OW) &&
           !(getenv("PASTA_CONV_S2") && getenv("PASTA_CONV_S2")[0] == '0');
}
static bool pair_launch_ok(const pasta_conv_desc* d, int pieces, int ksplit, FwdTile tile, bool plain);
// The parity-pair kernel carries no scales and no epilogue: ONE predicate for the planner and the launch.
static inline bool pair_plain(int launch_flags) { return launch_flags == 0; }
static inline int launch_flags_of(const float* iscale, const float* oscale, const pasta_conv_epilogue* ep) {
    return (iscale ? PASTA_PLAN_ISCALE : 0) | (oscale ? PASTA_PLAN_OSCALE : 0) | (ep ? PASTA_PLAN_EPILOGUE : 0);
}
}

extern "C" int pasta_conv2d_plan(const pasta_conv_desc* d, int launch_flags, int* tile, int* ksplit, int* math, int* launches, int* kernel) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d_plan")) return e;
    const bool has_iscale = (launch_flags & PASTA_PLAN_ISCALE) != 0;
    const FwdPlan f = plan_fwd(d);
    const bool packed = f.packed && !has_iscale && !(launch_flags & PASTA_PLAN_MODULATED);
    const bool sb = f.bf16x6 && (!has_iscale || isc_in_staging(d)) && (!f.packed || packed);
    const int few_kind = conv1x1_fewch_kind(d, has_iscale, (launch_flags & PASTA_PLAN_OSCALE) != 0, false, (launch_flags & PASTA_PLAN_MODULATED) != 0);
    if (d->io_dtype != PASTA_F32 && !sb && !few_kind)
        return fail("conv2d: no 16-bit-storage kernel for this shape (fewer than 16 input channels per group, at most 32 "
                    "output channels, or an input scale -- pointwise layers over more than 8192 pixels excepted): convert the tensors to fp32 for this launch");
    if (const int few = conv1x1_fewch_kind(d, has_iscale, (launch_flags & PASTA_PLAN_OSCALE) != 0, false, (launch_flags & PASTA_PLAN_MODULATED) != 0)) {
        // a streaming fp32 kernel on the raw weights (conv_fwd_fewch.h): no packing, no operand scale
        if (tile) *tile = (int)f.tile;
        if (ksplit) *ksplit = 1;
        if (math) *math = PASTA_MATH_F32;
        if (launches) *launches = 1;
        if (kernel) *kernel = 10 + few;
        return 0;
    }
    if (tile) *tile = (int)f.tile;
    if (ksplit) *ksplit = f.ksplit;
    if (math) *math = !sb ? PASTA_MATH_F32 : d->io_dtype != PASTA_F32 ? PASTA_MATH_BF16 : d->math == PASTA_MATH_BF16X3 ? PASTA_MATH_BF16X3 : d->math == PASTA_MATH_BF16 ? PASTA_MATH_BF16 :
                       d->math == PASTA_MATH_BF16X6 ? PASTA_MATH_BF16X6 : PASTA_MATH_F16X3;
    const bool t2 = sb && t2_launch_ok(d, math_pieces(d->math), f.ksplit, launch_flags);
    const bool pair = t2 || (sb && pair_launch_ok(d, math_pieces(d->math), f.ksplit, f.tile, pair_plain(launch_flags)));
    if (launches) *launches = !d->transposed ? 1 : t2 ? 1 : pair ? 1 + (d->OH > 2 * d->H || d->OW > 2 * d->W ? 1 : 0) : merged_classes(d, sb) ? 1 :
                              (d->stride < d->OH ? d->stride : d->OH) * (d->stride < d->OW ? d->stride : d->OW);
    if (kernel) {
        // the lattice of a stride-1 launch is the output plane itself, its taps kh rows of kw adjacent offsets
        const bool rows = sb && d->stride == 1 && d->kw == 3 && rows_tile_ok(d->OH, d->OW, f.tile == T128x128 ? 128 : 256);
        const bool plain6 = sb && d->stride == 1 && d->kw == 3 && d->kh == 3;
        const bool rows2d = plain6 && f.tile == T128x128 && rows2d_rows(d->OH, d->OW) > 0;
        const bool rows2d_256 = plain6 && f.tile == T64x256 && rows2d_rows256(d->OH, d->OW);
        const bool wide = rows2d && (!has_iscale || math_pieces(d->math) == NP_F16X3) && fp32_equivalent(math_pieces(d->math)) && d->io_dtype == PASTA_F32 && rows2d_wide(d->OH, d->OW);
        static const bool c1x1_on = !(getenv("PASTA_CONV1X1") && getenv("PASTA_CONV1X1")[0] == '0');
        const int bn1 = (d->C_out / d->groups) <= 64 ? 256 : 128;
        const bool c1x1 = c1x1_on && sb && math_pieces(d->math) == NP_F16X3 && d->io_dtype == PASTA_F32 && d->groups == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 &&
                          !d->pad_h && !d->pad_w && !(launch_flags & (PASTA_PLAN_ISCALE | PASTA_PLAN_OSCALE)) && f.ksplit == 1 && !packed && d->C_in >= 16 &&
                          d->C_out > 32 && d->OH == d->H && d->OW == d->W && ((int64_t)d->H * d->W) % bn1 == 0;
        static const bool s2_on = !(getenv("PASTA_CONV_S2") && getenv("PASTA_CONV_S2")[0] == '0');
        const bool s2k = s2_on && sb && !d->transposed && math_pieces(d->math) == NP_F16X3 && d->io_dtype == PASTA_F32 && d->groups == 1 && d->kh == 3 && d->kw == 3 &&
                         d->stride == 2 && d->pad_h == d->pad_w && d->pad_h <= 1 && !(launch_flags & (PASTA_PLAN_ISCALE | PASTA_PLAN_OSCALE)) && f.ksplit == 1 && !packed &&
                         d->C_in >= 16 && d->C_out > 32 && conv3x3s2_shape_ok(d->OH, d->OW);
        *kernel = !sb ? 0 : c1x1 ? 9 : s2k ? 10 : packed ? 8 : t2 ? 13 : pair ? 3 : wide ? 7 : rows2d ? (rows2d_rows(d->OH, d->OW) == 4 ? 4 : 5) : rows2d_256 ? 6 : rows ? 2 : 1;
    }
    if (d->x_layout == PASTA_LAYOUT_PIECES16 && !pieces_fwd_ok(d, launch_flags))
        return fail("conv2d: x_layout = PASTA_LAYOUT_PIECES16 is served by the 3x3 stride-2 forward kernel (conv2d, pad 0, fp32 y, PASTA_MATH_F16X3, one group, "
                    "C_in a multiple of 8 and >= 16, C_out > 32, output width a power of two >= 16, more than 8192 output pixels, no scale vectors) and by the "
                    "eight-wave 3x3 stride-1 tile kernel (plan kernel 7, no input scale, C_in a multiple of 8) only");
    return 0;
}

extern "C" int pasta_conv2d(const void* x, const float* w, void* y, const float* iscale, const float* oscale,
                            const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream) {
    return pasta_conv2d_ex(x, w, y, iscale, oscale, nullptr, d, workspace, workspace_bytes, stream);
}

namespace pasta {
// Stride-2 3x3 conv_transpose2d whose output covers the doubled input plane (OH = 2H or 2H + 1): the parity-pair mode of
// the row-reuse kernel (conv_fwd_bf16x6.h) when the launch is the plain six-product fp32 convolution.
static bool pair_launch_ok(const pasta_conv_desc* d, int pieces, int ksplit, FwdTile tile, bool plain) {
    static const bool enabled = !(getenv("PASTA_T2_PAIR") && getenv("PASTA_T2_PAIR")[0] == '0');
    if (!enabled || !d->transposed || d->stride != 2 || d->kh != 3 || d->kw != 3 || d->pad_h != d->pad_w || d->pad_h > 1) return false;
    if ((pieces != 3 && pieces != NP_F16X3) || d->io_dtype != PASTA_F32 || !plain || ksplit != 1) return false;
    if (d->OH < 2 * d->H || d->OH > 2 * d->H + 1 || d->OW < 2 * d->W || d->OW > 2 * d->W + 1) return false;
    if (tile != T128x128 && tile != T64x256) return false;
    // Measured (profiles/r2_conv_pairs.txt): onto 2H x 2W outputs (no remainder) the pair kernel is 1.4x the per-class launch at
    // every size; with the remainder row / column it wins where the main launch outlasts the remainder's K loop (a few
    // dozen workgroups, 0.1 - 0.3 ms of serial latency however little they compute): input planes of 128 x 128 and larger.
    const bool remainder = d->OH > 2 * d->H || d->OW > 2 * d->W;
    if (remainder && (int64_t)d->H * d->W < 128 * 128 && !(getenv("PASTA_T2_PAIR") && getenv("PASTA_T2_PAIR")[0] == '2')) return false;
    return rows_tile_ok(d->H, d->W, tile == T128x128 ? 128 : 256);
}

// ... or the one-pass kernel over the input lattice (conv_fwd_t2.h, round 5; plan kernel 13): pad 0, the three-product arithmetic, planes of
// 8 x 32 tiles, an input scale allowed (the modulated layers of the training step), nothing behind the sum.  Takes precedence over the pair mode.
static bool t2_launch_ok(const pasta_conv_desc* d, int pieces, int ksplit, int launch_flags) {
    static const bool enabled = !(getenv("PASTA_CONV_T2") && getenv("PASTA_CONV_T2")[0] == '0');
    if (!enabled || !d->transposed || d->stride != 2 || d->kh != 3 || d->kw != 3 || d->pad_h != 0 || d->pad_w != 0) return false;
    if (pieces != NP_F16X3 || d->io_dtype != PASTA_F32 || (launch_flags & ~(PASTA_PLAN_ISCALE | PASTA_PLAN_MODULATED)) || ksplit != 1 || d->x2 || d->x_layout) return false;
    if (d->OH < 2 * d->H || d->OH > 2 * d->H + 1 || d->OW < 2 * d->W || d->OW > 2 * d->W + 1) return false;
    // Every plane of 8 x 32 or 16 x 16 tiles.  (At 32 x 32 and 16 x 16 the regular tiles of a batch of 16 fill the chip once or half, and what the edge
    // tiles in front of them take is added to the launch: +7 % / +12 % there, +30 % / +50 % on the discriminator's stacked batches of 48 --
    // profiles/r5_ab_conv_t2.txt.)  PASTA_CONV_T2=1: planes of 64 x 64 and larger only.
    static const bool large_only = getenv("PASTA_CONV_T2") && getenv("PASTA_CONV_T2")[0] == '1';
    return d->C_in / d->groups >= 16 && ((d->H % 8 == 0 && d->W % 32 == 0) || (d->H % 16 == 0 && d->W % 16 == 0)) && ((int64_t)d->H * d->W >= 4096 || !large_only);
}

// The remainder of the parity-pair launch: output row 2H and / or column 2W of a stride-2 conv_transpose2d onto an odd plane --
// 1 % of the outputs, each a dot product over ONE input row or column (one or two taps).  As lattices of the MFMA kernels these
// were a few dozen workgroups whose K loops are as long as anyone's: 0.17 ms of latency behind a 0.2 ms main launch (measured,
// profiles/r3_ab_pair_f16x3.txt).  Here: plain fp32 FMAs on the raw weights, one thread per (pixel, 16 output channels), the
// lanes of a wave along the lattice -- thousands of short independent chains instead of thirty long ones.
struct EdgeWeights { const float* w; const float* mod_s; const float* mod_d; float wscale; int flip; };

template <bool MOD>          // MOD: one shared weight modulated per group on the way (pasta_conv2d_modulated), as the packing kernel does
__global__ __launch_bounds__(256) void conv_t2_edge_kernel(ConvFwdParams p, EdgeWeights ew) {
    constexpr int OC = 16, KC = 64;                      // a workgroup: 64 lattice pixels x 64 output channels, K in chunks of 64 channels
    __shared__ float xs[KC][64];                         // [channel][pixel]
    __shared__ __attribute__((aligned(16))) float wsm[KC][64];     // [channel][output channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    int c = 0, tile = blockIdx.x;
    for (; c < p.ncls - 1; c++) {                        // classes share the grid's x axis
        const int t = (p.N * p.cls[c].P * p.cls[c].Q + 63) >> 6;
        if (tile < t) break;
        tile -= t;
    }
    const int P = p.cls[c].P, Q = p.cls[c].Q, T = p.cls[c].T, tap0 = p.cls[c].tap0;
    const int pix = tile * 64 + lane;                    // the same pixel in all four waves: lanes along the lattice
    const bool live = pix < p.N * P * Q;
    const int n = live ? pix / (P * Q) : 0;
    const int rem = live ? pix - n * P * Q : 0;
    const int pp = rem / Q, qq = rem - pp * Q;
    const int ob = blockIdx.y * 64;                      // this workgroup's output channels; this wave's: ob + 16 wave ...
    const int HW = p.H * p.W;
    float acc[OC];
#pragma unroll
    for (int j = 0; j < OC; j++) acc[j] = 0.f;
    const float* const xb = p.x + ((int64_t)n * p.Cin + (int64_t)g * p.Ig) * HW;
    const int gs = MOD ? 0 : g;
    const int wo = ob + lane < p.Og ? ob + lane : p.Og - 1;           // staging role of this thread: weight column `lane`
    const float wlive = ob + lane < p.Og ? ew.wscale : 0.f;
    float md = 1.f;
    if constexpr (MOD) md = ew.mod_d ? ew.mod_d[(int64_t)g * p.Og + wo] : 1.f;
    for (int t = 0; t < T; t++) {
        const int iy = pp + p.tap_dy[tap0 + t], ix = qq + p.tap_dx[tap0 + t];
        const bool ok = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* const xp = xb + (ok ? iy * p.W + ix : 0);
        const int slab = ew.flip ? 8 - p.tap_slab[tap0 + t] : p.tap_slab[tap0 + t];
        const float* const wt = ew.w + ((int64_t)gs * p.Ig * p.Og + wo) * 9 + slab;      // [C_in][C_out / G][3][3]
        for (int i0 = 0; i0 < p.Ig; i0 += KC) {
            // every load of the chunk in flight at once: one round trip for the column gather, one for the weights
            float xr[KC / 4], wr[KC / 4];
#pragma unroll
            for (int k = 0; k < KC / 4; k++) {
                const int ii = i0 + wave + 4 * k;
                const int ic = ii < p.Ig ? ii : p.Ig - 1;
                xr[k] = (ok && ii < p.Ig) ? xp[(int64_t)ic * HW] : 0.f;
                float wv = wt[(int64_t)ic * p.Og * 9] * (ii < p.Ig ? wlive : 0.f);
                if constexpr (MOD) { wv *= ew.mod_s[(int64_t)g * p.Ig + ic]; wv *= md; }
                wr[k] = wv;
            }
            __syncthreads();                             // the previous chunk has been consumed
#pragma unroll
            for (int k = 0; k < KC / 4; k++) { xs[wave + 4 * k][lane] = xr[k]; wsm[wave + 4 * k][lane] = wr[k]; }
            __syncthreads();
#pragma unroll 8
            for (int ii = 0; ii < KC; ii++) {
                const float xv = xs[ii][lane];
#pragma unroll
                for (int q4 = 0; q4 < OC / 4; q4++) {
                    const float4 w4 = *(const float4*)&wsm[ii][wave * OC + 4 * q4];      // wave-uniform address: a broadcast read
                    acc[4 * q4 + 0] = fmaf(xv, w4.x, acc[4 * q4 + 0]);
                    acc[4 * q4 + 1] = fmaf(xv, w4.y, acc[4 * q4 + 1]);
                    acc[4 * q4 + 2] = fmaf(xv, w4.z, acc[4 * q4 + 2]);
                    acc[4 * q4 + 3] = fmaf(xv, w4.w, acc[4 * q4 + 3]);
                }
            }
        }
    }
    if (!live) return;
    const int o0 = ob + wave * OC;
    float* const yb = p.y + (((int64_t)n * p.Cout + (int64_t)g * p.Og + o0) * p.OH + p.cls[c].oy0 + pp * p.osy) * p.OW + p.cls[c].ox0 + qq * p.osx;
#pragma unroll
    for (int j = 0; j < OC; j++)
        if (o0 + j < p.Og) yb[(int64_t)j * p.OH * p.OW] = acc[j];
}

// Packed-K mode (conv_fwd_bf16x6_kernel, KT): byte offset of "channel" k = (input channel c, tap (ty, tx)) from a pixel's base
// address in the padded input -- the tap the weight element [o][c][ty][tx] multiplies (mirrored when the launch flips the weight).
__global__ __launch_bounds__(256) void packed_koff_kernel(unsigned* __restrict__ koff, int K, int kh, int kw, int flip, int HWp, int Wp) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int c = k / (kh * kw), t = k - c * kh * kw;
    int ty = t / kw, tx = t - ty * kw;
    if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
    koff[k] = (unsigned)(c * HWp + ty * Wp + tx) * 4u;
}

__global__ __launch_bounds__(256) void pad_planes_kernel(const float* __restrict__ x, float* __restrict__ xp, int64_t planes, int H, int W,
                                                         int ph, int pw) {
    const int Hp = H + 2 * ph, Wp = W + 2 * pw;
    const int64_t total = planes * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / (Hp * Wp);
        const int r = (int)(i - plane * Hp * Wp);
        const int y = r / Wp - ph, xx = r - (r / Wp) * Wp - pw;
        xp[i] = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) ? x[(plane * H + y) * W + xx] : 0.f;
    }
}

static void launch_transposed_pairs(const pasta_conv_desc* d, const ConvFwdParams& base, FwdTile tile, hipStream_t s, const EdgeWeights& ew) {
    const int pad = d->pad_h, H = d->H, W = d->W;
    // remainder: output row 2H and / or column 2W, by conv_t2_edge_kernel over the class tables below (everything stays on the
    // caller's stream: the library owns nothing persistent)
    const bool xrow = d->OH == 2 * H + 1, xcol = d->OW == 2 * W + 1;
    ConvFwdParams q = base;
    q.rows = 0; q.ncls = 0;
    if (xrow || xcol) {
        int ntap = 0;
        auto add_class = [&](int a, int b, int P, int Q, int oy0, int ox0, int py_shift, int px_shift) {
            const int tap0 = ntap;
            for (int r = 0; r < 3; r++) {
                if (posmod(a + pad - r, 2) != 0) continue;
                if (py_shift && floordiv(a + pad - r, 2) + py_shift >= H) continue;       // reads below the last input row: zero for the whole class
                for (int c = 0; c < 3; c++) {
                    if (posmod(b + pad - c, 2) != 0) continue;
                    if (px_shift && floordiv(b + pad - c, 2) + px_shift >= W) continue;   // right of the last input column
                    q.tap_dy[ntap] = floordiv(a + pad - r, 2) + py_shift;
                    q.tap_dx[ntap] = floordiv(b + pad - c, 2) + px_shift;
                    q.tap_slab[ntap] = r * 3 + c;
                    ntap++;
                }
            }
            q.cls[q.ncls++] = {P, Q, oy0, ox0, ntap - tap0, tap0};
        };
        if (xrow) {                                           // oy = 2H (a = 0, p = H): every column
            add_class(0, 0, 1, (d->OW + 1) / 2, 2 * H, 0, H, 0);
            add_class(0, 1, 1, d->OW / 2, 2 * H, 1, H, 0);
        }
        if (xcol) {                                           // ox = 2W (b = 0, q = W): the rows below 2H
            add_class(0, 0, H, 1, 0, 2 * W, 0, W);
            add_class(1, 0, H, 1, 1, 2 * W, 0, W);
        }
    }
    static const int edge_mode = getenv("PASTA_T2_EDGE") ? getenv("PASTA_T2_EDGE")[0] - '0' : 1;      // A/B: 0 = the remainder as MFMA lattices, 2 = after the main launch
    auto launch_edge = [&]() {
        if (!q.ncls) return;
        if (edge_mode == 0) { dispatch_fwd(tile, q, s); return; }
        int tiles = 0;
        for (int c = 0; c < q.ncls; c++) tiles += (q.N * q.cls[c].P * q.cls[c].Q + 63) >> 6;
        const dim3 grid((unsigned)tiles, (unsigned)((q.Og + 63) / 64), (unsigned)q.G);
        if (ew.mod_s) hipLaunchKernelGGL(conv_t2_edge_kernel<true>, grid, dim3(256), 0, s, q, ew);
        else hipLaunchKernelGGL(conv_t2_edge_kernel<false>, grid, dim3(256), 0, s, q, ew);
    };
    if (edge_mode != 2) launch_edge();
    ConvFwdParams p = base;
    // main lattice: (p, q) of the input plane -> outputs (2p + a, 2q + b), a, b in {0, 1}
    p.ncls = 2; p.rows = 1; p.rows_rev = 0;
    p.pair_bx = pad & 1;                                  // the column with two taps (c = 0, 2)
    const int dx0 = (p.pair_bx + pad) / 2;                // input offset of tap c = 0; tap c = 2 reads one pixel to its left
    const int dx1 = floordiv((1 - p.pair_bx) + pad - 1, 2);
    p.rows_d0 = dx0 - 1;
    p.pair_off[0] = 1; p.pair_off[1] = dx1 - p.rows_d0; p.pair_off[2] = 0;
    for (int k = 0; k < 2; k++) {
        const int a = k == 0 ? (pad & 1) : 1 - (pad & 1);          // class 0: the parity with two kernel rows
        int nt = 0;
        for (int r = 0; r < 3; r++) {
            if (posmod(a + pad - r, 2) != 0) continue;
            for (int c = 0; c < 3; c++) {
                p.tap_dy[6 * k + nt] = floordiv(a + pad - r, 2);
                p.tap_dx[6 * k + nt] = 0;
                p.tap_slab[6 * k + nt] = r * 3 + c;
                nt++;
            }
        }
        p.cls[k] = {H, W, a, 0, nt, 6 * k};
    }
    if (tile == T128x128) tu_fwd_pair_128(p, s); else tu_fwd_pair_64(p, s);
    if (edge_mode == 2) launch_edge();
}

static int conv2d_run(const void* x, const float* w, void* y, const float* iscale, const float* oscale,
                      const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                      void* stream, const float* wmod_s, const float* wmod_d);
}

extern "C" int pasta_conv2d_ex(const void* x, const float* w, void* y, const float* iscale, const float* oscale,
                               const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    return pasta::conv2d_run(x, w, y, iscale, oscale, ep, d, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int pasta_conv2d_modulated(const void* x, const float* w, const float* styles, const float* dcoefs, void* y,
                                      const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
    using namespace pasta;
    PASTA_CHECK(styles, "conv2d_modulated: null styles");
    return conv2d_run(x, w, y, nullptr, nullptr, ep, d, workspace, workspace_bytes, stream, styles, dcoefs);
}

namespace pasta {
// The packing job of a plain launch (no scale vectors, plain weights, one input tensor) of the default arithmetic on fp32 tensors, as conv2d_run
// would perform it at the head of the launch: false where the launch packs differently or not at all (few-channel kernels, the packed-K
// mode of the stems, 16-bit storage, the other arithmetics) -- the caller then leaves w_prepacked at 0.
static bool pack_job_of(const pasta_conv_desc* d, void* workspace, PackJob& j) {
    if (d->io_dtype != PASTA_F32 || math_pieces(d->math) != NP_F16X3 || d->x2) return false;
    if (conv1x1_fewch_kind(d, false, false, false, false)) return false;
    const FwdPlan plan = plan_fwd(d);
    if (!plan.bf16x6 || plan.packed) return false;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    float* const ws_rowinv = (float*)workspace + WS_AMAX_FLOATS;
    j.rowinv = ws_rowinv;
    j.wp = ws_rowinv + (int64_t)d->groups * round_up(Og, fwd_tile_bm(plan.tile));
    j.G = d->groups; j.Ig = Ig; j.Og = Og;
    j.Ig_pad = round_up(Ig, fwd_ipad(Ig, plan.tile)); j.Og_pad = round_up(Og, fwd_tile_bm(plan.tile));
    j.kh = d->kh; j.kw = d->kw; j.transposed = d->transposed; j.flip = d->flip;
    j.wscale = d->wscale == 0.f ? 1.f : d->wscale;
    static const int pack_xcd = getenv("PASTA_PACK_XCD") ? atoi(getenv("PASTA_PACK_XCD")) : 1;
    j.pack_xcd_rows = (j.Og_pad & 63) == 0 ? pack_xcd : 0;
    return true;
}
}

extern "C" int pasta_conv2d_pack_pair(const float* w, const pasta_conv_desc* da, void* ws_a, int64_t ws_a_bytes, const pasta_conv_desc* db, void* ws_b,
                                      int64_t ws_b_bytes, void* stream, int* packed_mask) {
    using namespace pasta;
    PASTA_CHECK(w && da && db && ws_a && ws_b && packed_mask, "conv2d_pack_pair: null pointer");
    *packed_mask = 0;
    if (int e = check_desc(da, "conv2d_pack_pair")) return e;
    if (int e = check_desc(db, "conv2d_pack_pair")) return e;
    PASTA_CHECK(ws_a_bytes >= pasta_conv2d_workspace(da) && ws_b_bytes >= pasta_conv2d_workspace(db), "conv2d_pack_pair: workspace too small");
    PASTA_CHECK((((uintptr_t)ws_a | (uintptr_t)ws_b) & 15) == 0, "conv2d_pack_pair: workspaces must be 16-byte aligned");
    static const bool enabled = !(getenv("PASTA_PACK_PAIR") && getenv("PASTA_PACK_PAIR")[0] == '0');
    PackJob a, b;
    // both or nothing: one orientation alone is the launch the convolution would have made itself
    if (!enabled || da->groups != db->groups || !pack_job_of(da, ws_a, a) || !pack_job_of(db, ws_b, b)) return 0;
    tu_pack_weights_f16x3_pair(w, a, b, (hipStream_t)stream);
    *packed_mask = 3;
    return launch_status("conv2d_pack_pair");
}

int pasta::conv2d_run(const void* x, const float* w, void* y, const float* iscale, const float* oscale,
                      const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                      void* stream, const float* wmod_s, const float* wmod_d) {
    if (int e = check_desc(d, "conv2d")) return e;
    PASTA_CHECK(!ep || (ep->act >= 1 && ep->act <= 3), "conv2d: fused epilogue supports act 1..3 (linear, relu, lrelu), got %d", ep ? ep->act : 0);
    PASTA_CHECK(!ep || !ep->noise || ep->noise_strength, "conv2d: noise without noise_strength");
    PASTA_CHECK(x && w && y, "conv2d: null pointer");
    const int64_t need = pasta_conv2d_workspace(d);
    PASTA_CHECK(workspace && workspace_bytes >= need, "conv2d: workspace of %lld bytes needed, %lld given", (long long)need, (long long)workspace_bytes);
    PASTA_CHECK(((uintptr_t)workspace & 15) == 0, "conv2d: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (const int few = conv1x1_fewch_kind(d, iscale != nullptr, oscale != nullptr, ep && ep->noise, wmod_s != nullptr)) {
        FewChParams q;
        q.x = x; q.w = w; q.y = y; q.iscale = iscale; q.io = d->io_dtype;
        q.bias = ep ? ep->bias : nullptr; q.res = ep ? ep->res : nullptr; q.y_amax = ep ? ep->y_amax : nullptr;
        q.N = d->N; q.Cin = d->C_in; q.Cout = d->C_out; q.HW = d->H * d->W;
        q.w_io = d->transposed ? 1 : 0;
        q.wscale = d->wscale == 0.f ? 1.f : d->wscale;
        q.act = ep ? ep->act : 0; q.alpha = ep ? ep->alpha : 0.f; q.gain = ep ? ep->gain : 1.f; q.clamp = ep ? ep->clamp : -1.f;
        tu_conv1x1_fewch(few, q, s);
        return launch_status("conv2d");
    }

    float* const ws_amax = (float*)workspace;                         // [2][AMAX_PARTS]: partial |max| of x (second row: spare)
    float* const ws_rowinv = ws_amax + WS_AMAX_FLOATS;                // [G][Og_pad]: 1 / S_w per packed weight row (PASTA_MATH_F16X3)
    {
        const FwdPlan pl = plan_fwd(d);
        workspace = ws_rowinv + (int64_t)d->groups * round_up(d->C_out / d->groups, fwd_tile_bm(pl.tile));      // packed weights and K-slice partial sums follow
    }
    ConvFwdParams p;
    p.x = (const float*)x; p.y = (float*)y; p.wp = (const float*)workspace; p.iscale = iscale; p.oscale = oscale;
    p.x_amax = nullptr; p.w_rowinv = nullptr;
    p.x_pieces = d->x_layout == PASTA_LAYOUT_PIECES16;
    if (p.x_pieces) {
        PASTA_CHECK(d->x_amax, "conv2d: x_layout = PASTA_LAYOUT_PIECES16 needs x_amax, the row pasta_blur_pieces wrote (the operand's scale)");
        PASTA_CHECK(pieces_fwd_ok(d, launch_flags_of(iscale, oscale, nullptr) | (wmod_s ? PASTA_PLAN_MODULATED : 0)) && !(ep && ep->noise),
                    "conv2d: no kernel takes x_layout = PASTA_LAYOUT_PIECES16 for this launch (pasta_conv2d_plan tells beforehand)");
    }
    p.x2 = (const float*)d->x2; p.x2_amax = nullptr; p.C1 = d->C1;
    PASTA_CHECK(!d->x2 || (d->C1 > 0 && d->C1 < d->C_in && d->groups == 1 && !wmod_s), "conv2d: a second input tensor needs 0 < C1 < C_in, one group and plain weights");
    p.N = d->N; p.Cin = d->C_in; p.H = d->H; p.W = d->W;
    p.Cout = d->C_out; p.OH = d->OH; p.OW = d->OW;
    p.G = d->groups; p.Ig = d->C_in / d->groups; p.Og = d->C_out / d->groups;
    const FwdPlan plan = plan_fwd(d);
    const FwdTile tile = plan.tile;
    p.Ig_pad = round_up(p.Ig, fwd_ipad(p.Ig, tile)); p.Og_pad = round_up(p.Og, fwd_tile_bm(tile));
    p.KK = d->kh * d->kw;
    p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0; p.res = ep ? (const float*)ep->res : nullptr;
    p.alpha = ep ? ep->alpha : 0.f; p.gain = ep ? ep->gain : 1.f; p.clamp = ep ? ep->clamp : -1.f;
    p.noise = ep ? ep->noise : nullptr; p.noise_strength = ep ? ep->noise_strength : nullptr; p.noise_ps = ep ? ep->noise_per_sample : 0;
    p.y_amax = ep ? ep->y_amax : nullptr;
    p.ksplit = plan.ksplit;
    p.o_tiles = 1;
    p.partial = (float*)workspace + round_up((int)(((int64_t)p.G * p.KK * p.Ig_pad * p.Og_pad * 3 + 1) / 2), 4);
    // bf16 pieces per operand; 0 = fp32 kernel.  An input scale rides in the staging of the six-product fp32-storage kernels only.
    p.bf16x6 = (plan.bf16x6 && (!iscale || isc_in_staging(d)) && !(plan.packed && (iscale || wmod_s))) ? math_pieces(d->math) : 0;
    p.io = d->io_dtype;
    if (p.io != IO_F32) {
        PASTA_CHECK(p.bf16x6, "conv2d: no 16-bit-storage kernel for this shape (pasta_conv2d_plan tells beforehand)");
        p.bf16x6 = 1;           // the stored element is the operand
    }
    p.rows = 0; p.rows_d0 = 0; p.rows_rev = 0;

    const float wscale = d->wscale == 0.f ? 1.f : d->wscale;
    if (p.bf16x6 == NP_F16X3 && p.io == IO_F32) {
        // operand scale of x: partial |max| (the caller's, or one pass here), times max |iscale| when the styles ride in the staging.
        // The weights carry one scale per output row, found by their packing kernel (no |max| of w is passed or cached).
        const float* xa = d->x_amax;
        if (!xa) {          // (never with the pieces layout: checked above)
            if (int e = tensor_amax(x, (int64_t)d->N * d->C_in * d->H * d->W, PASTA_F32, ws_amax, s)) return e;
            xa = ws_amax;
        }
        if (iscale) {
            hipLaunchKernelGGL(amax_times_kernel, dim3(1), dim3(256), 0, s, xa, iscale, d->N * d->C_in, ws_amax);
            xa = ws_amax;
        }
        p.x_amax = xa;
        p.w_rowinv = ws_rowinv;
        if (p.x2) {                                     // the second operand's maxima: the caller's, or one pass here (second row of ws_amax)
            p.x2_amax = d->x2_amax;
            if (!p.x2_amax) {
                if (int e = tensor_amax(d->x2, (int64_t)d->N * (d->C_in - d->C1) * d->H * d->W, PASTA_F32, ws_amax + AMAX_PARTS, s)) return e;
                p.x2_amax = ws_amax + AMAX_PARTS;
            }
        }
    }
    p.koff = nullptr;
    static const int xcd_order = getenv("PASTA_XCD_ORDER") ? atoi(getenv("PASTA_XCD_ORDER")) : 1;
    p.xcd_order = xcd_order;
    const bool packed = plan.packed && p.bf16x6 && !iscale && !wmod_s;
    int pk_kh = d->kh, pk_kw = d->kw, pk_tr = d->transposed, pk_flip = d->flip;
    if (packed) {
        // K = (input channel, tap) pairs: one pseudo-tap over C_in kh kw "channels" of a zero-padded input (workspace: ... | offsets | copy)
        const int K = p.Ig * d->kh * d->kw, Kpad = round_up(K, 16);
        const int Hp = d->H + 2 * d->pad_h, Wp = d->W + 2 * d->pad_w;
        int64_t pk = ((int64_t)p.G * d->kh * d->kw * p.Ig_pad * p.Og_pad * 3 + 1) / 2;
        const int64_t pk2 = ((int64_t)Kpad * p.Og_pad * 3 + 1) / 2;
        pk = pk > pk2 ? pk : pk2;
        unsigned* const koff = (unsigned*)((float*)workspace + round_up((int)pk, 4));      // ksplit == 1: no partial sums in between
        hipLaunchKernelGGL(packed_koff_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, koff, K, d->kh, d->kw, d->flip, Hp * Wp, Wp);
        if (d->pad_h || d->pad_w) {
            float* const xp = (float*)koff + Kpad;
            const int64_t total = (int64_t)d->N * d->C_in * Hp * Wp;
            hipLaunchKernelGGL(pad_planes_kernel, dim3((unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096)), dim3(256), 0, s,
                               (const float*)x, xp, (int64_t)d->N * d->C_in, d->H, d->W, d->pad_h, d->pad_w);
            p.x = xp;
        }
        p.koff = koff;
        p.H = Hp; p.W = Wp;
        p.Ig = K; p.Ig_pad = Kpad; p.KK = 1;
        pk_kh = pk_kw = 1; pk_tr = 0; pk_flip = 0;           // [O][C_in kh kw] as it lies: a 1x1 weight over the K "channels"
    }
    if (d->w_prepacked) {
        // the caller packed the weights for this very descriptor beforehand (pasta_conv2d_pack_pair): the kinds of launch pack_job_of describes
        PASTA_CHECK(p.bf16x6 == NP_F16X3 && p.io == IO_F32 && !packed && !wmod_s && !p.x2, "conv2d: w_prepacked with a launch pasta_conv2d_pack_pair does not serve");
    } else {   // pack weights (times wscale)
        if (p.bf16x6 == NP_F16X3 && p.io == IO_F32) {       // two fp16 pieces, one scale per output row found on the way
            static const int pack_xcd = getenv("PASTA_PACK_XCD") ? atoi(getenv("PASTA_PACK_XCD")) : 1;      // A/B switch: 0 = row = workgroup index
            tu_pack_weights_f16x3(w, workspace, ws_rowinv, p.G, p.Ig, p.Og, p.Ig_pad, p.Og_pad, pk_kh, pk_kw, pk_tr, pk_flip, wscale, wmod_s, wmod_d,
                                  (p.Og_pad & 63) == 0 ? pack_xcd : 0, s);
        }
        else if (p.bf16x6)
            tu_pack_weights_bf16(w, workspace, p.G, p.Ig, p.Og, p.Ig_pad, p.Og_pad, pk_kh, pk_kw, pk_tr, pk_flip, wscale, p.io == IO_F16 ? 1 : 0, wmod_s, wmod_d, s);
        else
            tu_pack_weights_f32(w, (float*)workspace, p.G, p.Ig, p.Og, p.Ig_pad, p.Og_pad, d->kh, d->kw, d->transposed, d->flip, wscale, wmod_s, wmod_d, s);
    }

    if (conv1x1_ok(p, d->kh, d->kw, d->stride, d->pad_h, d->pad_w)) {
        tu_conv1x1(p, s);           // conv2d and conv_transpose2d coincide for 1x1 / stride 1 (the packing kernel reads either weight layout)
        return launch_status("conv2d");
    }
    PASTA_CHECK(!p.x2, "conv2d: a second input tensor is served by the pointwise kernel only (1x1, stride 1, fp32 tensors, PASTA_MATH_F16X3, "
                       ">= 16 input and > 32 output channels, planes that divide into 128- / 256-pixel tiles, no scale vectors or noise)");
    if (packed) {
        p.P = d->OH; p.Q = d->OW; p.oy0 = 0; p.ox0 = 0; p.osy = 1; p.osx = 1; p.isy = d->stride; p.isx = d->stride;
        p.T = 1; p.tap_dy[0] = 0; p.tap_dx[0] = 0; p.tap_slab[0] = 0;      // the window's corner in the padded plane; the taps are in koff
        p.ncls = 1; p.cls[0] = {p.P, p.Q, 0, 0, 1, 0};
        launch_fwd_bf16x6(tile == T128x128, p, s);
    } else if (!d->transposed) {
        p.P = d->OH; p.Q = d->OW; p.oy0 = 0; p.ox0 = 0; p.osy = 1; p.osx = 1; p.isy = d->stride; p.isx = d->stride;
        p.T = p.KK;
        for (int r = 0; r < d->kh; r++)
            for (int c = 0; c < d->kw; c++) {
                const int t = r * d->kw + c;
                p.tap_dy[t] = r - d->pad_h; p.tap_dx[t] = c - d->pad_w; p.tap_slab[t] = t;
            }
        p.ncls = 1; p.cls[0] = {p.P, p.Q, 0, 0, p.T, 0};
        if (conv3x3s2_ok(p, d->kh, d->kw, d->stride, d->pad_h, d->pad_w, d->transposed)) {
            tu_conv3x3s2(p, s);
            return launch_status("conv2d");
        }
        detect_tap_rows(p, p.T);
        dispatch_fwd(tile, p, s);
    } else {
        // output row oy = iy*u - pad + r.  For parity class a (oy = a + u*pp): taps r with (a + pad - r) % u == 0,
        // input row = pp + (a + pad - r)/u.
        const int u = d->stride;
        p.osy = u; p.osx = u; p.isy = 1; p.isx = 1;
        if (t2_launch_ok(d, p.bf16x6, p.ksplit, launch_flags_of(iscale, oscale, ep))) {
            p.x2 = p.partial;                             // (no K slices: the region behind the packed weights holds the gathered column)
            tu_conv_t2(p, s);                             // the whole lattice, remainder row and column included, in one launch
            return launch_status("conv2d");
        }
        if (pair_launch_ok(d, p.bf16x6, p.ksplit, tile, pair_plain(launch_flags_of(iscale, oscale, ep)))) {
            launch_transposed_pairs(d, p, tile, s, EdgeWeights{w, wmod_s, wmod_d, wscale, d->flip});
            return launch_status("conv2d");
        }
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
#define PASTA_SK(IO_) hipLaunchKernelGGL(splitk_reduce_kernel<IO_>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)p.partial, (void*)y, oscale, numel, \
                                         d->OH * d->OW, p.ksplit, p.bias, d->C_out, p.act, p.alpha, p.gain, p.clamp, (const void*)p.res, p.noise, p.noise_strength, p.noise_ps, (float*)nullptr)
        if (p.io == IO_BF16) PASTA_SK(IO_BF16); else if (p.io == IO_F16) PASTA_SK(IO_F16); else PASTA_SK(IO_F32);
#undef PASTA_SK
    }
    // y_amax with a launch whose kernel does not take it (fp32 MFMA tiles; K slices: few pixels, thousands of small workgroups in
    // the reduction): one scan of y
    if (p.y_amax && (!p.bf16x6 || p.ksplit > 1) && p.io == IO_F32)
        if (int e = tensor_amax(y, (int64_t)d->N * d->C_out * d->OH * d->OW, PASTA_F32, p.y_amax, s)) return e;
    return launch_status("conv2d");
}

namespace pasta {
// The split-bf16 weight-gradient kernel covers 3x3, stride 1, pad 1, rows of a multiple of 32 pixels.
static bool wgrad_bf16x6(const pasta_conv_desc* d, const WgradPlan& w) {
    const int P = d->transposed ? d->H : d->OH, Q = d->transposed ? d->W : d->OW;
    const int LH = d->transposed ? d->OH : d->H, LW = d->transposed ? d->OW : d->W;
    return (d->math != PASTA_MATH_F32 || d->io_dtype != PASTA_F32) && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           (Q % 32 == 0 || wgrad_wide16(d)) && LH == P && LW == Q && w.kp == 32 && w.cw_log2 == 5;
}
// ... and its stride-2 sibling: 3x3, stride 2, equal pads of 0 or 1, rows of a multiple of 16 pixels.
static bool wgrad_s2_bf16x6(const pasta_conv_desc* d, const WgradPlan& w) {
    const int Q = d->transposed ? d->W : d->OW;
    return (d->math != PASTA_MATH_F32 || d->io_dtype != PASTA_F32) && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_h == d->pad_w && d->pad_h <= 1 &&
           Q % 16 == 0 && w.kp == 16 && w.cw_log2 == 4;
}
// ... with x as the producer wrote it (PASTA_LAYOUT_PIECES16): conv_wgrad3x3s2_pieces_kernel
static bool wgrad_pieces_ok(const pasta_conv_desc* d, const WgradPlan& w) {
    return d->x_layout == PASTA_LAYOUT_PIECES16 && wgrad_s2_bf16x6(d, w) && !d->transposed && d->pad_h == 0 && d->groups == 1 && d->io_dtype == PASTA_F32 &&
           math_pieces(d->math) == NP_F16X3 && (d->C_in & 7) == 0;
}
// ... and the pointwise one: 1x1, stride 1, no padding, planes of a multiple of 32 pixels (ToRGB heads included: the shape is
// bandwidth-bound, so a mostly empty 64-channel tile costs nothing).
static bool wgrad_1x1_bf16x6(const pasta_conv_desc* d, const WgradPlan& w) {
    const int P = d->transposed ? d->H : d->OH, Q = d->transposed ? d->W : d->OW;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    return (d->math != PASTA_MATH_F32 || d->io_dtype != PASTA_F32) && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 &&
           ((int64_t)P * Q) % 32 == 0 && (Ig >= 16 || Og >= 16) && w.kp == 32 && w.WA == w.WB &&
           d->H == d->OH && d->W == d->OW && (int64_t)w.chunks_total == (int64_t)d->N * P * Q / 32;
}
}  // namespace pasta

extern "C" int pasta_conv2d_wgrad_plan(const pasta_conv_desc* d, int* kernel) {
    using namespace pasta;
    if (int e = check_desc(d, "conv2d_wgrad_plan")) return e;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    int k = 0;
    if (plan_wgrad1x1_fewcin(d, plan_wgrad_small(d))) k = 5;
    else if (plan_wgrad_small(d).use) k = 1;
    else {
        const WgradPlan w = d->transposed ? plan_wgrad(d->N, d->H, d->W, d->groups, Ig, Og, d->kh, d->kw, d->stride, 1, wgrad_wide16(d))
                                          : plan_wgrad(d->N, d->OH, d->OW, d->groups, Og, Ig, d->kh, d->kw, d->stride, 1, wgrad_wide16(d));
        if (wgrad_bf16x6(d, w)) k = 2;
        else if (wgrad_s2_bf16x6(d, w)) k = wgrad_pieces_ok(d, w) ? 6 : 3;
        else if (wgrad_1x1_bf16x6(d, w)) k = 4;
    }
    if (kernel) *kernel = k;
    if (d->x_layout == PASTA_LAYOUT_PIECES16 && k != 6)
        return fail("conv2d_wgrad: x_layout = PASTA_LAYOUT_PIECES16 is served by the 3x3 stride-2 weight gradient only (conv2d, pad 0, fp32 dy, PASTA_MATH_F16X3, "
                    "one group, C_in a multiple of 8, output rows of a multiple of 16 pixels)");
    if (d->io_dtype != PASTA_F32 && k < 2)
        return fail("conv2d_wgrad: no 16-bit-storage kernel for this shape: convert the tensors to fp32 for this launch");
    return 0;
}

extern "C" int64_t pasta_conv2d_wgrad_workspace(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_wgrad_workspace")) return -1;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    const WgradSmallPlan ws = plan_wgrad_small(d);
    if (ws.use) return (WS_AMAX_FLOATS + ws.slab_floats) * (int64_t)sizeof(float);
    const WgradPlan w = d->transposed ? plan_wgrad(d->N, d->H, d->W, d->groups, Ig, Og, d->kh, d->kw, d->stride, 1, wgrad_wide16(d))
                                      : plan_wgrad(d->N, d->OH, d->OW, d->groups, Og, Ig, d->kh, d->kw, d->stride, 1, wgrad_wide16(d));
    return (WS_AMAX_FLOATS + w.slab_floats) * (int64_t)sizeof(float);
}

namespace pasta {
// Can the weight gradient of a MODULATED convolution come from the plain kernels with sample-aligned K slices (pasta_conv2d_wgrad_modulated)?
// The main split kernels only (3x3 stride 1 / stride 2, pointwise), fp32 storage, one group, up to 32 samples that the chunk count divides into.
static bool wgrad_modulated_ok(const pasta_conv_desc* d, WgradPlan* out) {
    if (d->groups != 1 || d->io_dtype != PASTA_F32 || d->N < 1 || d->N > 32 || plan_wgrad_small(d).use) return false;
    const int Ig = d->C_in, Og = d->C_out;
    const WgradPlan w = d->transposed ? plan_wgrad(d->N, d->H, d->W, 1, Ig, Og, d->kh, d->kw, d->stride, d->N, wgrad_wide16(d))
                                      : plan_wgrad(d->N, d->OH, d->OW, 1, Og, Ig, d->kh, d->kw, d->stride, d->N, wgrad_wide16(d));
    if (!(wgrad_bf16x6(d, w) || wgrad_s2_bf16x6(d, w) || (wgrad_1x1_bf16x6(d, w) && !(w.TR == 3 && w.TS == 3) && w.TS != 7 && w.TS != 4))) return false;
    const int P = d->transposed ? d->H : d->OH;
    const int chh = w.kp >> w.cw_log2;
    if (P % chh != 0 || w.chunks_total % d->N != 0 || w.ksplit > w.chunks_total) return false;     // whole chunks per sample, at least one chunk per slice
    if (out) *out = w;
    return true;
}
// partial-ds blocks of wgrad_reduce_modulated_kernel: (16-row a blocks | 64-column b tiles when the modulated index is a) x taps
static int wgrad_modulated_blocks(const pasta_conv_desc* d, const WgradPlan& w) {
    const int Ap = w.a_tiles * 64 * w.WA, Bp = w.b_tiles * 64 * w.WB;
    return (d->transposed ? Bp / 64 : Ap / wgrad_mod_rows(Ap, Bp, d->kh * d->kw)) * d->kh * d->kw;
}
static int wgrad_run(const void* xv, const void* dyv, float* dw, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream,
                     const float* mod_s, const float* mod_w, float* ds);
}  // namespace pasta

extern "C" int64_t pasta_conv2d_wgrad_modulated_workspace(const pasta_conv_desc* d) {
    using namespace pasta;
    if (check_desc(d, "conv2d_wgrad_modulated_workspace")) return -1;
    WgradPlan w;
    if (!wgrad_modulated_ok(d, &w)) return -1;
    // [amax rows][slabs][partial ds: one [N][C_in] block per workgroup row of wgrad_reduce_modulated_kernel]
    return (WS_AMAX_FLOATS + w.slab_floats + (int64_t)wgrad_modulated_blocks(d, w) * d->N * d->C_in) * (int64_t)sizeof(float);
}

extern "C" int pasta_conv2d_wgrad(const void* xv, const void* dyv, float* dw, const pasta_conv_desc* d, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    return pasta::wgrad_run(xv, dyv, dw, d, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}

extern "C" int pasta_conv2d_wgrad_modulated(const void* x, const void* dy, const float* styles, const float* w, float* dw, float* dstyles,
                                            const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace pasta;
    PASTA_CHECK(styles && w && dstyles, "conv2d_wgrad_modulated: null pointer");
    return wgrad_run(x, dy, dw, d, workspace, workspace_bytes, stream, styles, w, dstyles);
}

int pasta::wgrad_run(const void* xv, const void* dyv, float* dw, const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream,
                     const float* mod_s, const float* mod_w, float* ds) {
    if (int e = check_desc(d, "conv2d_wgrad")) return e;
    const float* x = (const float*)xv; const float* dy = (const float*)dyv;       // elements of d->io_dtype behind these pointers
    PASTA_CHECK(x && dy && dw, "conv2d_wgrad: null pointer");
    const int64_t need = mod_s ? pasta_conv2d_wgrad_modulated_workspace(d) : pasta_conv2d_wgrad_workspace(d);
    PASTA_CHECK(need >= 0, "conv2d_wgrad_modulated: this shape has no sample-aligned split kernel (pasta_conv2d_wgrad_modulated_workspace tells beforehand)");
    PASTA_CHECK(workspace && workspace_bytes >= need, "conv2d_wgrad: workspace of %lld bytes needed, %lld given", (long long)need, (long long)workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int Ig = d->C_in / d->groups, Og = d->C_out / d->groups;
    PASTA_CHECK(((uintptr_t)workspace & 15) == 0, "conv2d_wgrad: workspace must be 16-byte aligned");
    float* const ws_amax = (float*)workspace;                         // [2][AMAX_PARTS]: x, dy
    workspace = (float*)workspace + WS_AMAX_FLOATS;                  // the partial slabs follow

    const WgradSmallPlan ws = mod_s ? WgradSmallPlan{} : plan_wgrad_small(d);
    if (d->x_layout == PASTA_LAYOUT_PIECES16) {
        int k = 0;
        if (int e = pasta_conv2d_wgrad_plan(d, &k)) return e;
        PASTA_CHECK(!mod_s && d->x_amax, "conv2d_wgrad: x_layout = PASTA_LAYOUT_PIECES16 needs x_amax (the row pasta_blur_pieces wrote) and plain weights");
    }
    PASTA_CHECK(d->io_dtype == PASTA_F32 || !ws.use || plan_wgrad1x1_fewcin(d, ws), "conv2d_wgrad: no 16-bit-storage kernel for this shape (pasta_conv2d_wgrad_plan tells beforehand)");
    if (const int fks = plan_wgrad1x1_fewcin(d, ws)) {
        // few input channels, 1x1: one bandwidth-bound pass over dy with plain FMAs (conv_wgrad_f32.h)
        const int64_t total = (int64_t)d->N * ((int64_t)d->H * d->W / 4);
        const int64_t per = (total + fks - 1) / fks;
        const int a_pad = ws.a_tiles * 64, bpad = ws.nb * 32;
        const dim3 grid((unsigned)fks, (unsigned)((d->C_out + 7) / 8));
        tu_wgrad1x1_fewcin(d->C_in, d->io_dtype, grid, dy, x, (float*)workspace, d->N, d->C_out, d->H * d->W, per, a_pad, bpad, s);
        tu_wgrad_smallcin_reduce((const float*)workspace, dw, fks, d->C_out, ws.bprime, a_pad, bpad, d->wscale == 0.f ? 1.f : d->wscale, s);
        return launch_status("conv2d_wgrad(few-channel 1x1)");
    }
    if (ws.use) {
        WgradSmallParams q;
        q.S = dy; q.L = x; q.slab = (float*)workspace;
        q.N = d->N; q.Ag = d->C_out; q.P = d->OH; q.Q = d->OW; q.Bg = Ig; q.LH = d->H; q.LW = d->W;
        q.kh = d->kh; q.kw = d->kw; q.pad_h = d->pad_h; q.pad_w = d->pad_w;
        q.bprime = ws.bprime; q.nb = ws.nb; q.cw_log2 = ws.cw_log2; q.rows_total = ws.rows_total; q.qblocks = ws.qblocks;
        q.chunks_total = ws.chunks_total; q.ksplit = ws.ksplit; q.a_tiles = ws.a_tiles;
        PASTA_CHECK(ws.lds_bytes <= 64 * 1024, "conv2d_wgrad: small-cin LDS footprint %zu too large", ws.lds_bytes);
        tu_wgrad_smallcin(q, ws.a_tiles * ws.ksplit, ws.lds_bytes, s);
        tu_wgrad_smallcin_reduce((const float*)workspace, dw, ws.ksplit, d->C_out, ws.bprime, ws.a_tiles * 64, ws.nb * 32, d->wscale == 0.f ? 1.f : d->wscale, s);
        return launch_status("conv2d_wgrad(small-cin)");
    }

    WgradParams p;
    p.slab = (float*)workspace;
    p.io = d->io_dtype;
    p.G = d->groups; p.kh = d->kh; p.kw = d->kw; p.st = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.N = d->N;
    if (!d->transposed) {   // dw[o][i]: S = dy, L = x
        p.S = dy; p.SC = d->C_out; p.P = d->OH; p.Q = d->OW; p.Ag = Og;
        p.L = x;  p.LC = d->C_in;  p.LH = d->H; p.LW = d->W; p.Bg = Ig;
    } else {                // dw[i][o]: S = x, L = dy
        p.S = x;  p.SC = d->C_in;  p.P = d->H; p.Q = d->W; p.Ag = Ig;
        p.L = dy; p.LC = d->C_out; p.LH = d->OH; p.LW = d->OW; p.Bg = Og;
    }
    const WgradPlan w = plan_wgrad(p.N, p.P, p.Q, p.G, p.Ag, p.Bg, p.kh, p.kw, p.st, mod_s ? d->N : 1, wgrad_wide16(d));
    p.cw_log2 = w.cw_log2; p.rows_total = w.rows_total; p.qblocks = w.qblocks; p.chunks_total = w.chunks_total;
    p.ksplit = w.ksplit; p.a_tiles = w.a_tiles; p.b_tiles = w.b_tiles; p.tap_groups_r = w.tgr; p.tap_groups_s = w.tgs;
    // measured (profiles/r3_ab_wgrad_xcd.txt): 256 -> 128 at 128^2 298.6 -> 303.4 TFLOP/s, stride 2 at 256^2 143 -> 154, at 257^2 140.7 -> 143.4,
    // every other shape within 0.5 %
    static const int wgrad_xcd = getenv("PASTA_WGRAD_XCD") ? atoi(getenv("PASTA_WGRAD_XCD")) : 1;
    p.xcd_order = wgrad_xcd;
    p.l_pieces = d->x_layout == PASTA_LAYOUT_PIECES16;
    PASTA_CHECK((int64_t)w.chunks_total * (w.ksplit + 1) < (1ll << 32), "conv2d_wgrad: %d chunks x %d K slices overflow the kernels' 32-bit slice bounds", w.chunks_total, w.ksplit);
    PASTA_CHECK(w.lds_bytes <= 160 * 1024, "conv2d_wgrad: LDS footprint %zu too large", w.lds_bytes);
    PASTA_CHECK(w.npos <= 256, "conv2d_wgrad: halo of %d positions per chunk is not supported", w.npos);

    const int64_t blocks = (int64_t)p.G * w.a_tiles * w.b_tiles * w.tgr * w.tgs * w.ksplit;
    PASTA_CHECK(blocks <= INT32_MAX, "conv2d_wgrad: grid too large");
    const int np = p.io != IO_F32 ? 1 : math_pieces(d->math);          // bf16 pieces per operand of the split-bf16 kernels (NP_F16X3: fp16 pieces)
    p.s_amax = p.l_amax = nullptr;
    if (np == NP_F16X3 && (wgrad_bf16x6(d, w) || wgrad_s2_bf16x6(d, w) || (wgrad_1x1_bf16x6(d, w) && !(w.TR == 3 && w.TS == 3) && w.TS != 7 && w.TS != 4))) {
        const float* xa = d->x_amax; const float* ya = d->dy_amax;
        if (!xa) { if (int e = tensor_amax(x, (int64_t)d->N * d->C_in * d->H * d->W, PASTA_F32, ws_amax, s)) return e; xa = ws_amax; }      // (pieces layout: given, checked above)
        if (!ya) { if (int e = tensor_amax(dy, (int64_t)d->N * d->C_out * d->OH * d->OW, PASTA_F32, ws_amax + AMAX_PARTS, s)) return e; ya = ws_amax + AMAX_PARTS; }
        p.s_amax = d->transposed ? xa : ya;
        p.l_amax = d->transposed ? ya : xa;
    }
    PASTA_CHECK(p.io == IO_F32 || wgrad_bf16x6(d, w) || wgrad_s2_bf16x6(d, w) || (wgrad_1x1_bf16x6(d, w) && !(w.TR == 3 && w.TS == 3) && w.TS != 7 && w.TS != 4),
                "conv2d_wgrad: no 16-bit-storage kernel for this shape (pasta_conv2d_wgrad_plan tells beforehand)");
    const bool split1x1 = wgrad_1x1_bf16x6(d, w) && !(w.TR == 3 && w.TS == 3) && w.TS != 7 && w.TS != 4;
    if (wgrad_bf16x6(d, w)) tu_wgrad3x3(np, p, blocks, s);
    else if (wgrad_s2_bf16x6(d, w)) tu_wgrad3x3s2(np, p, blocks, s);
    else if (split1x1) tu_wgrad1x1(np, w.WA, p, blocks, s);
    else if (int e = tu_wgrad_f32(w.TR, w.TS, w.WA, w.pipe, w.kp, p, blocks, w.lds_bytes, s)) return e;
    if (mod_s) {
        // slices [n m, (n + 1) m) hold sample n's gradient with respect to the modulated weight: dw = sum_n s[n, i] (.), ds[n, i] = sum_{o, taps} w (.)
        const int Ap = w.a_tiles * 64 * w.WA, Bp = w.b_tiles * 64 * w.WB;
        float* const dsp = (float*)workspace + w.slab_floats;
        const float wsc = d->wscale == 0.f ? 1.f : d->wscale;
        const int wg_rows = wgrad_mod_rows(Ap, Bp, p.kh * p.kw);
        const dim3 grid((unsigned)(Bp / 64), (unsigned)(Ap / wg_rows), (unsigned)(p.kh * p.kw));
        tu_wgrad_reduce_modulated(d->transposed != 0, grid, (const float*)workspace, mod_s, mod_w, dw, dsp, w.ksplit, d->N, p.Ag, p.Bg, Ap, Bp, p.kh, p.kw, d->flip, wsc, wg_rows, s);
        const int tiles = wgrad_modulated_blocks(d, w);
        const int nc = d->N * d->C_in;
        tu_sum_blocks((const float*)dsp, ds, tiles, nc, s);
        return launch_status("conv2d_wgrad_modulated");
    }
    tu_wgrad_reduce((const float*)workspace, dw, w.ksplit, p.G, p.Ag, p.Bg, w.a_tiles * 64 * w.WA, w.b_tiles * 64 * w.WB, p.kh, p.kw, d->flip,
                    d->wscale == 0.f ? 1.f : d->wscale, s);
    return launch_status("conv2d_wgrad");
}
