/*
 * pasta_hip.h -- C ABI of libpasta_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the PASTA-GAN generator/discriminator hot path.  Every
 * entry point replaces one native (or ATen-delegated) call of the reference;
 * the reference interface it stands in for is cited per function as
 * <file>:<line> relative to the reference repository root.
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes, no torch types.  Device pointers unless noted.
 *   - the CALLER allocates every output and workspace; the library never
 *     allocates device memory and keeps no mutable global state after load.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); the
 *     caller has already selected the device.
 *   - return 0 on success, non-zero on error; pasta_last_error() then returns a
 *     thread-local, NUL-terminated description (the Python shim raises
 *     RuntimeError with it, mirroring TORCH_CHECK in the reference wrappers).
 *   - dtype codes: PASTA_F32 = 0, PASTA_F16 = 1, PASTA_F64 = 2, PASTA_BF16 = 3.
 */
#ifndef PASTA_HIP_H
#define PASTA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PASTA_F32 = 0, PASTA_F16 = 1, PASTA_F64 = 2, PASTA_BF16 = 3 };

/* Library identification / error reporting. */
const char* pasta_last_error(void);
int         pasta_abi_version(void);           /* bumps when a signature changes   */
const char* pasta_build_info(void);            /* "gfx950 <date> <hip version>"    */

/* ------------------------------------------------------------------------- *
 * upfirdn2d -- pad, zero-stuff upsample, 2-D FIR, decimate (one launch).
 * Replaces: upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0,
 *           padx1, pady0, pady1, flip, gain)  torch_utils/ops/upfirdn2d.cpp:16
 *           (kernels torch_utils/ops/upfirdn2d.cu:29-200).
 * x: [N,C,inH,inW] with element strides in_stride[4] (N,C,H,W order);
 * f: float32 [fH,fW] dense row-major (always fp32, upfirdn2d.cpp:21);
 * y: [N,C,outH,outW] with out_stride[4]; outH/outW must equal
 *    (in*up + pad0 + pad1 - f + down) / down  (upfirdn2d.cpp:32-33).
 * ------------------------------------------------------------------------- */
int pasta_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                    const int32_t in_size[4], const int64_t in_stride[4],
                    const int32_t f_size[2],
                    const int32_t out_size[4], const int64_t out_stride[4],
                    int upx, int upy, int downx, int downy,
                    int padx0, int padx1, int pady0, int pady1,
                    int flip, float gain, void* stream, float* y_amax,
                    const void* y_add);     /* ABI 18, optional: a tensor with y's shape, strides and type, added to the result on its way out.  The
                                               gradient of upfirdn2d is upfirdn2d (upfirdn2d.py:246-264): when x has further consumers their gradient
                                               rides here instead of in an addition pass over two tensors */

/* -------------------------------------------------------------------------
 * Producer-written operand pieces (ABI 19).  Replaces, for the low-pass in front of a stride-2 convolution
 * (torch_utils/ops/conv2d_resample.py:119-122: `x = upfirdn2d(x, f, padding)` then `conv2d(x, w, stride=2)`), the fp32 NCHW
 * intermediate of the reference by the matrix-core operand itself: y = upfirdn2d(x, f, up = down = 1, padding, flip, gain) for a
 * 4x4 filter f (fp32 [4][4] on the device) is written ONCE as PASTA_LAYOUT_PIECES16 -- units [N][C / 8][OH][2][OW] of 16 bytes: eight fp16
 * values of eight consecutive channels, piece 0 = h = fp16(v S), piece 1 = l' = fp16(2^11 (v S - h)) -- and the consumers (pasta_conv2d_ex /
 * pasta_conv2d_wgrad with pasta_conv_desc.x_layout = PASTA_LAYOUT_PIECES16) copy sixteen-byte pieces into LDS instead of gathering
 * channel-strided fp32 and splitting it at every launch.  OH = H + pady0 + pady1 - 3, OW likewise; the fp32 value in front of the split
 * is bit-identical to pasta_upfirdn2d's.
 *   x_amax  (in)  PASTA_AMAX_PARTS partial |max| of x (its producer's row, or pasta_tensor_amax).
 *   y_amax  (out) PASTA_AMAX_PARTS floats: x_amax times gain * sum |f| -- a bound of |y| known BEFORE the blur runs.  The power of two S
 *                 comes from this row on both sides (pass it as pasta_conv_desc.x_amax); being a power of two, the consumers' results do
 *                 not depend on which admissible S was used (tests/test_pieces_gpu.py scales the row by 2 and by 1/2).
 * pasta_pieces_bytes: size of the piece tensor (-1: C is not a multiple of 8).  pasta_pieces_unpack: (h + 2^-11 l') / S back to fp32
 * NCHW (tests and diagnostics: 22 of the 24 bits).
 * ------------------------------------------------------------------------- */
int64_t pasta_pieces_bytes(int N, int C, int H, int W);
int pasta_blur_pieces(const float* x, const float* f, void* pieces, const float* x_amax, float* y_amax,
                      int N, int C, int H, int W, int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream);
int pasta_pieces_unpack(const void* pieces, const float* y_amax, float* y, int N, int C, int H, int W, void* stream);
/* (ABI 20) an fp32 NCHW tensor whose partial maxima are known, split as the consuming kernels split it in their staging (bit for bit): for a
 * caller whose producer is not one of this library's kernels, and for the tests of the pieces-reading kernels. */
int pasta_pieces_pack(const float* x, const float* x_amax, void* pieces, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------- *
 * bias_act -- fused bias + activation + gain + clamp, and its 1st/2nd grads.
 * Replaces: bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act,
 *           alpha, gain, clamp)  torch_utils/ops/bias_act.cpp:32
 *           (kernel torch_utils/ops/bias_act.cu:23-147).
 * All tensors are dense with identical layout, n elements; NULL = absent
 * (the reference passes an empty tensor).  b has size_b elements and is
 * indexed by (i / step_b) % size_b.  act = 1..9 (bias_act.py:23-33 cuda_idx).
 * grad = 0 forward, 1 first derivative (x is dy), 2 second derivative.
 * clamp < 0 disables clamping.
 * ------------------------------------------------------------------------- */
int pasta_bias_act(const void* x, const void* b, const void* xref,
                   const void* yref, const void* dy, void* y, int dtype,
                   int64_t n, int size_b, int64_t step_b, int grad, int act,
                   float alpha, float gain, float clamp, void* stream, float* y_amax);

/* Column sums used for the bias gradient (bias_act.py:173 `dx.sum(...)`):
 * db[c] = sum over all i with (i / step_b) % size_b == c of dx[i].
 * fp32 accumulate; db has dtype of dx.  work: size_b*nsplit floats scratch
 * (nsplit returned by pasta_bias_grad_workspace). */
int64_t pasta_bias_grad_workspace(int64_t n, int size_b, int64_t step_b);
int pasta_bias_grad(const void* dx, void* db, float* work, int dtype, int64_t n,
                    int size_b, int64_t step_b, void* stream);

/* First derivative and bias gradient in one pass over HBM (the pair bias_act.py:162-173 `dx = plugin.bias_act(dy, ...,
 * grad=1, ...)` + `db = dx.sum(...)`):  dx = dy * act'(yref / gain) * gain, zero where |yref| >= clamp >= 0, and
 * db[c] = sum of dx over its (n, c) planes (fp32 accumulate, fixed order).  act 1..3 (linear, relu, lrelu), fp32 or
 * fp16, tensors viewed as [outer, size_b, step_b] with step_b a multiple of 16 bytes and >= 256 packs of 16 bytes.
 * pasta_bias_act_grad_db_workspace returns the bytes of `work`, or 0 when the case is not covered (then call
 * pasta_bias_act(grad=1) followed by pasta_bias_grad).  yref may be NULL only for act 1 without clamp. */
int64_t pasta_bias_act_grad_db_workspace(int dtype, int64_t n, int size_b, int64_t step_b, int act);
int pasta_bias_act_grad_db(const void* dy, const void* yref, void* dx, void* db, float* work, int dtype, int64_t n,
                           int size_b, int64_t step_b, int act, float alpha, float gain, float clamp, void* stream, float* dx_amax);

/* ------------------------------------------------------------------------- *
 * Dense convolution family on fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 * Replaces the ATen/cuDNN calls behind torch_utils/ops/conv2d_gradfix.py:38,43
 * (forward), :125-128 (input gradient) and :140-148 (weight gradient).
 *
 * One descriptor covers conv2d, conv_transpose2d and both of their gradients:
 *   y[n, g*Og + o, oy, ox] = sum_{i,r,s} x[n, g*Ig + i, iy, ix] * w[...]
 * `transposed` selects conv_transpose2d semantics (weight [I, O/g, kh, kw]).
 * ------------------------------------------------------------------------- */
typedef struct pasta_conv_desc {
    int32_t N, C_in, H, W;        /* input  x: [N, C_in, H, W]  contiguous NCHW      */
    int32_t C_out, OH, OW;        /* output y: [N, C_out, OH, OW] contiguous NCHW    */
    int32_t kh, kw;               /* kernel size                                      */
    int32_t stride;               /* conv stride (conv2d) or upsampling (transposed) */
    int32_t pad_h, pad_w;         /* symmetric zero padding                           */
    int32_t groups;               /* 1 (training) or N-style grouped (eval modconv)   */
    int32_t transposed;           /* 0 = conv2d, 1 = conv_transpose2d                 */
    int32_t flip;                 /* 1 = true convolution (flip taps), 0 = correlation */
    int32_t math;                 /* PASTA_MATH_*: arithmetic of the matrix-core products (see below)  */
    float   wscale;               /* the weights are used as w * wscale (0 = 1): Conv2dLayer's `self.weight *
                                     self.weight_gain` (networks.py:171) folded into the weight packing; the weight
                                     gradient is returned with respect to the unscaled w (i.e. times wscale)        */
    int32_t io_dtype;             /* storage type of x, y, dy (and the fused residual): PASTA_F32 (default), PASTA_F16 or
                                     PASTA_BF16.  With 16-bit storage the stored element is the matrix-core operand: ONE
                                     product per multiply-add (v_mfma_f32_32x32x16_f16 / _bf16), fp32 accumulation and
                                     epilogue, one rounding on the way out -- the arithmetic of the reference's fp16
                                     blocks (networks.py:1107-1120) and of BASELINE config 5.  Weights, weight gradients,
                                     bias and the scale vectors stay fp32.  Only shapes the matrix-core kernels cover
                                     (>= 16 input channels per group, > 32 output channels; 3x3 / 1x1 weight gradients on
                                     rows of a multiple of 16 / 32 pixels): pasta_conv2d_plan / pasta_conv2d_wgrad_plan
                                     return an error otherwise and the caller converts that launch to fp32.            */
    const float* x_amax;          /* PASTA_MATH_F16X3 only, optional: PASTA_AMAX_PARTS partial |max| of the tensor passed as x
                                     (pasta_tensor_amax wrote them).  NULL: the launch computes them itself into its workspace
                                     (one extra pass over x).  A caller that uses a tensor in several launches (forward and
                                     weight gradient; input gradient and weight gradient) computes them once.            */
    const void*  x2;              /* optional second input tensor of a POINTWISE convolution (kh = kw = 1, stride 1, fp32 storage, PASTA_MATH_F16X3):
                                     x holds input channels [0, C1), x2 ([N, C_in - C1, H, W], contiguous) channels [C1, C_in) -- the convolution of
                                     torch.cat([x, x2], 1) (the merge layers, networks.py:5698-5700) without forming it.  NULL: one tensor.  An error
                                     where the pointwise kernel does not apply (pasta_conv2d_plan reports kernel 9 where it does)           */
    const float* x2_amax;         /* partial |max| of x2, as x_amax                                                      */
    int32_t      C1;              /* channels held by x when x2 is given                                                  */
    const float* dy_amax;         /* the same for dy (pasta_conv2d_wgrad only).  The WEIGHTS need nothing of the kind (ABI 17): their
                                     packing kernel finds one scale per output row itself at every launch, so no |max| of w is
                                     passed, cached or trusted across launches                                            */
    int32_t      x_layout;        /* ABI 19.  0 = x is a contiguous NCHW tensor of io_dtype (everything above).  PASTA_LAYOUT_PIECES16 = x is the
                                     producer-written operand of the three-product arithmetic (pasta_blur_pieces below): [N][C_in / 8][H][2][W] units of
                                     16 bytes, fp16 h[8] / l'[8] of v S, and x_amax (REQUIRED then) is the 256-float row the producer wrote -- the bound
                                     both sides take the power-of-two scale S from.  Served by the 3x3 stride-2 forward kernel and its weight gradient
                                     (pasta_conv2d_plan kernel 10, pasta_conv2d_wgrad_plan kernel 6: fp32 y / dy, PASTA_MATH_F16X3, one group, C_in a
                                     multiple of 8, pad 0) and by the eight-wave 3x3 stride-1 tile kernel, convolution or input gradient
                                     (pasta_conv2d_plan kernel 7: no input scale, one group, C_in a multiple of 8; pasta_pieces_pack writes such an
                                     operand from an fp32 tensor); the planners return an error for every other launch and the caller keeps the
                                     fp32 tensor. */
    int32_t      w_prepacked;     /* ABI 21.  1 = the workspace already holds this launch's packed weights and row scales (pasta_conv2d_pack_pair wrote them
                                     for THIS descriptor and these weights): the launch skips its packing kernel.  0 everywhere else.                 */
} pasta_conv_desc;

#define PASTA_LAYOUT_NCHW      0
#define PASTA_LAYOUT_PIECES16  1

/* Arithmetic of the convolution products.  Accumulation is fp32 in every mode.
 *   PASTA_MATH_F32    v_mfma_f32_32x32x2_f32: every product and sum is an fp32 FMA (bit-exact fp32 chains).
 *   PASTA_MATH_BF16X6 each fp32 operand is split into three bf16 pieces (24 significand bits) and a*b is formed
 *                     from six exact bf16 x bf16 products on v_mfma_f32_32x32x16_bf16: fp32-equivalent accuracy
 *                     (dropped terms < 2^-22 |ab|) at 2.67x the fp32 matrix-core rate.  Used by the 128x128- and
 *                     64x256-tile forward / input-gradient launches with >= 16 input channels per group and
 *                     by the 3x3 stride-1 weight gradient; all other launches run PASTA_MATH_F32.
 *   PASTA_MATH_BF16X3 two bf16 pieces per operand, three products (hi*hi, hi*mid, mid*hi): ~2^-16 relative error per
 *                     product, half of the matrix work -- the counterpart of the reference's `allow_tf32=True`
 *                     (training_loop_wo_flow_fullbody.py:247-249; TF32 keeps 2^-11).  Opt-in, never the default.
 *   PASTA_MATH_BF16   operands rounded to bf16, one product, fp32 accumulate and fp32 tensors in HBM: the arithmetic of
 *                     mixed-precision training (BASELINE config 5).  Opt-in.
 *   PASTA_MATH_F16X3  fp32-equivalent products from THREE fp16 products (v_mfma_f32_32x32x16_f16): each operand is scaled
 *                     by a power of two taken from its tensor's largest magnitude and split into fp16 pieces h + 2^-11 l'
 *                     (22 + 1 significand bits: representation error <= 2^-23, fp32's own rounding is 2^-24); h h, h l'
 *                     and l' h are exact in fp32 and accumulate in fp32; the result is scaled back exactly.  Half the
 *                     matrix work of BF16X6 at the same accuracy class (rms error against fp64 within 10 % of an fp32 FMA
 *                     chain's: the fp32 accumulation dominates both).  Range: activations keep full precision down to 2^-28
 *                     of their tensor's largest element; weights are scaled PER OUTPUT ROW (output channel of the launch) and keep
 *                     full precision down to 2^-16 of their row's largest element; both operands of a weight gradient down to
 *                     2^-17 of their tensor's; smaller elements contribute with an absolute error <= 2^-28 amax each.  Non-finite
 *                     elements are skipped by the scale and stay local.  fp32 storage only; same kernels and coverage as BF16X6,
 *                     per-sample modulated weights (pasta_conv2d_modulated) included since ABI 17.
 *   The split modes share kernels (template argument NP = pieces) and the same coverage.
 *   PASTA_MATH_DEFAULT = PASTA_MATH_F16X3 (round 3; BF16X6 before). */
enum { PASTA_MATH_DEFAULT = 0, PASTA_MATH_F32 = 1, PASTA_MATH_BF16X6 = 2, PASTA_MATH_BF16X3 = 3, PASTA_MATH_BF16 = 4, PASTA_MATH_F16X3 = 5 };

/* Largest finite magnitude of a contiguous fp32 tensor as PASTA_AMAX_PARTS partial maxima (parts[i] >= 0; the maximum over
 * i is the tensor's): the operand scales of PASTA_MATH_F16X3.  One pass at HBM rate, no atomics, no host round trip;
 * non-finite elements are skipped.  parts 16-byte aligned.  No reference counterpart (the reference hands fp32
 * tensors to cuDNN, conv2d_gradfix.py:38); it exists so that a tensor used by several launches is scanned once. */
#define PASTA_AMAX_PARTS 256
int pasta_tensor_amax(const void* x, int64_t numel, int dtype, float* parts, void* stream);
/* Producer-side maxima: the operators that WRITE activation tensors (pasta_upfirdn2d, pasta_bias_act, pasta_bias_act_grad_db,
 * pasta_scale_add, pasta_mod_bias_act(_bwd), pasta_spade_norm(_bwd) and the convolutions through pasta_conv_epilogue.y_amax) take
 * a last argument `float* y_amax` (NULL = off): PASTA_AMAX_PARTS floats ZEROED by the caller, into which the kernel leaves
 * partial maxima of the largest finite magnitude it stored (one agent-scope integer atomic per wavefront on the bit patterns:
 * order-independent, hence deterministic).  What pasta_tensor_amax would find, without the extra pass over the tensor. */

/* Bytes of scratch the forward / weight-gradient launches need (caller allocs). */
int64_t pasta_conv2d_workspace(const pasta_conv_desc* d);
int64_t pasta_conv2d_wgrad_workspace(const pasta_conv_desc* d);

/* Which forward-type kernel instance the launch will use: 0 = 128x128 tile, 1 = 64x256,
 * 2 = 32x256, 3 = 64x64 (rows = output channels, columns = pixels).  Reporting only. */
int pasta_conv2d_tile(const pasta_conv_desc* d);

/* (ABI 21) One weight tensor packed for TWO launches by ONE kernel -- typically a convolution and its input gradient (the same weights, transposed and
 * mirrored), so that the backward pass finds its operand packed: 120 of the 310 packing launches of a training step.  ws_a / ws_b: the workspaces the two
 * launches will be given (pasta_conv2d_workspace(da / db) bytes); *packed_mask = 3 when both were packed (set pasta_conv_desc.w_prepacked = 1 in both
 * launches), 0 when the pair is not served (16-bit storage, another arithmetic than PASTA_MATH_F16X3, few-channel or packed-K launches, scale vectors or
 * modulated weights are the CALLER's business: it must not ask for those) -- the launches then pack for themselves as always. */
int pasta_conv2d_pack_pair(const float* w, const pasta_conv_desc* da, void* ws_a, int64_t ws_a_bytes, const pasta_conv_desc* db, void* ws_b, int64_t ws_b_bytes,
                           void* stream, int* packed_mask);

/* The full launch plan of pasta_conv2d(_ex) for d, for reporting (bench.py attributes time and FLOPs to kernel
 * families with it): *tile as pasta_conv2d_tile, *ksplit = number of K slices (> 1: partial sums in the workspace,
 * reduced by a second kernel), *math = PASTA_MATH_F32 or PASTA_MATH_BF16X6 actually used (launch_flags = OR of PASTA_PLAN_*:
 * what the launch will pass besides x, w, y -- an iscale vector, which the split-bf16 kernels take in their staging for fp32
 * storage and six products only; an oscale vector; a fused epilogue.  0 / 1 keep their round-2 meaning), *launches = launches of the main kernel (conv_transpose2d: one per output parity class unless the classes share a
 * grid), *kernel = 0 conv_fwd_kernel (fp32 MFMA), 1 conv_fwd_bf16x6_kernel, 2 conv_fwd_rows_bf16x6_kernel (split-bf16 with
 * row reuse: 3-wide stride-1 kernels on planes whose rows are a multiple of 32 pixels), 3 the same kernel's parity-pair
 * mode (3x3 stride-2 conv_transpose2d onto 2H(+1) x 2W(+1) outputs: one launch over the input lattice + one small
 * launch of kernel 1 for the last row / column), 4 / 5 / 6 conv_fwd_rows2d_bf16x6_kernel<128,128,4>, <128,128,2>, <64,256,8>: R output rows per
 * pixel tile (3x3 stride-1 lattices on the 128 x 128 tile whose planes divide into R x 128/R tiles: the R + 2 input rows of a
 * tile are staged once per 16-channel chunk), 7 the same kernel on eight waves and a 128 x 256 tile (<128,256,8,3,0,false,512>: plain
 * six-product fp32 launches on planes of a multiple of 8 rows; round 3: the launch of kernel 3's last row / column is
 * conv_t2_edge_kernel, and kernels 1 - 7 also run PASTA_MATH_F16X3), 8 conv_fwd_bf16x6_kernel in its packed-K mode (round 3: fewer than
 * 16 input channels, more than 32 output channels, at least 64 (channel, tap) pairs, planes above 8192 pixels -- the 7x7 RGB stems:
 * K runs over the pairs; the workspace then also holds the offset table and a zero-padded copy of the input, and two small
 * kernels fill them), 9 conv1x1_f16x3_kernel (round 4: 1x1 stride-1 convolutions and their input gradients under PASTA_MATH_F16X3, fp32 tensors,
 * >= 16 input and > 32 output channels, planes of a multiple of 128 / 256 pixels: 16-byte loads along the pixels, 32 channels per barrier pair,
 * optionally over two input tensors -- pasta_conv_desc.x2), 10 conv3x3s2_f16x3_kernel (round 4: 3x3 stride-2 conv2d with pads 0 / 1 under PASTA_MATH_F16X3,
 * fp32 tensors, >= 16 input and > 32 output channels, output widths 16 .. 128 .. that are powers of two: a round stages the input ROW segments of a kernel
 * row once, de-interleaved by pixel parity, for its three taps; round 5: with pasta_conv_desc.x_layout = PASTA_LAYOUT_PIECES16 its staging is a copy of
 * the producer's sixteen-byte pieces, no split), 11 / 12 conv1x1_fewcin_kernel / conv1x1_fewcout_kernel (round 5: 1x1 stride-1 launches with <= 16 input
 * or <= 16 output channels over more than 8192 pixels, fp32 tensors, one group, planes of a multiple of four pixels -- the RGB / pose stems, the ToRGB and
 * parsing heads and their input gradients: streaming kernels of plain fp32 FMAs on the raw weights, no packing launch, *math = PASTA_MATH_F32),
 * 13 conv_t2_f16x3_kernel (round 5: 3x3 stride-2 conv_transpose2d with pad 0 onto 2H(+1) x 2W(+1) outputs under PASTA_MATH_F16X3, fp32 tensors, >= 16
 * input channels, input planes of 8 x 32 or of 16 x 16 tiles, optionally an input scale, nothing behind the sum: ONE launch over the input
 * lattice computes the four output parity classes of a tile from one staged (8 + 1) x (32 + 1) window image -- nine taps, weights by LDS-DMA --
 * and the remainder row / column as edge tiles of the same grid; *launches = 1; the workspace also holds the input's last column, gathered by a
 * small kernel in front).  Kernel 7 also takes pasta_conv_desc.x_layout = PASTA_LAYOUT_PIECES16 (round 5).
 * Any out pointer may be NULL. */
#define PASTA_PLAN_ISCALE   1
#define PASTA_PLAN_OSCALE   2
#define PASTA_PLAN_EPILOGUE 4
#define PASTA_PLAN_MODULATED 8   /* pasta_conv2d_modulated (per-group modulated weights) */
int pasta_conv2d_plan(const pasta_conv_desc* d, int launch_flags, int* tile, int* ksplit, int* math, int* launches, int* kernel);

/* Same for pasta_conv2d_wgrad: *kernel = 0 conv_wgrad_kernel (fp32 MFMA, taps x 64 x 64 tiles), 1
 * conv_wgrad_smallcin_kernel (<= 8 input channels: (channel, tap) pairs as GEMM columns), 2
 * conv_wgrad3x3_bf16x6_kernel (split-bf16; 3x3, stride 1, pad 1, row length a multiple of 32 -- round 5: or exactly 16, as half-filled chunks), 3
 * conv_wgrad3x3s2_bf16x6_kernel (split-bf16; 3x3, stride 2, pad 0 or 1, row length a multiple of 16), 4
 * conv_wgrad1x1_bf16x6_kernel (split-bf16; 1x1, stride 1, planes of a multiple of 32 pixels, >= 16 channels), 5
 * wgrad1x1_fewcin_kernel (round 4: 1x1, <= 8 input channels, planes of a multiple of 4 pixels: one bandwidth-bound fp32 pass over dy), 6
 * conv_wgrad3x3s2_pieces_kernel (round 5: kernel 3's shapes with pad 0 and x given as PASTA_LAYOUT_PIECES16: the x halo is copied into a
 * [pixel][channel] LDS image and the stride-2 tap operands are gathered by ds_read_b64_tr_b16, no split and no permutes). */
int pasta_conv2d_wgrad_plan(const pasta_conv_desc* d, int* kernel);

/* y = conv(x, w).  w is the PyTorch-layout weight ([C_out, C_in/g, kh, kw], or
 * [C_in, C_out/g, kh, kw] when transposed).  Optional fused epilogue:
 *   y = y * oscale[n, c] (NULL = 1)  -- demodulation, networks.py:77-79
 * and optional fused prologue on x:
 *   x'[n, c, :, :] = x * iscale[n, c] (NULL = 1) -- modulation, networks.py:74. */
int pasta_conv2d(const void* x, const float* w, void* y,      /* x, y: elements of d->io_dtype; w: fp32 */
                 const float* iscale, const float* oscale,
                 const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* Optional fused epilogue of pasta_conv2d_ex: after the output scale,
 *   y = clamp(act(y + noise * noise_strength[0] + res + bias[c]) * gain)
 * -- Conv2dLayer's bias_act (training/networks.py:176-178) and, with oscale = the demodulation coefficients and the noise
 * operand, the whole tail of SynthesisLayer (networks.py:77-82, 313-314: x * dcoefs + noise, bias_act) in the convolution's
 * own epilogue; forward only (the training path keeps pasta_mod_bias_act, whose backward needs the convolution output).
 * act: 1 linear, 2 relu, 3 lrelu (bias_act.py:24-26); bias NULL = none; clamp < 0 = none. */
typedef struct pasta_conv_epilogue {
    const float* bias;            /* [C_out] or NULL */
    int32_t act;
    float alpha, gain, clamp;
    const void* res;              /* (elements of d->io_dtype) [N, C_out, OH, OW] added to the convolution BEFORE bias / activation, or NULL
                                     (residual sums and the halves of a convolution over a channel concatenation without a
                                     pass of their own).  Measured on the two uses this path offers -- merge_conv over
                                     torch.cat (networks.py:5690-5693) as two 1x1 convolutions, and the SPADE block's
                                     y + conv(x) (:5273) -- it is time-neutral (+0.4 % / 0.0 %), so the networks keep the
                                     reference's formulation and the operand stays an option of the operator. */
    const float* noise;           /* fp32 [OH*OW] (noise_per_sample 0) or [N][OH*OW] (1), or NULL */
    const float* noise_strength;  /* device scalar (SynthesisLayer.noise_strength); required with noise */
    int32_t noise_per_sample;
    float* y_amax;                /* optional: PASTA_AMAX_PARTS floats, ZEROED by the caller, that receive partial maxima of the
                                     largest finite |y| (fp32 storage): the launch that consumes y under PASTA_MATH_F16X3 then needs
                                     no scan of y (see "producer-side maxima" below).  NULL = off. */
} pasta_conv_epilogue;

/* pasta_conv2d with the epilogue above (ep NULL = plain pasta_conv2d). */
int pasta_conv2d_ex(const void* x, const float* w, void* y,
                    const float* iscale, const float* oscale, const pasta_conv_epilogue* ep,
                    const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes,
                    void* stream);

/* Modulated convolution in its per-sample-weight form (networks.py:84-94, the `fused_modconv` branch test.py runs):
 * d describes the grouped convolution the reference launches (groups = N samples, C_in = N*I, C_out = N*O, x viewed as
 * [1, N*I, H, W]); w is the ONE shared weight [O, I, kh, kw] ([I, O, kh, kw] when d->transposed).  The weight-packing
 * kernel forms each group's operand  w[o,i,:] * styles[n,i] * dcoefs[n,o]  on its way into the staging layout, so the
 * [N, O, I, kh, kw] tensor of the reference and the passes that build it do not exist (dcoefs NULL = no demodulation;
 * pasta_demod_coefs computes them).  Forward only. */
int pasta_conv2d_modulated(const void* x, const float* w, const float* styles, const float* dcoefs, void* y,
                           const pasta_conv_epilogue* ep, const pasta_conv_desc* d, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* dw = d(conv)/dw given x and dy (same descriptor as the forward); x, dy: elements of d->io_dtype, dw: fp32. */
int pasta_conv2d_wgrad(const void* x, const void* dy, float* dw,
                       const pasta_conv_desc* d, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* Weight gradient AND style gradient of the shared-weight modulated convolution  y = conv(x * styles[n, i], w)  (training/networks.py:72-76, the
 * training branch of modulated_conv2d) WITHOUT the tensor x * styles: the weight-gradient kernels run on the unmodulated x with K slices that do
 * not straddle samples, and the reduction forms  dw[o,i,t] = wscale sum_n styles[n,i] Dw_n[o,i,t]  and  dstyles[n,i] = sum_{o,t} wscale w[o,i,t] Dw_n[o,i,t]
 * (Dw_n: sample n's gradient with respect to the weight it saw).  With the forward launched as pasta_conv2d(x, w, iscale = styles) and the input
 * gradient as pasta_conv2d(dy, w, oscale = styles) the reference's  x * styles  (one pass to form it, its saved copy, one pass to scale the input
 * gradient back, two reads for sum_hw dx x) is gone from the training step.  d, x, dy, x_amax / dy_amax as for pasta_conv2d_wgrad (x_amax: of the
 * UNMODULATED x); styles: [N, C_in] fp32; w: the weight; dw: its gradient; dstyles: [N, C_in].  fp32 storage, groups == 1, N <= 32, the split kernels'
 * shapes (pasta_conv2d_wgrad_plan kernels 2 - 4): pasta_conv2d_wgrad_modulated_workspace returns -1 where it does not apply (ABI 18). */
int64_t pasta_conv2d_wgrad_modulated_workspace(const pasta_conv_desc* d);
int pasta_conv2d_wgrad_modulated(const void* x, const void* dy, const float* styles, const float* w, float* dw, float* dstyles,
                                 const pasta_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Modulated-convolution helpers (training/networks.py:36-94).
 * ------------------------------------------------------------------------- */
/* Demodulation coefficients (networks.py:65-68):
 *   d[n,o] = rsqrt(sum_{i,k} (w[o,i,k] * styles[n,i])^2 + eps),  w: [O,I,KK] fp32, styles: [N,I] fp32, d: [N,O] fp32.
 * One workgroup per output channel, tap-summed squared weights in LDS, wavefront-shuffle reduction over I (I <= 4096);
 * the reference's per-sample weight tensor [N,O,I,kh,kw] is never formed. */
int pasta_demod_coefs(const float* w, const float* styles, float* d, int N, int O, int I, int KK,
                      float eps, void* stream);

/* y[n,c,h,w] = x[n,c,h,w] * a[n,c] + (b ? b[n,0,h,w] : 0)   fma.py:15 with the
 * broadcast shapes modulated_conv2d uses (a: [N,C], b: [N,HW] or [HW], or NULL) */
/* (this and the following plane kernels: `dtype` = storage type of the activation tensors -- PASTA_F32, PASTA_F16 or
 * PASTA_BF16; per-channel scales, statistics, bias, noise strength and partial sums are always fp32, as is the arithmetic) */
int pasta_scale_add(const void* x, const float* a, const void* b, void* y, int dtype,
                    int N, int C, int64_t HW, int b_per_sample, void* stream, float* y_amax);

/* Per-(n,c) plane reductions used by fma / modulation backward:
 * out[n,c] = sum_hw p[n,c,hw] * q[n,c,hw]   (q NULL => sum of p) */
int pasta_plane_dot(const void* p, const void* q, float* out, int dtype, int64_t planes,
                    int64_t HW, void* stream);

/* Tail of SynthesisLayer in one pass (networks.py:72-82 demodulation + noise, :313-314 bias_act):
 *   y = clamp(act(u * d[n,c] + noise * strength[0] + b[c]) * gain),  act = 1 (linear) or 3 (lrelu),
 * u: [N,C,HW] convolution output, d: [N,C] demodulation coefficients (NULL = 1), noise: [N,HW] (noise_per_sample)
 * or [HW] or NULL, strength: device scalar, b: [C] or NULL, clamp < 0 = none.
 * Backward: du = dz * d with dz = dy * act'(y) * gain (0 where |y| >= clamp), and per (plane, 4096-element chunk)
 * the triple (sum dz*u, sum dz*noise, sum dz) in `partial` ([N*C][chunks][3] floats, pasta_mod_bias_act_bwd_workspace
 * bytes), from which the caller forms dd[n,c], dstrength and db[c]. */
int pasta_mod_bias_act(const void* u, const float* d, const float* noise, const float* strength, const float* b, void* y,
                       int dtype, int N, int C, int64_t HW, int noise_per_sample, int act, float alpha, float gain, float clamp,
                       void* stream, float* y_amax);
int64_t pasta_mod_bias_act_bwd_workspace(int N, int C, int64_t HW);
int pasta_mod_bias_act_bwd(const void* dy, const void* y, const void* u, const float* d, const float* noise, void* du,
                           float* partial, int dtype, int N, int C, int64_t HW, int noise_per_sample, int act, float alpha, float gain,
                           float clamp, void* stream, float* du_amax);

/* ------------------------------------------------------------------------- *
 * SPADE normalisation (training/networks.py:4371-4379):
 *   out = InstanceNorm(x) * (1 + gamma) + beta, eps 1e-5, biased variance.
 * stats: [N*C, 2] (mean, rstd) written by the forward, read by the backward.
 * Optional fused activation (act = 2): out = min(relu(out) * gain, clamp) -- what the Spade_Conv2dLayer that
 * consumes the block's output applies in front of its convolution (networks.py:4346-4352); act 0/1 = none,
 * clamp < 0 = none.  The backward then needs beta (to recompute the activation mask) and a dbeta buffer.
 * gamma and beta (and dgamma, dbeta) may be the two channel halves of ONE [N, 2C, H, W] tensor -- the output (gradient) of a
 * single convolution with the concatenated conv_gamma / conv_beta weights: C = channels of x, gb_stride / dgb_stride = the
 * distance in elements between consecutive samples of gamma (dgamma), 0 = C * HW (separate contiguous tensors).
 * ------------------------------------------------------------------------- */
int pasta_spade_norm(const void* x, const void* gamma, const void* beta,
                     void* out, float* stats, int dtype, int64_t planes, int64_t HW,
                     float eps, int act, float gain, float clamp, int C, int64_t gb_stride, void* stream, float* y_amax);
int pasta_spade_norm_bwd(const void* dout, const void* x, const void* gamma,
                         const float* stats, void* dx, void* dgamma,
                         void* dbeta, int dtype, int64_t planes, int64_t HW,
                         const void* beta, int act, float gain, float clamp, int C, int64_t gb_stride, int64_t dgb_stride,
                         void* stream, float* dx_amax, float* dgb_amax,      /* dgb_amax: |max| over what is written to dgamma AND dbeta (they
                                                                               feed one convolution when they are halves of one tensor) */
                         const void* dx_add);                                /* ABI 18, optional ([planes, HW] like dx): added to dx on its way out -- the
                                                                               gradient another consumer of x returned (a block's second normalisation
                                                                               of the same tensor), instead of an addition pass over both */

/* ------------------------------------------------------------------------- *
 * Garment features of the SPADE stage (training/networks.py:5777-5800, get_spade_feat): fp32, NCHW.
 *   backward = 0:  out[n,c,i] = a[n,c,i] * (1 - hole[n,i]) + hole[n,i]  * inv_count[n] * sum_j a[n,c,j] * valid[n,j]
 *   backward = 1:  out[n,c,i] = a[n,c,i] * (1 - hole[n,i]) + valid[n,i] * inv_count[n] * sum_j a[n,c,j] * hole[n,j]   (a = d out)
 * valid, hole: [N, HW]; inv_count: [N]; a / out: sample strides in elements (0 = C * HW), so that the two garments' results
 * are written into (their gradients read from) the channel halves of ONE [N, 2C, H, W] tensor -- no torch.cat.
 * ------------------------------------------------------------------------- */
int pasta_masked_mean_fill(const float* a, const float* valid, const float* hole, const float* inv_count, float* out,
                           int N, int C, int64_t HW, int64_t a_sample_stride, int64_t out_sample_stride, int backward,
                           void* stream, float* y_amax);

/* ------------------------------------------------------------------------- *
 * ADA augmentation (training/augment.py:121-431).
 * pasta_ada_matrices: sample s turns its draws u[s, :] ~ U(0,1), z[s, :] ~ N(0,1) into the inverse geometric
 *   transform g_inv[s] (3x3 row-major, augment.py:186-263) and the colour transform c[s] (4x4, :306-350); margins[4]
 *   = the reflect-padding widths (x0, y0, x1, y1) over the whole batch (:272-282).  Column order of u / z = the order
 *   augment.py draws them (the enum in csrc/augment.hip; training/augment.py DRAWS_U / DRAWS_Z).  p: device scalar,
 *   the overall probability multiplier (AugmentPipe.p).  debug_percentile < 0 = off (:179-180).
 * pasta_ada_theta: theta[s] = (a @ g_inv[s] @ b)[:2, :]; a, b are HOST arrays of 9 floats (:285-296).
 * pasta_color_affine: mode 0: out[n, :, p] = c[n][:3, :3] @ x[n, :, p] + c[n][:3, 3] for [N,3,HW] images (:356-360);
 *   mode 1: the adjoint (c[n][:3, :3]^T, no offset); mode 2: the linear part alone (for second derivatives).
 * ------------------------------------------------------------------------- */
typedef struct pasta_ada_config {
    float xflip, rotate90, xint, xint_max;
    float scale, rotate, aniso, xfrac, scale_std, rotate_max, aniso_std, xfrac_std;
    float brightness, contrast, lumaflip, hue, saturation, brightness_std, contrast_std, hue_max, saturation_std;
} pasta_ada_config;
int pasta_ada_matrices(const float* u, const float* z, int64_t n, int u_cols, int z_cols, const float* p,
                       const pasta_ada_config* cfg, int width, int height, int channels, int hz_pad,
                       float debug_percentile, float* g_inv, float* c, int32_t* margins, void* stream);
int pasta_ada_theta(const float* g_inv, int64_t n, const float* a, const float* b, float* theta, void* stream);
int pasta_color_affine(const float* x, const float* c, float* out, int64_t n, int64_t hw, int mode, void* stream);
/* grid[n, y, x, :] = theta[n] @ ((2x + 1) / W - 1, (2y + 1) / H - 1, 1): F.affine_grid(theta, [n, C, H, W], align_corners=False) (:297) */
int pasta_ada_grid(const float* theta, int64_t n, int H, int W, float* grid, void* stream);
/* y = grid_sample(x, affine_grid(theta, [n, C, OH, OW], align_corners=False), bilinear, zeros, align_corners=False)
 * (augment.py:297-298) without the grid tensor; x: [n, C, IH, IW], theta: [n, 2, 3].  The adjoint is the gradient with
 * respect to x, computed as a gather (no atomics: bitwise reproducible). */
int pasta_affine_sample(const float* x, const float* theta, float* y, int64_t n, int C, int IH, int IW, int OH, int OW, void* stream);
int pasta_affine_sample_adjoint(const float* dy, const float* theta, float* dx, int64_t n, int C, int IH, int IW, int OH, int OW,
                                void* stream);

/* nan_to_num(t, nan, posinf, neginf) in place over n float tensors in one launch per 96 tensors
 * (training_loop_wo_flow_fullbody.py:513-515; misc.py:45).  ptrs / numels: HOST arrays of device pointers / element
 * counts (< 2^31 each); empty tensors are skipped. */
int pasta_nan_to_num_multi(float* const* ptrs, const int64_t* numels, int n, float nan, float posinf, float neginf,
                           void* stream);

/* ------------------------------------------------------------------------- *
 * Body-part patch pipeline (training/dataset.py:838-927 `normalize`,
 * which the reference runs on the host through cv2.warpPerspective, ~28 warps per sample).
 * uint8 HWC images; bilinear interpolation in OpenCV's fixed point (1/32-pixel source
 * coordinates, 2^15 weights).  OpenCV is unavailable where this was built: the arithmetic is
 * held bit for bit to oracle/ref_patches.py's restatement of the published algorithm, parity
 * with cv2 itself is UNPINNED.
 * ------------------------------------------------------------------------- */
/* dst[b] = cv2.warpPerspective(src[src_index ? src_index[b] : b], M_b, (dw, dh), INTER_LINEAR, border) for b < B.
 * src: [*, sh, sw, C] uint8, dst: [B, dh, dw, C]; minv: [B][9] doubles = the INVERTED matrices (destination -> source,
 * what cv2 forms first); valid (optional, [B]): 0 writes zeros; border: 0 = BORDER_CONSTANT (0), 1 = BORDER_REPLICATE. */
int pasta_warp_perspective_u8(const uint8_t* src, const int32_t* src_index, const double* minv, const uint8_t* valid,
                              uint8_t* dst, int B, int sh, int sw, int dh, int dw, int C, int border, void* stream);

/* dataset.py:884-888 / 894-898 for N samples x P parts in one pass: out[n] starts black; for k = 0..P-1 with valid[n][k]:
 * where channel 0 of warpPerspective(masks[n][k], BORDER_CONSTANT) is 255, the pixel becomes warpPerspective(patches[n][k]).
 * patches, masks: [N, P, ph, pw, 3] uint8; minv: [N][P][9] doubles (destination -> patch); out: [N, H, W, 3];
 * part_mask (optional): [N, P, H, W] receives each part's 0 / 1 mask (the reference keeps those of the four arm parts). */
int pasta_patch_composite_u8(const uint8_t* patches, const uint8_t* masks, const double* minv, const uint8_t* valid,
                             uint8_t* out, uint8_t* part_mask, int N, int P, int ph, int pw, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PASTA_HIP_H */
