// Launch entry points of the convolution family's translation units.
//
// The planner and the C ABI live in conv_igemm.hip; every kernel family is compiled in a translation unit of its own (conv_tu_*.hip:
// explicit instances behind plain host functions, built in parallel by torch_utils/custom_ops.py) and reached through the functions
// declared here.  The kernel headers hold templates and host predicates only, so including them instantiates nothing; the few
// non-template kernels are fenced by PASTA_TU_PACK / PASTA_TU_WGRAD_F32 and defined by exactly one unit.
#pragma once
#include "conv_common.h"

namespace pasta {

// conv_tu_pack_f32.hip: weight packing (all arithmetics) and the fp32-MFMA forward-type kernels
void tu_pack_weights_f32(const float* w, float* wp, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                         float wscale, const float* mod_s, const float* mod_d, hipStream_t s);
void tu_pack_weights_bf16(const float* w, void* wp, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                          float wscale, int f16, const float* mod_s, const float* mod_d, hipStream_t s);
void tu_pack_weights_f16x3(const float* w, void* wp, float* rowinv, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                           int flip, float wscale, const float* mod_s, const float* mod_d, int pack_xcd_rows, hipStream_t s);
void tu_pack_weights_f16x3_pair(const float* w, const PackJob& a, const PackJob& b, hipStream_t s);      // one launch for two orientations (a.G == b.G)
void tu_fwd_f32(FwdTile t, const ConvFwdParams& p, hipStream_t s);

// conv_tu_fwd_base_{128,64}.hip: conv_fwd_bf16x6_kernel (q: o_tiles set, grid computed by the caller)
void tu_fwd_base_128(const ConvFwdParams& q, dim3 grid, hipStream_t s);
void tu_fwd_base_64(const ConvFwdParams& q, dim3 grid, hipStream_t s);
// conv_tu_fwd_rows_{128,64}.hip: conv_fwd_rows_bf16x6_kernel; false = the lattice is not made of whole row segments (the base kernel takes it)
bool tu_fwd_rows_128(const ConvFwdParams& q, dim3 grid, hipStream_t s);
bool tu_fwd_rows_64(const ConvFwdParams& q, dim3 grid, hipStream_t s);
void tu_fwd_pair_128(const ConvFwdParams& p, hipStream_t s);      // parity-pair mode (p.bf16x6: 3 or NP_F16X3)
void tu_fwd_pair_64(const ConvFwdParams& p, hipStream_t s);

// conv_tu_rows2d_*.hip: conv_fwd_rows2d_bf16x6_kernel, one unit per tile shape (q.rows_y0 set by the caller)
void tu_rows2d_wide(const ConvFwdParams& q, hipStream_t s);       // <128, 256, 8, ., IO_F32, ., 512>: the dominant kernel
void tu_rows2d_wide_glds(const ConvFwdParams& w8, dim3 grid8, hipStream_t s);     // conv_tu_rows2d_wide_glds.hip: the same with the weights by LDS-DMA
void tu_rows2d_wide_pieces(const ConvFwdParams& q, hipStream_t s);       // conv_tu_rows2d_wide_xp.hip: x as PASTA_LAYOUT_PIECES16
void tu_rows2d_128_r4(const ConvFwdParams& q, hipStream_t s);
void tu_rows2d_128_r2(const ConvFwdParams& q, hipStream_t s);
void tu_rows2d_64_r8(const ConvFwdParams& q, hipStream_t s);

// conv_tu_fwd_small.hip: the pointwise and the stride-2 kernels
void tu_conv1x1(const ConvFwdParams& p, hipStream_t s);
void tu_conv3x3s2(const ConvFwdParams& p, hipStream_t s);

// conv_tu_fwd_t2.hip: 3x3 stride-2 conv_transpose2d in one pass over the input lattice (p.iscale: the modulated layers)
void tu_conv_t2(const ConvFwdParams& p, hipStream_t s);

// conv_tu_fwd_fewch.hip: pointwise convolutions with <= 16 channels on one side (kind 1: few input channels, 2: few output channels)
struct FewChParams;
void tu_conv1x1_fewch(int kind, const FewChParams& p, hipStream_t s);

// conv_tu_wgrad_f32.hip: fp32-MFMA weight gradients, few-channel kernels, slab reductions
int  tu_wgrad_f32(int TR, int TS, int WA, int pipe, int kp, const WgradParams& p, int64_t blocks, size_t lds_bytes, hipStream_t s);
void tu_wgrad_reduce(const float* slab, float* dw, int ksplit, int G, int Ag, int Bg, int Ag_pad, int Bg_pad, int kh, int kw, int flip, float wscale,
                     hipStream_t s);
void tu_wgrad_reduce_modulated(bool mod_a, dim3 grid, const float* slab, const float* sty, const float* w, float* dw, float* dsp, int ksplit, int N,
                               int Ag, int Bg, int Ap, int Bp, int kh, int kw, int flip, float wscale, int wg_rows, hipStream_t s);
void tu_sum_blocks(const float* blocks, float* out, int nblocks, int n, hipStream_t s);
struct WgradSmallParams;
void tu_wgrad_smallcin(const WgradSmallParams& q, int blocks, size_t lds_bytes, hipStream_t s);
void tu_wgrad_smallcin_reduce(const float* slab, float* dw, int ksplit, int Ag, int bprime, int a_pad, int bpad, float wscale, hipStream_t s);
void tu_wgrad1x1_fewcin(int CI, int io, dim3 grid, const void* dy, const void* x, float* slab, int N, int Co, int HW, int64_t quads_per_slice, int a_pad,
                        int bpad, hipStream_t s);

// conv_tu_wgrad_{3x3,3x3s2,1x1}.hip: the split weight-gradient kernels (np: pieces per operand, NP_F16X3 included; p.io: storage type)
void tu_wgrad3x3(int np, const WgradParams& p, int64_t blocks, hipStream_t s);
void tu_wgrad3x3s2(int np, const WgradParams& p, int64_t blocks, hipStream_t s);
void tu_wgrad1x1(int np, int WA, const WgradParams& p, int64_t blocks, hipStream_t s);

}  // namespace pasta
