// Translation unit of the convolution family (conv_launch.h): fp32-MFMA weight gradients (conv_wgrad_kernel), the few-channel kernels and
// every slab reduction.
#define PASTA_TU_WGRAD_F32 1
#include "conv_launch.h"
#include "conv_wgrad_f32.h"

namespace pasta {

template <int TR, int TS, int WA, int WB, int PIPE, int KP>
static int launch_wgrad1(const WgradParams& p, int64_t blocks, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024)
        PASTA_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_kernel<TR, TS, WA, WB, PIPE, KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((conv_wgrad_kernel<TR, TS, WA, WB, PIPE, KP>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, p);
    return 0;
}
template <int TR, int TS, int WA, int WB>
static int launch_wgrad(int pipe, int kp, const WgradParams& p, int64_t blocks, size_t lds_bytes, hipStream_t s) {
    if (kp == 16) return pipe ? launch_wgrad1<TR, TS, WA, WB, 1, 16>(p, blocks, lds_bytes, s) : launch_wgrad1<TR, TS, WA, WB, 0, 16>(p, blocks, lds_bytes, s);
    return pipe ? launch_wgrad1<TR, TS, WA, WB, 1, 32>(p, blocks, lds_bytes, s) : launch_wgrad1<TR, TS, WA, WB, 0, 32>(p, blocks, lds_bytes, s);
}

int tu_wgrad_f32(int TR, int TS, int WA, int pipe, int kp, const WgradParams& p, int64_t blocks, size_t lds_bytes, hipStream_t s) {
    if (TR == 3 && TS == 3) return launch_wgrad<3, 3, 1, 1>(pipe, kp, p, blocks, lds_bytes, s);
    if (TS == 7) return launch_wgrad<1, 7, 1, 1>(pipe, kp, p, blocks, lds_bytes, s);
    if (TS == 4) return launch_wgrad<1, 4, 1, 1>(pipe, kp, p, blocks, lds_bytes, s);
    if (WA == 2) return launch_wgrad<1, 1, 2, 2>(pipe, kp, p, blocks, lds_bytes, s);
    return launch_wgrad<1, 1, 1, 1>(pipe, kp, p, blocks, lds_bytes, s);
}

void tu_wgrad_reduce(const float* slab, float* dw, int ksplit, int G, int Ag, int Bg, int Ag_pad, int Bg_pad, int kh, int kw, int flip, float wscale,
                     hipStream_t s) {
    const int64_t total = (int64_t)G * kh * kw * Ag * Bg;
    int64_t rb = ceil_div64(total, 256);
    if (rb > 8192) rb = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, slab, dw, ksplit, G, Ag, Bg, Ag_pad, Bg_pad, kh, kw, flip, wscale);
}

void tu_wgrad_reduce_modulated(bool mod_a, dim3 grid, const float* slab, const float* sty, const float* w, float* dw, float* dsp, int ksplit, int N,
                               int Ag, int Bg, int Ap, int Bp, int kh, int kw, int flip, float wscale, int wg_rows, hipStream_t s) {
    if (mod_a) hipLaunchKernelGGL((wgrad_reduce_modulated_kernel<true>), grid, dim3(256), 0, s, slab, sty, w, dw, dsp, ksplit, N, Ag, Bg, Ap, Bp, kh, kw, flip, wscale, wg_rows);
    else       hipLaunchKernelGGL((wgrad_reduce_modulated_kernel<false>), grid, dim3(256), 0, s, slab, sty, w, dw, dsp, ksplit, N, Ag, Bg, Ap, Bp, kh, kw, flip, wscale, wg_rows);
}

void tu_sum_blocks(const float* blocks, float* out, int nblocks, int n, hipStream_t s) {
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blocks, out, nblocks, n);
}

void tu_wgrad_smallcin(const WgradSmallParams& q, int blocks, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(conv_wgrad_smallcin_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, s, q);
}

void tu_wgrad_smallcin_reduce(const float* slab, float* dw, int ksplit, int Ag, int bprime, int a_pad, int bpad, float wscale, hipStream_t s) {
    const int total = Ag * bprime;
    hipLaunchKernelGGL(wgrad_smallcin_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slab, dw, ksplit, Ag, bprime, a_pad, bpad, wscale);
}

void tu_wgrad1x1_fewcin(int CI, int io, dim3 grid, const void* dy, const void* x, float* slab, int N, int Co, int HW, int64_t quads_per_slice, int a_pad,
                        int bpad, hipStream_t s) {
#define PASTA_FEW(CI_) case CI_: if (io == IO_BF16) hipLaunchKernelGGL((wgrad1x1_fewcin_kernel<CI_, IO_BF16>), grid, dim3(256), 0, s, dy, x, slab, N, Co, HW, quads_per_slice, a_pad, bpad); \
                                else if (io == IO_F16) hipLaunchKernelGGL((wgrad1x1_fewcin_kernel<CI_, IO_F16>), grid, dim3(256), 0, s, dy, x, slab, N, Co, HW, quads_per_slice, a_pad, bpad); \
                                else hipLaunchKernelGGL((wgrad1x1_fewcin_kernel<CI_, IO_F32>), grid, dim3(256), 0, s, dy, x, slab, N, Co, HW, quads_per_slice, a_pad, bpad); break;
    switch (CI) { PASTA_FEW(1) PASTA_FEW(2) PASTA_FEW(3) PASTA_FEW(4) PASTA_FEW(5) PASTA_FEW(6) PASTA_FEW(7) PASTA_FEW(8) }
#undef PASTA_FEW
}

}  // namespace pasta
