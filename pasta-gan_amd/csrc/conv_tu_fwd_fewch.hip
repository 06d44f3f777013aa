// Translation unit of the convolution family (conv_launch.h): pointwise convolutions with very few channels on one side (conv_fwd_fewch.h).
#include "conv_launch.h"
#include "conv_fwd_fewch.h"

namespace pasta {

void tu_conv1x1_fewch(int kind, const FewChParams& p, hipStream_t s) {
    const int hwq = p.HW >> 2;
    if (kind == 1) {        // few input channels: all output channels per thread
        const int64_t quads = (int64_t)p.N * hwq;
        const dim3 grid((unsigned)((quads + 255) / 256));
        const size_t lds = (size_t)p.Cout * ((p.Cin + 3) & ~3) * sizeof(float);
#define PASTA_FEW(CI_) case CI_: if (p.io == IO_BF16) hipLaunchKernelGGL((conv1x1_fewcin_kernel<CI_, IO_BF16>), grid, dim3(256), lds, s, p); \
                                else if (p.io == IO_F16) hipLaunchKernelGGL((conv1x1_fewcin_kernel<CI_, IO_F16>), grid, dim3(256), lds, s, p); \
                                else hipLaunchKernelGGL((conv1x1_fewcin_kernel<CI_, IO_F32>), grid, dim3(256), lds, s, p); break;
        switch (p.Cin) { PASTA_FEW(1) PASTA_FEW(2) PASTA_FEW(3) PASTA_FEW(4) PASTA_FEW(5) PASTA_FEW(6) PASTA_FEW(7) PASTA_FEW(8) PASTA_FEW(9) PASTA_FEW(10) PASTA_FEW(11)
                         PASTA_FEW(12) PASTA_FEW(13) PASTA_FEW(14) PASTA_FEW(15) PASTA_FEW(16) }
#undef PASTA_FEW
        return;
    }
    const dim3 grid((unsigned)((hwq + 255) / 256), (unsigned)p.N);
    const size_t lds = (size_t)p.Cin * ((p.Cout + 3) & ~3) * sizeof(float);
#define PASTA_FEW(CO_) case CO_: if (p.io == IO_BF16) hipLaunchKernelGGL((conv1x1_fewcout_kernel<CO_, IO_BF16>), grid, dim3(256), lds, s, p); \
                                else if (p.io == IO_F16) hipLaunchKernelGGL((conv1x1_fewcout_kernel<CO_, IO_F16>), grid, dim3(256), lds, s, p); \
                                else hipLaunchKernelGGL((conv1x1_fewcout_kernel<CO_, IO_F32>), grid, dim3(256), lds, s, p); break;
    switch (p.Cout) { PASTA_FEW(1) PASTA_FEW(2) PASTA_FEW(3) PASTA_FEW(4) PASTA_FEW(5) PASTA_FEW(6) PASTA_FEW(7) PASTA_FEW(8) PASTA_FEW(9) PASTA_FEW(10) PASTA_FEW(11)
                      PASTA_FEW(12) PASTA_FEW(13) PASTA_FEW(14) PASTA_FEW(15) PASTA_FEW(16) }
#undef PASTA_FEW
}

}  // namespace pasta
