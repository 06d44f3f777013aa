// Translation unit of the convolution family (conv_launch.h): the split weight-gradient kernel for 3x3 convolutions, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_wgrad_bf16x6.h"

namespace pasta {

// LAUNCH_(NP, IO) for the runtime piece count and storage type
#define PASTA_NP(LAUNCH_)                                                                                                  \
    do { if (p.io == IO_BF16) { LAUNCH_(1, IO_BF16); } else if (p.io == IO_F16) { LAUNCH_(1, IO_F16); }                    \
         else if (np == 1) { LAUNCH_(1, IO_F32); } else if (np == 2) { LAUNCH_(2, IO_F32); } else if (np == NP_F16X3) { LAUNCH_(NP_F16X3, IO_F32); } else { LAUNCH_(3, IO_F32); } } while (0)

void tu_wgrad3x3(int np, const WgradParams& p, int64_t blocks, hipStream_t s) {
    const int npw = np == NP_F16X3 ? 2 : np;                           // pieces per operand in LDS
    const size_t lds = (size_t)(npw * 64 * 40 + npw * 64 * 3 * 40) * 2;
#define PASTA_L(NP_, IO_) hipLaunchKernelGGL((conv_wgrad3x3_bf16x6_kernel<NP_, IO_>), dim3((unsigned)blocks), dim3(256), lds, s, p)
    PASTA_NP(PASTA_L);
#undef PASTA_L
}
#undef PASTA_NP
}  // namespace pasta
