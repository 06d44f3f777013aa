// Translation unit of the convolution family (conv_launch.h): the split weight-gradient kernel for 1x1 convolutions, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_wgrad_bf16x6.h"

namespace pasta {

// LAUNCH_(NP, IO) for the runtime piece count and storage type
#define PASTA_NP(LAUNCH_)                                                                                                  \
    do { if (p.io == IO_BF16) { LAUNCH_(1, IO_BF16); } else if (p.io == IO_F16) { LAUNCH_(1, IO_F16); }                    \
         else if (np == 1) { LAUNCH_(1, IO_F32); } else if (np == 2) { LAUNCH_(2, IO_F32); } else if (np == NP_F16X3) { LAUNCH_(NP_F16X3, IO_F32); } else { LAUNCH_(3, IO_F32); } } while (0)

void tu_wgrad1x1(int np, int WA, const WgradParams& p, int64_t blocks, hipStream_t s) {
    const int npw = np == NP_F16X3 ? 2 : np;
    const size_t lds = (size_t)(npw * 64 * 40) * 2 * 2 * WA;        // S and L images of 64 WA (= 64 WB) channels
#define PASTA_L(NP_, IO_)                                                                                                     \
    if (WA == 2) hipLaunchKernelGGL((conv_wgrad1x1_bf16x6_kernel<2, 2, NP_, IO_>), dim3((unsigned)blocks), dim3(256), lds, s, p); \
    else         hipLaunchKernelGGL((conv_wgrad1x1_bf16x6_kernel<1, 1, NP_, IO_>), dim3((unsigned)blocks), dim3(256), lds, s, p)
    PASTA_NP(PASTA_L);
#undef PASTA_L
}
#undef PASTA_NP
}  // namespace pasta
