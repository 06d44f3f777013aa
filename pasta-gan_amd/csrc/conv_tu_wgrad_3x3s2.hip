// Translation unit of the convolution family (conv_launch.h): the split weight-gradient kernel for 3x3s2 convolutions, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_wgrad_bf16x6.h"

namespace pasta {

// LAUNCH_(NP, IO) for the runtime piece count and storage type
#define PASTA_NP(LAUNCH_)                                                                                                  \
    do { if (p.io == IO_BF16) { LAUNCH_(1, IO_BF16); } else if (p.io == IO_F16) { LAUNCH_(1, IO_F16); }                    \
         else if (np == 1) { LAUNCH_(1, IO_F32); } else if (np == 2) { LAUNCH_(2, IO_F32); } else if (np == NP_F16X3) { LAUNCH_(NP_F16X3, IO_F32); } else { LAUNCH_(3, IO_F32); } } while (0)

void tu_wgrad3x3s2(int np, const WgradParams& p, int64_t blocks, hipStream_t s) {
    if (p.l_pieces) {       // L as the producer wrote it (pieces.hip): copies and transposed LDS reads, no split (the planner checked: F16X3, pad 0, one group)
        hipLaunchKernelGGL((conv_wgrad3x3s2_pieces_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        return;
    }
    const int npw = np == NP_F16X3 ? 2 : np;
    const size_t lds = (size_t)(npw * 64 * 16 + npw * 64 * 3 * 40) * 2;
#define PASTA_L(NP_, IO_)                                                                                                     \
    if (p.pad_w == 1) hipLaunchKernelGGL((conv_wgrad3x3s2_bf16x6_kernel<1, NP_, IO_>), dim3((unsigned)blocks), dim3(256), lds, s, p); \
    else              hipLaunchKernelGGL((conv_wgrad3x3s2_bf16x6_kernel<0, NP_, IO_>), dim3((unsigned)blocks), dim3(256), lds, s, p)
    PASTA_NP(PASTA_L);
#undef PASTA_L
}
#undef PASTA_NP
}  // namespace pasta
