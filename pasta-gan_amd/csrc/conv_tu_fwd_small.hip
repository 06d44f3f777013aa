// Translation unit of the convolution family (conv_launch.h): the pointwise kernel (conv_fwd_1x1.h) and the 3x3 stride-2 kernel (conv_fwd_s2.h).
#include "conv_launch.h"
#include "conv_fwd_1x1.h"
#include "conv_fwd_s2.h"

namespace pasta {

void tu_conv1x1(const ConvFwdParams& p, hipStream_t s) {
    const int64_t pixels = (int64_t)p.N * p.H * p.W;
    if (p.Og <= 64) hipLaunchKernelGGL((conv1x1_f16x3_kernel<64, 256>), dim3((unsigned)(pixels / 256), (unsigned)((p.Og + 63) / 64)), dim3(256), 0, s, p);
    else            hipLaunchKernelGGL((conv1x1_f16x3_kernel<128, 128>), dim3((unsigned)(pixels / 128), (unsigned)((p.Og + 127) / 128)), dim3(256), 0, s, p);
}

void tu_conv3x3s2(const ConvFwdParams& p, hipStream_t s) {
    const int64_t tiles = (int64_t)p.N * p.OH * p.OW / 128;
    const dim3 g64((unsigned)tiles, (unsigned)((p.Og + 63) / 64)), g128((unsigned)tiles, (unsigned)((p.Og + 127) / 128));
    if (p.x_pieces) {       // x as the producer wrote it (pieces.hip)
        if (p.Og <= 64) hipLaunchKernelGGL((conv3x3s2_f16x3_kernel<64, true>), g64, dim3(256), 0, s, p);
        else            hipLaunchKernelGGL((conv3x3s2_f16x3_kernel<128, true>), g128, dim3(256), 0, s, p);
        return;
    }
    if (p.Og <= 64) hipLaunchKernelGGL((conv3x3s2_f16x3_kernel<64>), g64, dim3(256), 0, s, p);
    else            hipLaunchKernelGGL((conv3x3s2_f16x3_kernel<128>), g128, dim3(256), 0, s, p);
}

}  // namespace pasta
