// Translation unit of the convolution family (conv_launch.h): the one-pass 3x3 stride-2 conv_transpose2d kernel.
#include "conv_launch.h"
#include "conv_fwd_t2.h"

namespace pasta {
void tu_conv_t2(const ConvFwdParams& p, hipStream_t s) { launch_conv_t2(p, s); }
}  // namespace pasta
