// Translation unit of the convolution family (conv_launch.h): conv_fwd_rows2d_bf16x6_kernel, 64 x 256 tile of 8 rows, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {
void tu_rows2d_64_r8(const ConvFwdParams& q, hipStream_t s) { launch_fwd_rows2d<64, 256, 8>(q, s); }
}  // namespace pasta
