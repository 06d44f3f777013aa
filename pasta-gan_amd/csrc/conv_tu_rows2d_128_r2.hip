// Translation unit of the convolution family (conv_launch.h): conv_fwd_rows2d_bf16x6_kernel, 128 x 128 tile of 2 rows, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {
void tu_rows2d_128_r2(const ConvFwdParams& q, hipStream_t s) { launch_fwd_rows2d<128, 128, 2>(q, s); }
}  // namespace pasta
