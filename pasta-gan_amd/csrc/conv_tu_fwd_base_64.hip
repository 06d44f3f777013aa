// Translation unit of the convolution family (conv_launch.h): conv_fwd_bf16x6_kernel on the 64 x 256 tile, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_fwd_bf16x6.h"

namespace pasta {
void tu_fwd_base_64(const ConvFwdParams& q, dim3 grid, hipStream_t s) { launch_fwd_base_any<64, 256>(q, grid, s); }
}  // namespace pasta
