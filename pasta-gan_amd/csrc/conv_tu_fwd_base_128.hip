// Translation unit of the convolution family (conv_launch.h): conv_fwd_bf16x6_kernel on the 128 x 128 tile, every arithmetic and storage type.
#include "conv_launch.h"
#include "conv_fwd_bf16x6.h"

namespace pasta {
void tu_fwd_base_128(const ConvFwdParams& q, dim3 grid, hipStream_t s) { launch_fwd_base_any<128, 128>(q, grid, s); }
}  // namespace pasta
