// Translation unit of the convolution family (conv_launch.h): conv_fwd_rows_bf16x6_kernel on the 64 x 256 tile (row reuse; the parity-pair mode).
#include "conv_launch.h"
#include "conv_fwd_bf16x6.h"

namespace pasta {
bool tu_fwd_rows_64(const ConvFwdParams& q, dim3 grid, hipStream_t s) { return launch_fwd_rows_any<64, 256>(q, grid, s); }
void tu_fwd_pair_64(const ConvFwdParams& p, hipStream_t s) {
    if (p.bf16x6 == NP_F16X3) launch_fwd_pair<64, 256, NP_F16X3>(p, s); else launch_fwd_pair<64, 256>(p, s);
}
}  // namespace pasta
