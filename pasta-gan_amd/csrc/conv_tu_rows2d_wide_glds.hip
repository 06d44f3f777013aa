// Translation unit of the convolution family (conv_launch.h): the eight-wave 128 x 256 2-D tile with its weights by LDS-DMA (GA; round 5).
#include "conv_launch.h"
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {
void tu_rows2d_wide_glds(const ConvFwdParams& w8, dim3 grid8, hipStream_t s) {       // w8.o_tiles and the grid: tu_rows2d_wide's
    launch_fwd_rows2d_np<128, 256, 8, NP_F16X3, IO_F32, false, 512, false, true>(w8, grid8, s);
}
}  // namespace pasta
