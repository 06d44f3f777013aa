// Translation unit of the convolution family (conv_launch.h): the eight-wave 128 x 256 2-D tile reading operand pieces (PASTA_LAYOUT_PIECES16).
#include "conv_launch.h"
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {
void tu_rows2d_wide_pieces(const ConvFwdParams& q, hipStream_t s) {
    ConvFwdParams w8 = q;
    w8.o_tiles = (q.Og + 127) / 128;
    const int64_t tiles = (int64_t)q.N * (q.cls[0].P / 8) * (q.cls[0].Q / 32);
    const dim3 grid8((unsigned)tiles, w8.o_tiles * w8.ksplit, q.G);
    launch_fwd_rows2d_np<128, 256, 8, NP_F16X3, IO_F32, false, 512, true>(w8, grid8, s);
}
}  // namespace pasta
