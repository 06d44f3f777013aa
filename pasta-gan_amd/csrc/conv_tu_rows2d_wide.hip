// Translation unit of the convolution family (conv_launch.h): the eight-wave 128 x 256 2-D tile -- the dominant kernel of the training step
// (every 3x3 stride-1 convolution and input gradient over planes of a multiple of 8 rows x 32 columns, >= 128 output channels).
#include "conv_launch.h"
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {
void tu_rows2d_wide(const ConvFwdParams& q, hipStream_t s) {
    ConvFwdParams w8 = q;
    w8.o_tiles = (q.Og + 127) / 128;
    const int64_t tiles = (int64_t)q.N * (q.cls[0].P / 8) * (q.cls[0].Q / 32);
    const dim3 grid8((unsigned)tiles, w8.o_tiles * w8.ksplit, q.G);
    if (q.bf16x6 == NP_F16X3 && q.iscale) launch_fwd_rows2d_np<128, 256, 8, NP_F16X3, IO_F32, true, 512>(w8, grid8, s);      // the training step's modulated layers (round 4)
    else if (q.bf16x6 == NP_F16X3) {
        // weights by LDS-DMA: +1 % on the micro-benchmark's shapes, -2.3 % on the training step's mix (profiles/r5_ab_rows2d_glds.txt): opt-in
        static const bool glds = getenv("PASTA_ROWS2D_GLDS") && getenv("PASTA_ROWS2D_GLDS")[0] == '1';
        if (glds) tu_rows2d_wide_glds(w8, grid8, s);
        else launch_fwd_rows2d_np<128, 256, 8, NP_F16X3, IO_F32, false, 512>(w8, grid8, s);
    }
    else launch_fwd_rows2d_np<128, 256, 8, 3, IO_F32, false, 512>(w8, grid8, s);
}
}  // namespace pasta
