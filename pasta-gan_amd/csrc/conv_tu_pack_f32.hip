// Translation unit of the convolution family (conv_launch.h): the weight-packing kernels of every arithmetic and the fp32-MFMA
// forward-type kernels.
#define PASTA_TU_PACK 1
#include "conv_launch.h"
#include "conv_fwd_f32.h"
#include "conv_fwd_bf16x6.h"

namespace pasta {

static unsigned pack_blocks(int64_t total) {
    const int64_t b = ceil_div64(total, 256);
    return (unsigned)(b > 4096 ? 4096 : b);
}

void tu_pack_weights_f32(const float* w, float* wp, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                         float wscale, const float* mod_s, const float* mod_d, hipStream_t s) {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(pack_blocks((int64_t)G * kh * kw * Ig_pad * Og_pad)), dim3(256), 0, s, w, wp, G, Ig, Og, Ig_pad, Og_pad,
                       kh, kw, transposed, flip, wscale, mod_s, mod_d);
}

void tu_pack_weights_bf16(const float* w, void* wp, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                          float wscale, int f16, const float* mod_s, const float* mod_d, hipStream_t s) {
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(pack_blocks((int64_t)G * kh * kw * Ig_pad * Og_pad)), dim3(256), 0, s, w, (__bf16*)wp, G, Ig, Og,
                       Ig_pad, Og_pad, kh, kw, transposed, flip, wscale, f16, mod_s, mod_d);
}

// one workgroup per packed row: two fp16 pieces and h'', one scale per output row found on the way
void tu_pack_weights_f16x3(const float* w, void* wp, float* rowinv, int G, int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                           int flip, float wscale, const float* mod_s, const float* mod_d, int pack_xcd_rows, hipStream_t s) {
    hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3((unsigned)Og_pad, (unsigned)G), dim3(256), 0, s, w, (__bf16*)wp, rowinv, Ig, Og, Ig_pad, Og_pad,
                       kh, kw, transposed, flip, wscale, mod_s, mod_d, pack_xcd_rows);
}

void tu_pack_weights_f16x3_pair(const float* w, const PackJob& a, const PackJob& b, hipStream_t s) {
    hipLaunchKernelGGL(pack_weights_f16x3_pair_kernel, dim3((unsigned)(a.Og_pad + b.Og_pad), (unsigned)a.G), dim3(256), 0, s, w, a, b);
}

void tu_fwd_f32(FwdTile t, const ConvFwdParams& p, hipStream_t s) {
    constexpr int KC = 8;
    switch (t) {
        case T128x128: launch_fwd<128, 128, 2, 2, KC, 4>(p, s); break;        // 4 waves/SIMD: 99-112 TFLOP/s vs 95-104 at 3
        case T64x256:
            if (p.Ig_pad == 4) launch_fwd<64, 256, 2, 2, 4>(p, s);             // RGB stems: 4-channel K chunks
            else launch_fwd<64, 256, 2, 2, KC, 4>(p, s);
            break;
        case T32x256:  launch_fwd<32, 256, 1, 2, KC>(p, s); break;
        case T64x64:   launch_fwd<64, 64, 1, 1, KC>(p, s); break;
    }
}

}  // namespace pasta
