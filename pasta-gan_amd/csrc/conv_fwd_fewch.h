// Pointwise (1x1, stride 1) convolutions with very few channels on ONE side (round 5): the RGB / pose stems and their input gradients
// (3 .. 16 input channels into 64 .. 512 outputs) and the ToRGB / parsing heads and fromrgb's input gradient (64 .. 512 input channels
// into 3 .. 16 outputs).  Instantiated by conv_tu_fwd_fewch.hip (conv_launch.h).
//
// These launches are one pass over the LARGE tensor -- fromrgb over 48 stacked images writes 805 MB for 2.4 GFLOP -- and ran on the fp32
// MFMA tiles (conv_fwd_kernel<64,256,2,2,4> / <32,256,1,2,8>) at 1.7 - 1.9 TB/s, 0.22 of the HBM rate: 3 of 4 (or 3 of 32) matrix rows
// filled, a weight-packing launch in front, the activations gathered pixel by pixel through LDS (profiles/r5_byshape_start.txt: the class
// "1x1 few-channel", 3.2 ms per training step at 3.3 TFLOP/s).  Here they are what they are, streaming kernels of plain fp32 FMAs on the RAW
// weights (no packing launch, no operand scale: nothing to scan):
//   * a thread owns four consecutive pixels (16-byte accesses, a wave = 1 KB runs of every channel plane);
//   * few INPUT channels (conv1x1_fewcin_kernel<CI>): the CI input quads stay in registers, the thread walks ALL output channels -- the
//     weight row of a channel is one LDS address for the wave (a broadcast read) -- and stores one quad per channel: x is read once, y written once;
//   * few OUTPUT channels (conv1x1_fewcout_kernel<CO>): CO accumulator quads, the thread walks the input channels (eight loads in flight);
//     the weights of the launch sit in LDS, already multiplied by wscale and, for a modulated head (networks.py:74), by the sample's styles --
//     a workgroup lies inside one sample -- and are read as broadcasts.
// Epilogue as everywhere: residual, bias, linear / relu / lrelu, gain, clamp, the output's partial maxima.  conv2d and conv_transpose2d
// coincide for 1x1 / stride 1 up to the weight tensor's index order ([O][I] or [I][O]).  The sums are fp32 FMA chains in channel order.
#pragma once
#include "conv_common.h"

namespace pasta {

struct FewChParams {
    const void* x; const float* w; void* y;      // x, y, res: elements of the storage type IO (fp32, or 16-bit storage: BASELINE config 5)
    const float* iscale;            // [N][Cin] or null (fewcout only)
    const float* bias; const void* res;
    float* y_amax;
    int io;                         // IO_F32 / IO_F16 / IO_BF16
    int N, Cin, Cout, HW;
    int w_io;                       // the weight tensor is [Cin][Cout] (conv_transpose2d), else [Cout][Cin]
    float wscale;
    int act; float alpha, gain, clamp;
};

// four consecutive elements at element index `idx` of `base` (a multiple of four: naturally aligned), as fp32 / from fp32
template <int IO> __device__ __forceinline__ float4 fewch_ld4(const void* base, int64_t idx) { return io_ld4<IO>((const char*)base + idx * io_size<IO>::value); }
template <int IO> __device__ __forceinline__ void fewch_st4(void* base, int64_t idx, float4 v) {
    if constexpr (IO == IO_F32) *(float4*)((float*)base + idx) = v;
    else {
        uint2 q;
        q.x = io_pack2<IO>(v.x, v.y); q.y = io_pack2<IO>(v.z, v.w);
        *(uint2*)((char*)base + idx * 2) = q;
    }
}

__device__ __forceinline__ float4 fewch_epilogue(float4 v, float b, const FewChParams& p) {
    if (p.act) {
        v.x = conv_epilogue(v.x, b, p.act, p.alpha, p.gain, p.clamp); v.y = conv_epilogue(v.y, b, p.act, p.alpha, p.gain, p.clamp);
        v.z = conv_epilogue(v.z, b, p.act, p.alpha, p.gain, p.clamp); v.w = conv_epilogue(v.w, b, p.act, p.alpha, p.gain, p.clamp);
    }
    return v;
}

// grid.x = N * HW / 4 / 256 quads of pixels (HW % 4 == 0; a workgroup may straddle samples: the sample is per thread).  The weights of the
// launch (Cout x CI values times wscale, <= 32 KB) sit in LDS as [Cout][CI] and are read as broadcasts.
template <int CI, int IO>
__global__ __launch_bounds__(256) void conv1x1_fewcin_kernel(FewChParams p) {
    constexpr int CIP = (CI + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float fewch_w[];      // [Cout][CIP]
    for (int e = threadIdx.x; e < p.Cout * CIP; e += 256) {
        const int o = e / CIP, i = e - o * CIP;
        fewch_w[e] = i < CI ? (p.w_io ? p.w[(int64_t)i * p.Cout + o] : p.w[(int64_t)o * CI + i]) * p.wscale : 0.f;
    }
    __syncthreads();
    const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int hwq = p.HW >> 2;
    const int64_t total = (int64_t)p.N * hwq;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    if (quad < total) {
        const int n = (int)(quad / hwq);
        const int off = (int)(quad - (int64_t)n * hwq) * 4;
        float4 xv[CI];
#pragma unroll
        for (int i = 0; i < CI; i++) xv[i] = fewch_ld4<IO>(p.x, ((int64_t)n * CI + i) * p.HW + off);
        const int64_t ybase = (int64_t)n * p.Cout * p.HW + off;
        for (int o = 0; o < p.Cout; o++) {              // wave-uniform trip count; the weight row: one LDS address for the wave
            float wv[CIP];
#pragma unroll
            for (int i4 = 0; i4 < CIP; i4 += 4) *(float4*)&wv[i4] = *(const float4*)&fewch_w[o * CIP + i4];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < CI; i++) {
                a.x = fmaf(wv[i], xv[i].x, a.x); a.y = fmaf(wv[i], xv[i].y, a.y); a.z = fmaf(wv[i], xv[i].z, a.z); a.w = fmaf(wv[i], xv[i].w, a.w);
            }
            if (p.res) { const float4 r = fewch_ld4<IO>(p.res, ybase + (int64_t)o * p.HW); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
            a = fewch_epilogue(a, (p.act && p.bias) ? p.bias[o] : 0.f, p);
            fewch_st4<IO>(p.y, ybase + (int64_t)o * p.HW, a);
            if (p.y_amax) { amax_take(am, a.x); amax_take(am, a.y); amax_take(am, a.z); amax_take(am, a.w); }
        }
    }
    amax_commit(am, aslot);
}

// grid = (HW / 4 / 256 rounded up, N): a workgroup lies inside one sample (its styles ride in the LDS copy of the weights)
template <int CO, int IO>
__global__ __launch_bounds__(256) void conv1x1_fewcout_kernel(FewChParams p) {
    constexpr int COP = (CO + 3) & ~3;                  // outputs padded to whole 16-byte LDS reads
    extern __shared__ __attribute__((aligned(16))) float fewch_w[];      // [Cin][COP]: w[o][i] * wscale (* iscale[n][i])
    const int n = blockIdx.y;
    for (int e = threadIdx.x; e < p.Cin * COP; e += 256) {
        const int i = e / COP, o = e - i * COP;
        float v = 0.f;
        if (o < CO) {
            v = (p.w_io ? p.w[(int64_t)i * CO + o] : p.w[(int64_t)o * p.Cin + i]) * p.wscale;
            if (p.iscale) v *= p.iscale[(int64_t)n * p.Cin + i];
        }
        fewch_w[e] = v;
    }
    __syncthreads();
    const int hwq = p.HW >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    if (q < hwq) {
        const int off = q * 4;
        const int64_t xbase = (int64_t)n * p.Cin * p.HW + off;
        float4 acc[CO];
#pragma unroll
        for (int o = 0; o < CO; o++) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        int i = 0;
        for (; i + 8 <= p.Cin; i += 8) {                // eight channel quads in flight
            float4 xv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) xv[k] = fewch_ld4<IO>(p.x, xbase + (int64_t)(i + k) * p.HW);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float wv[COP];
#pragma unroll
                for (int o4 = 0; o4 < COP; o4 += 4) *(float4*)&wv[o4] = *(const float4*)&fewch_w[(i + k) * COP + o4];      // one address for the wave: a broadcast
#pragma unroll
                for (int o = 0; o < CO; o++) {
                    acc[o].x = fmaf(wv[o], xv[k].x, acc[o].x); acc[o].y = fmaf(wv[o], xv[k].y, acc[o].y);
                    acc[o].z = fmaf(wv[o], xv[k].z, acc[o].z); acc[o].w = fmaf(wv[o], xv[k].w, acc[o].w);
                }
            }
        }
        for (; i < p.Cin; i++) {
            const float4 xv = fewch_ld4<IO>(p.x, xbase + (int64_t)i * p.HW);
#pragma unroll
            for (int o = 0; o < CO; o++) {
                const float wv = fewch_w[i * COP + o];
                acc[o].x = fmaf(wv, xv.x, acc[o].x); acc[o].y = fmaf(wv, xv.y, acc[o].y); acc[o].z = fmaf(wv, xv.z, acc[o].z); acc[o].w = fmaf(wv, xv.w, acc[o].w);
            }
        }
        const int64_t ybase = (int64_t)n * CO * p.HW + off;
#pragma unroll
        for (int o = 0; o < CO; o++) {
            float4 a = acc[o];
            if (p.res) { const float4 r = fewch_ld4<IO>(p.res, ybase + (int64_t)o * p.HW); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
            a = fewch_epilogue(a, (p.act && p.bias) ? p.bias[o] : 0.f, p);
            fewch_st4<IO>(p.y, ybase + (int64_t)o * p.HW, a);
            if (p.y_amax) { amax_take(am, a.x); amax_take(am, a.y); amax_take(am, a.z); amax_take(am, a.w); }
        }
    }
    amax_commit(am, aslot);
}

// Which of the two takes a launch (0: neither): fp32 or 16-bit tensors (the stored element is converted on the way in and out: fp32 FMAs), 1x1, stride 1, no padding, one group, plain weights, no output scale / noise,
// planes of a multiple of four pixels, more than 8192 pixels (the K-sliced small-plane path keeps the rest); an input scale on the few-output side only.
static int conv1x1_fewch_kind(const pasta_conv_desc* d, bool has_iscale, bool has_oscale, bool has_noise, bool modulated) {
    static const bool enabled = !(getenv("PASTA_CONV_FEWCH") && getenv("PASTA_CONV_FEWCH")[0] == '0');       // A/B switch
    if (!enabled || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad_h || d->pad_w || d->groups != 1) return 0;
    if (has_oscale || has_noise || modulated || d->x2 || d->x_layout || d->OH != d->H || d->OW != d->W) return 0;
    const int64_t hw = (int64_t)d->H * d->W;
    if (hw % 4 || (int64_t)d->N * hw <= 8192) return 0;
    if (d->C_in <= 16 && !has_iscale && d->C_out >= 16 && d->C_out <= 512) return 1;
    if (d->C_out <= 16 && d->C_in >= 16 && (int64_t)d->C_in * ((d->C_out + 3) & ~3) * 4 <= 64 * 1024) return 2;      // the launch's weights fit the default LDS window
    return 0;
}

}  // namespace pasta
