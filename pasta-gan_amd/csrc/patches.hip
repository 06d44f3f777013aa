// Body-part patch pipeline on the GPU: the perspective warps of training/dataset.py:838-927
// (`normalize`: ten body parts of a person warped into (W/4) x (H/4) patches, warped back and composited where the warped
// garment mask is 255), which the reference runs per sample on the host through cv2.warpPerspective -- about 28 warps per
// sample, the reason its loader caps real-data throughput.
//
// Numerics: uint8 HWC images as the reference holds them; bilinear interpolation in OpenCV's fixed point (source
// coordinates in 1/32 pixel from a double-precision projective map, weights (32 - a)(32 - b) * 32 summing to 2^15,
// (sum + 2^14) >> 15), BORDER_REPLICATE / BORDER_CONSTANT(0).  OpenCV is not available where this was written and the
// reference ships no fixture of the pipeline: the arithmetic follows oracle/ref_patches.py's restatement of OpenCV's
// published algorithm bit for bit (tests/test_patches_gpu.py); parity with cv2 itself is UNPINNED.
#include "common.h"

#pragma clang fp contract(off)      // the coordinate arithmetic is compared bit for bit with separately rounded double operations

namespace pasta {

constexpr int PW_TAB = 32;          // sub-pixel positions per pixel
constexpr int PW_BLOCK_W = 64;      // the projective map is evaluated relative to the first column of 64-column blocks

// destination pixel (x, y) -> source coordinates in 1/32 pixel; m = the inverted 3 x 3 matrix (dst -> src), row-major doubles
__device__ __forceinline__ void pw_source(const double* __restrict__ m, int x, int y, int& X, int& Y) {
    const int xb = (x / PW_BLOCK_W) * PW_BLOCK_W;
    const double x1 = (double)(x - xb), xbd = (double)xb, yd = (double)y;
    const double x0 = m[0] * xbd + m[1] * yd + m[2];
    const double y0 = m[3] * xbd + m[4] * yd + m[5];
    const double w0 = m[6] * xbd + m[7] * yd + m[8];
    double w = w0 + m[6] * x1;
    w = w != 0.0 ? (double)PW_TAB / w : 0.0;
    double fx = (x0 + m[0] * x1) * w, fy = (y0 + m[3] * x1) * w;
    fx = fmin(fmax(fx, -2147483648.0), 2147483647.0);
    fy = fmin(fmax(fy, -2147483648.0), 2147483647.0);
    X = (int)rint(fx);              // round half to even
    Y = (int)rint(fy);
}

struct PwTaps { int x0, x1, y0, y1; int w00, w01, w10, w11; bool in00, in01, in10, in11; };

// tap positions and weights of one destination pixel; border = 1: coordinates clamped (replicate), 0: taps outside read 0
__device__ __forceinline__ PwTaps pw_taps(int X, int Y, int sw, int sh, int border) {
    PwTaps t;
    int sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;           // remap carries short coordinates
    sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
    const int ax = X & 31, ay = Y & 31;
    t.w00 = (32 - ay) * (32 - ax) * 32; t.w01 = (32 - ay) * ax * 32;
    t.w10 = ay * (32 - ax) * 32;        t.w11 = ay * ax * 32;
    const int xa = sx, xb = sx + 1, ya = sy, yb = sy + 1;
    const bool xin0 = (unsigned)xa < (unsigned)sw, xin1 = (unsigned)xb < (unsigned)sw;
    const bool yin0 = (unsigned)ya < (unsigned)sh, yin1 = (unsigned)yb < (unsigned)sh;
    t.in00 = border || (xin0 && yin0); t.in01 = border || (xin1 && yin0);
    t.in10 = border || (xin0 && yin1); t.in11 = border || (xin1 && yin1);
    t.x0 = xa < 0 ? 0 : xa >= sw ? sw - 1 : xa; t.x1 = xb < 0 ? 0 : xb >= sw ? sw - 1 : xb;
    t.y0 = ya < 0 ? 0 : ya >= sh ? sh - 1 : ya; t.y1 = yb < 0 ? 0 : yb >= sh ? sh - 1 : yb;
    return t;
}

__device__ __forceinline__ int pw_sample(const uint8_t* __restrict__ img, int sw, int C, int c, const PwTaps& t) {
    const int v00 = t.in00 ? img[((int64_t)t.y0 * sw + t.x0) * C + c] : 0, v01 = t.in01 ? img[((int64_t)t.y0 * sw + t.x1) * C + c] : 0;
    const int v10 = t.in10 ? img[((int64_t)t.y1 * sw + t.x0) * C + c] : 0, v11 = t.in11 ? img[((int64_t)t.y1 * sw + t.x1) * C + c] : 0;
    const int acc = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
    const int r = (acc + (1 << 14)) >> 15;
    return r < 0 ? 0 : r > 255 ? 255 : r;
}

// dst[b] = warpPerspective(src[src_index[b]], M_b): one thread per destination pixel, all channels.
__global__ __launch_bounds__(256) void warp_perspective_u8_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ src_index,
                                                                  const double* __restrict__ minv, const uint8_t* __restrict__ valid,
                                                                  uint8_t* __restrict__ dst, int sh, int sw, int dh, int dw, int C, int border) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= dh * dw) return;
    const int y = pix / dw, x = pix - y * dw;
    uint8_t* out = dst + ((int64_t)b * dh * dw + pix) * C;
    if (valid && !valid[b]) {
        for (int c = 0; c < C; c++) out[c] = 0;
        return;
    }
    const uint8_t* img = src + (int64_t)(src_index ? src_index[b] : b) * sh * sw * C;
    int X, Y;
    pw_source(minv + (int64_t)b * 9, x, y, X, Y);
    const PwTaps t = pw_taps(X, Y, sw, sh, border);
    for (int c = 0; c < C; c++) out[c] = (uint8_t)pw_sample(img, sw, C, c, t);
}

// The second half of `normalize` (dataset.py:884-888, 894-898) without its intermediate full-size images: for every output
// pixel the parts are visited in order; part k warps its patch and its mask patch back (BORDER_CONSTANT) and, where channel
// 0 of the warped mask is exactly 255, overwrites the pixel.  part_mask (optional, [N][P][H][W]) receives the 0 / 1 mask of
// every part (the reference keeps those of the four arm parts).
__global__ __launch_bounds__(256) void patch_composite_u8_kernel(const uint8_t* __restrict__ patches, const uint8_t* __restrict__ masks,
                                                                 const double* __restrict__ minv, const uint8_t* __restrict__ valid,
                                                                 uint8_t* __restrict__ out, uint8_t* __restrict__ part_mask,
                                                                 int P, int ph, int pw, int H, int W) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix - y * W;
    int r = 0, g = 0, bl = 0;
    for (int k = 0; k < P; k++) {
        const int64_t item = (int64_t)n * P + k;
        int hit = 0;
        if (valid[item]) {
            int X, Y;
            pw_source(minv + item * 9, x, y, X, Y);
            const PwTaps t = pw_taps(X, Y, pw, ph, 0);
            const uint8_t* mk = masks + item * ph * pw * 3;
            if (pw_sample(mk, pw, 3, 0, t) == 255) {
                const uint8_t* pt = patches + item * ph * pw * 3;
                r = pw_sample(pt, pw, 3, 0, t); g = pw_sample(pt, pw, 3, 1, t); bl = pw_sample(pt, pw, 3, 2, t);
                hit = 1;
            }
        }
        if (part_mask) part_mask[item * H * W + pix] = (uint8_t)hit;
    }
    uint8_t* o = out + ((int64_t)n * H * W + pix) * 3;
    o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)bl;
}

}  // namespace pasta

extern "C" int pasta_warp_perspective_u8(const uint8_t* src, const int32_t* src_index, const double* minv, const uint8_t* valid,
                                         uint8_t* dst, int B, int sh, int sw, int dh, int dw, int C, int border, void* stream) {
    using namespace pasta;
    PASTA_CHECK(src && minv && dst, "warp_perspective_u8: null pointer");
    PASTA_CHECK(B >= 1 && sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && C >= 1 && C <= 4, "warp_perspective_u8: bad shape");
    PASTA_CHECK(border == 0 || border == 1, "warp_perspective_u8: border %d (0 = constant 0, 1 = replicate)", border);
    PASTA_CHECK(B <= 65535, "warp_perspective_u8: at most 65535 images per call");
    dim3 grid((unsigned)((dh * dw + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(warp_perspective_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_index, minv, valid, dst, sh, sw, dh, dw, C, border);
    return launch_status("warp_perspective_u8");
}

extern "C" int pasta_patch_composite_u8(const uint8_t* patches, const uint8_t* masks, const double* minv, const uint8_t* valid,
                                        uint8_t* out, uint8_t* part_mask, int N, int P, int ph, int pw, int H, int W, void* stream) {
    using namespace pasta;
    PASTA_CHECK(patches && masks && minv && valid && out, "patch_composite_u8: null pointer");
    PASTA_CHECK(N >= 1 && N <= 65535 && P >= 1 && ph >= 1 && pw >= 1 && H >= 1 && W >= 1, "patch_composite_u8: bad shape");
    dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)N);
    hipLaunchKernelGGL(patch_composite_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, patches, masks, minv, valid, out, part_mask, P, ph, pw, H, W);
    return launch_status("patch_composite_u8");
}
