// ADA augmentation (training/augment.py:121-431 of the reference): the per-sample transform parameters and the
// colour transform as kernels.
//
//   ada_matrices   one thread per sample turns that sample's uniform / normal draws into the inverse geometric transform
//                  G_inv (3x3, augment.py:186-263) and the colour transform C (4x4, :306-350), and the workgroup reduces
//                  the reflect-padding margins over the batch (:272-282).  The reference builds these from ~150 tiny
//                  tensor ops per call; here it is one launch.
//   ada_theta      theta = (A @ G_inv @ B)[:2, :] for the host-known pre/post matrices of the pad / upsample / sampling
//                  grid steps (:285-296).
//   color_affine   out[n, :, p] = M[n] @ x[n, :, p] + t[n] for 3-channel images (:356-360); HBM-bound, one pass.
#include "common.h"

namespace pasta {

struct M3 { float m[9]; };
struct M4 { float m[16]; };

__device__ __forceinline__ M3 mul3(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ M4 mul4(const M4& a, const M4& b) {
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) s += a.m[i * 4 + k] * b.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
__device__ __forceinline__ M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
__device__ __forceinline__ M3 scale3(float sx, float sy) { return M3{{sx, 0, 0, 0, sy, 0, 0, 0, 1}}; }
__device__ __forceinline__ M3 shift3(float tx, float ty) { return M3{{1, 0, tx, 0, 1, ty, 0, 0, 1}}; }
// rotate2d(theta) of the reference: [[cos, sin(-theta)], [sin, cos]]
__device__ __forceinline__ M3 rot3(float th) { return M3{{cosf(th), sinf(-th), 0, sinf(th), cosf(th), 0, 0, 0, 1}}; }

// Column layout of the draws (augment.py draws them in this order; see training/augment.py of this package).
enum {
    U_XFLIP_I, U_XFLIP_ON, U_ROT90_I, U_ROT90_ON, U_XINT_X, U_XINT_Y, U_XINT_ON, U_SCALE_ON, U_ROT_PRE, U_ROT_PRE_ON,
    U_ANISO_ON, U_ROT_POST, U_ROT_POST_ON, U_XFRAC_ON, U_BRIGHT_ON, U_CONTRAST_ON, U_LUMA_I, U_LUMA_ON, U_HUE, U_HUE_ON,
    U_SAT_ON, U_COLS_MIN
};
enum { Z_SCALE, Z_ANISO, Z_XFRAC_X, Z_XFRAC_Y, Z_BRIGHT, Z_CONTRAST, Z_SAT, Z_COLS_MIN };

__global__ __launch_bounds__(256) void ada_matrices_kernel(const float* __restrict__ u, const float* __restrict__ z, int n, int u_cols,
                                                           int z_cols, const float* __restrict__ p_ptr, pasta_ada_config cfg, int width,
                                                           int height, int channels, int hz_pad, float dp, float* __restrict__ g_out,
                                                           float* __restrict__ c_out, int32_t* __restrict__ margins) {
    __shared__ float red[4][4];
    const float p = *p_ptr;
    const bool dbg = dp >= 0.f;
    const float PI = 3.14159265358979323846f;
    const float dpn = dbg ? erfinvf(dp * 2.f - 1.f) : 0.f;       // the normal-distributed parameters at that percentile
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // max over samples and corners of -x, -y, x, y
    const float cx = (width - 1) * 0.5f, cy = (height - 1) * 0.5f;

    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const float* us = u + (int64_t)s * u_cols;
        const float* zs = z + (int64_t)s * z_cols;
        M3 G = eye3();
        if (cfg.xflip > 0.f) {
            float i = floorf(us[U_XFLIP_I] * 2.f);
            i = us[U_XFLIP_ON] < cfg.xflip * p ? i : 0.f;
            if (dbg) i = floorf(dp * 2.f);
            G = mul3(G, scale3(1.f / (1.f - 2.f * i), 1.f));
        }
        if (cfg.rotate90 > 0.f) {
            float i = floorf(us[U_ROT90_I] * 4.f);
            i = us[U_ROT90_ON] < cfg.rotate90 * p ? i : 0.f;
            if (dbg) i = floorf(dp * 4.f);
            G = mul3(G, rot3(-(-PI / 2.f * i)));
        }
        if (cfg.xint > 0.f) {
            float tx = (us[U_XINT_X] * 2.f - 1.f) * cfg.xint_max, ty = (us[U_XINT_Y] * 2.f - 1.f) * cfg.xint_max;
            if (!(us[U_XINT_ON] < cfg.xint * p)) tx = ty = 0.f;
            if (dbg) tx = ty = (dp * 2.f - 1.f) * cfg.xint_max;
            G = mul3(G, shift3(-rintf(tx * width), -rintf(ty * height)));      // torch.round = half to even
        }
        if (cfg.scale > 0.f) {
            float sc = exp2f(zs[Z_SCALE] * cfg.scale_std);
            if (!(us[U_SCALE_ON] < cfg.scale * p)) sc = 1.f;
            if (dbg) sc = exp2f(dpn * cfg.scale_std);
            G = mul3(G, scale3(1.f / sc, 1.f / sc));
        }
        const float p_rot = 1.f - sqrtf(fminf(fmaxf(1.f - cfg.rotate * p, 0.f), 1.f));   // P(pre or post) = rotate * p
        if (cfg.rotate > 0.f) {
            float th = (us[U_ROT_PRE] * 2.f - 1.f) * PI * cfg.rotate_max;
            if (!(us[U_ROT_PRE_ON] < p_rot)) th = 0.f;
            if (dbg) th = (dp * 2.f - 1.f) * PI * cfg.rotate_max;
            G = mul3(G, rot3(th));
        }
        if (cfg.aniso > 0.f) {
            float sc = exp2f(zs[Z_ANISO] * cfg.aniso_std);
            if (!(us[U_ANISO_ON] < cfg.aniso * p)) sc = 1.f;
            if (dbg) sc = exp2f(dpn * cfg.aniso_std);
            G = mul3(G, scale3(1.f / sc, 1.f / (1.f / sc)));
        }
        if (cfg.rotate > 0.f) {
            float th = (us[U_ROT_POST] * 2.f - 1.f) * PI * cfg.rotate_max;
            if (!(us[U_ROT_POST_ON] < p_rot)) th = 0.f;
            if (dbg) th = 0.f;
            G = mul3(G, rot3(th));
        }
        if (cfg.xfrac > 0.f) {
            float tx = zs[Z_XFRAC_X] * cfg.xfrac_std, ty = zs[Z_XFRAC_Y] * cfg.xfrac_std;
            if (!(us[U_XFRAC_ON] < cfg.xfrac * p)) tx = ty = 0.f;
            if (dbg) tx = ty = dpn * cfg.xfrac_std;
            G = mul3(G, shift3(-(tx * width), -(ty * height)));
        }
#pragma unroll
        for (int k = 0; k < 9; k++) g_out[(int64_t)s * 9 + k] = G.m[k];
        // the image corners under G_inv (:272-276)
        const float px[4] = {-cx, cx, cx, -cx}, py[4] = {-cy, -cy, cy, cy};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float x = G.m[0] * px[k] + G.m[1] * py[k] + G.m[2], y = G.m[3] * px[k] + G.m[4] * py[k] + G.m[5];
            mx[0] = fmaxf(mx[0], -x); mx[1] = fmaxf(mx[1], -y); mx[2] = fmaxf(mx[2], x); mx[3] = fmaxf(mx[3], y);
        }

        // colour: C @ colour_in = colour_out (:306-350)
        M4 C = M4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
        const float v = 0.57735026918962576f;                    // luma axis (1, 1, 1, 0) / sqrt(3)
        const float vv[4] = {v, v, v, 0.f};
        if (cfg.brightness > 0.f) {
            float b = zs[Z_BRIGHT] * cfg.brightness_std;
            if (!(us[U_BRIGHT_ON] < cfg.brightness * p)) b = 0.f;
            if (dbg) b = dpn * cfg.brightness_std;
            C = mul4(M4{{1, 0, 0, b, 0, 1, 0, b, 0, 0, 1, b, 0, 0, 0, 1}}, C);
        }
        if (cfg.contrast > 0.f) {
            float c = exp2f(zs[Z_CONTRAST] * cfg.contrast_std);
            if (!(us[U_CONTRAST_ON] < cfg.contrast * p)) c = 1.f;
            if (dbg) c = exp2f(dpn * cfg.contrast_std);
            C = mul4(M4{{c, 0, 0, 0, 0, c, 0, 0, 0, 0, c, 0, 0, 0, 0, 1}}, C);
        }
        if (cfg.lumaflip > 0.f) {
            float i = floorf(us[U_LUMA_I] * 2.f);
            i = us[U_LUMA_ON] < cfg.lumaflip * p ? i : 0.f;
            if (dbg) i = floorf(dp * 2.f);
            M4 H;                                                  // Householder reflection about the luma axis
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) H.m[a * 4 + b] = (a == b ? 1.f : 0.f) - 2.f * (vv[a] * vv[b]) * i;
            C = mul4(H, C);
        }
        if (cfg.hue > 0.f && channels > 1) {
            float th = (us[U_HUE] * 2.f - 1.f) * PI * cfg.hue_max;
            if (!(us[U_HUE_ON] < cfg.hue * p)) th = 0.f;
            if (dbg) th = (dp * 2.f - 1.f) * PI * cfg.hue_max;
            const float sn = sinf(th), cs = cosf(th), cc = 1.f - cs;
            M4 R = M4{{v * v * cc + cs,     v * v * cc - v * sn, v * v * cc + v * sn, 0,
                       v * v * cc + v * sn, v * v * cc + cs,     v * v * cc - v * sn, 0,
                       v * v * cc - v * sn, v * v * cc + v * sn, v * v * cc + cs,     0,
                       0, 0, 0, 1}};
            C = mul4(R, C);
        }
        if (cfg.saturation > 0.f && channels > 1) {
            float sa = exp2f(zs[Z_SAT] * cfg.saturation_std);
            if (!(us[U_SAT_ON] < cfg.saturation * p)) sa = 1.f;
            if (dbg) sa = exp2f(dpn * cfg.saturation_std);
            M4 S;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) S.m[a * 4 + b] = vv[a] * vv[b] + ((a == b ? 1.f : 0.f) - vv[a] * vv[b]) * sa;
            C = mul4(S, C);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) c_out[(int64_t)s * 16 + k] = C.m[k];
    }

    // margins = ceil(clamp(max + (2 * hz_pad - c), 0, size - 1)) over the batch (:277-282)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float m = mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = m;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float m = fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]));
        m += hz_pad * 2 - ((k & 1) ? cy : cx);
        m = fminf(fmaxf(m, 0.f), (float)(((k & 1) ? height : width) - 1));
        margins[k] = (int32_t)ceilf(m);
    }
}

struct ThetaParams { float a[9], b[9]; };

__global__ void ada_theta_kernel(const float* __restrict__ g, int n, ThetaParams prm, float* __restrict__ theta) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    M3 G, A, B;
#pragma unroll
    for (int k = 0; k < 9; k++) { G.m[k] = g[(int64_t)s * 9 + k]; A.m[k] = prm.a[k]; B.m[k] = prm.b[k]; }
    const M3 T = mul3(mul3(A, G), B);
#pragma unroll
    for (int k = 0; k < 6; k++) theta[(int64_t)s * 6 + k] = T.m[k];
}

// Sampling grid of F.affine_grid(theta, [N, C, H, W], align_corners=False) without the batched GEMM it is made of:
// grid[n, y, x] = theta[n] @ ((2x + 1) / W - 1, (2y + 1) / H - 1, 1).  One thread per 2 horizontally adjacent points (16-byte stores).
__global__ __launch_bounds__(256) void ada_grid_kernel(const float* __restrict__ theta, int H, int W, float* __restrict__ grid) {
    const int n = blockIdx.z, y = blockIdx.y;
    const float* t = theta + (int64_t)n * 6;
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const float yn = (2.f * y + 1.f) / H - 1.f;
    float* row = grid + ((int64_t)n * H + y) * W * 2;
    for (int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x); x < W; x += 2 * gridDim.x * blockDim.x) {
        const float xa = (2.f * x + 1.f) / W - 1.f, xb = (2.f * (x + 1) + 1.f) / W - 1.f;
        const float4 v = make_float4(fmaf(t0, xa, fmaf(t1, yn, t2)), fmaf(t3, xa, fmaf(t4, yn, t5)),
                                     fmaf(t0, xb, fmaf(t1, yn, t2)), fmaf(t3, xb, fmaf(t4, yn, t5)));
        if (x + 1 < W && (W & 1) == 0) *(float4*)(row + 2 * x) = v;
        else { row[2 * x] = v.x; row[2 * x + 1] = v.y; if (x + 1 < W) { row[2 * x + 2] = v.z; row[2 * x + 3] = v.w; } }
    }
}

// Bilinear resampling under a per-sample affine map, zeros outside, align_corners = False: what
// grid_sample(x, affine_grid(theta, [N, C, OH, OW])) computes (augment.py:297-298), without the grid tensor.
//   output pixel (ox, oy) -> normalised (xn, yn) -> (gx, gy) = theta @ (xn, yn, 1) -> input position (u, v)
// The adjoint (gradient with respect to x) is a GATHER: input pixel (j, i) collects w * dy from the output pixels whose
// 2x2 footprint covers it -- the lattice points of a parallelogram around A^-1 (j - b0, i - b1).  No atomics, so the
// gradient is bitwise reproducible (ATen's grid_sampler_2d_backward scatters with atomicAdd: 2.8 ms for a 48 x 3 x 524^2
// output on this GPU, against 0.9 ms here).
struct AffinePix { float a00, a01, b0, a10, a11, b1; };     // u = a00 ox + a01 oy + b0, v = a10 ox + a11 oy + b1

__device__ __forceinline__ void affine_uv(const float* t, int ox, int oy, int IW, int IH, int OW, int OH, float& u, float& v) {
    const float xn = (2.f * ox + 1.f) / OW - 1.f, yn = (2.f * oy + 1.f) / OH - 1.f;
    const float gx = fmaf(t[0], xn, fmaf(t[1], yn, t[2])), gy = fmaf(t[3], xn, fmaf(t[4], yn, t[5]));
    u = ((gx + 1.f) * IW - 1.f) * 0.5f;
    v = ((gy + 1.f) * IH - 1.f) * 0.5f;
}

__global__ __launch_bounds__(256) void affine_sample_kernel(const float* __restrict__ x, const float* __restrict__ theta, float* __restrict__ y,
                                                            int C, int IH, int IW, int OH, int OW) {
    const int n = blockIdx.z, oy = blockIdx.y;
    const float* t = theta + (int64_t)n * 6;
    const float th[6] = {t[0], t[1], t[2], t[3], t[4], t[5]};
    for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < OW; ox += gridDim.x * blockDim.x) {
        float u, v;
        affine_uv(th, ox, oy, IW, IH, OW, OH, u, v);
        if (!(fabsf(u) < 1e8f && fabsf(v) < 1e8f)) u = v = -4.f;      // far outside (or NaN): every corner is out of range
        const float fu = floorf(u), fv = floorf(v);
        const int j0 = (int)fu, i0 = (int)fv;
        const float wx1 = u - fu, wy1 = v - fv, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool in_x0 = (unsigned)j0 < (unsigned)IW, in_x1 = (unsigned)(j0 + 1) < (unsigned)IW;
        const bool in_y0 = (unsigned)i0 < (unsigned)IH, in_y1 = (unsigned)(i0 + 1) < (unsigned)IH;
        for (int c = 0; c < C; c++) {
            const float* xp = x + ((int64_t)n * C + c) * IH * IW;
            float acc = 0.f;
            if (in_y0 && in_x0) acc = fmaf(xp[(int64_t)i0 * IW + j0], wx0 * wy0, acc);
            if (in_y0 && in_x1) acc = fmaf(xp[(int64_t)i0 * IW + j0 + 1], wx1 * wy0, acc);
            if (in_y1 && in_x0) acc = fmaf(xp[(int64_t)(i0 + 1) * IW + j0], wx0 * wy1, acc);
            if (in_y1 && in_x1) acc = fmaf(xp[(int64_t)(i0 + 1) * IW + j0 + 1], wx1 * wy1, acc);
            y[(((int64_t)n * C + c) * OH + oy) * OW + ox] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void affine_sample_adjoint_kernel(const float* __restrict__ dy, const float* __restrict__ theta,
                                                                    float* __restrict__ dx, int C, int IH, int IW, int OH, int OW) {
    const int n = blockIdx.z, i = blockIdx.y;
    const float* t = theta + (int64_t)n * 6;
    const float th[6] = {t[0], t[1], t[2], t[3], t[4], t[5]};
    // pixel-space form of the map and its inverse (only to bound the search; the weights use affine_uv itself)
    const float a00 = th[0] * IW / OW, a01 = th[1] * IW / OH, a10 = th[3] * IH / OW, a11 = th[4] * IH / OH;
    const float b0 = 0.5f * IW * (th[0] / OW - th[0] + th[1] / OH - th[1] + th[2] + 1.f) - 0.5f;
    const float b1 = 0.5f * IH * (th[3] / OW - th[3] + th[4] / OH - th[4] + th[5] + 1.f) - 0.5f;
    const float det = a00 * a11 - a01 * a10, inv = 1.f / det;
    const float i00 = a11 * inv, i01 = -a01 * inv, i10 = -a10 * inv, i11 = a00 * inv;
    // half extents of the image of the open square (-1, 1)^2, plus slack for the rounding of this bound
    const float hx = fabsf(i00) + fabsf(i01) + 1e-3f * (1.f + fabsf(i00) + fabsf(i01)), hy = fabsf(i10) + fabsf(i11) + 1e-3f * (1.f + fabsf(i10) + fabsf(i11));
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < IW; j += gridDim.x * blockDim.x) {
        const float du = j - b0, dv = i - b1;
        const float cx = i00 * du + i01 * dv, cy = i10 * du + i11 * dv;
        // NaN / infinite bounds (singular map) fall back to the whole output
        int x0 = 0, x1 = OW - 1, y0 = 0, y1 = OH - 1;
        if (isfinite(cx) && isfinite(hx)) { x0 = max(0, (int)ceilf(fmaxf(cx - hx, -1.f))); x1 = min(OW - 1, (int)floorf(fminf(cx + hx, (float)OW))); }
        if (isfinite(cy) && isfinite(hy)) { y0 = max(0, (int)ceilf(fmaxf(cy - hy, -1.f))); y1 = min(OH - 1, (int)floorf(fminf(cy + hy, (float)OH))); }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};       // C <= 4 per pass
        for (int c0 = 0; c0 < C; c0 += 4) {
            const int cn = min(4, C - c0);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = 0.f;
            for (int oy = y0; oy <= y1; oy++)
                for (int ox = x0; ox <= x1; ox++) {
                    float u, v;
                    affine_uv(th, ox, oy, IW, IH, OW, OH, u, v);
                    if (!(fabsf(u) < 1e8f && fabsf(v) < 1e8f)) continue;
                    // weight of input pixel (j, i) in the sample at (u, v): the forward's (1 - frac) / frac factors
                    const float fu = floorf(u), fv = floorf(v);
                    const int j0 = (int)fu, i0 = (int)fv;
                    float wx, wy;
                    if (j == j0) wx = 1.f - (u - fu); else if (j == j0 + 1) wx = u - fu; else continue;
                    if (i == i0) wy = 1.f - (v - fv); else if (i == i0 + 1) wy = v - fv; else continue;
                    const float w = wx * wy;
                    for (int k = 0; k < cn; k++)
                        acc[k] = fmaf(dy[(((int64_t)n * C + c0 + k) * OH + oy) * OW + ox], w, acc[k]);
                }
            for (int k = 0; k < cn; k++) dx[(((int64_t)n * C + c0 + k) * IH + i) * IW + j] = acc[k];
        }
    }
}

// mode 0: out[n, i, p] = sum_k M[n][i][k] x[n, k, p] + M[n][i][3];  mode 1: the adjoint (M[n][k][i], no offset);
// mode 2: the linear part alone (the adjoint's adjoint) -- first and second derivatives with respect to the image.
template <int V>
__global__ __launch_bounds__(256) void color_affine_kernel(const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ out,
                                                           int64_t hw, int mode) {
    const int n = blockIdx.y;
    const bool transpose = mode == 1;
    const float* cm = c + (int64_t)n * 16;
    float m[3][3], t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 3; k++) m[i][k] = transpose ? cm[k * 4 + i] : cm[i * 4 + k];
        t[i] = mode == 0 ? cm[i * 4 + 3] : 0.f;
    }
    typedef Pack<float, V> P;
    const float* xn = x + (int64_t)n * 3 * hw;
    float* on = out + (int64_t)n * 3 * hw;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < hw; i += (int64_t)gridDim.x * blockDim.x * V) {
        const P r = *(const P*)(xn + i), g = *(const P*)(xn + hw + i), b = *(const P*)(xn + 2 * hw + i);
        P o0, o1, o2;
#pragma unroll
        for (int k = 0; k < V; k++) {
            o0.v[k] = fmaf(m[0][0], r.v[k], fmaf(m[0][1], g.v[k], fmaf(m[0][2], b.v[k], t[0])));
            o1.v[k] = fmaf(m[1][0], r.v[k], fmaf(m[1][1], g.v[k], fmaf(m[1][2], b.v[k], t[1])));
            o2.v[k] = fmaf(m[2][0], r.v[k], fmaf(m[2][1], g.v[k], fmaf(m[2][2], b.v[k], t[2])));
        }
        *(P*)(on + i) = o0; *(P*)(on + hw + i) = o1; *(P*)(on + 2 * hw + i) = o2;
    }
}

}  // namespace pasta

using namespace pasta;

extern "C" int pasta_ada_matrices(const float* u, const float* z, int64_t n, int u_cols, int z_cols, const float* p,
                                  const pasta_ada_config* cfg, int width, int height, int channels, int hz_pad,
                                  float debug_percentile, float* g_inv, float* c, int32_t* margins, void* stream) {
    PASTA_CHECK(u && z && p && cfg && g_inv && c && margins, "ada_matrices: null pointer");
    PASTA_CHECK(n >= 1 && n <= (1 << 24), "ada_matrices: batch of %lld samples", (long long)n);
    PASTA_CHECK(u_cols >= U_COLS_MIN && z_cols >= Z_COLS_MIN, "ada_matrices: %d uniform / %d normal columns, need >= %d / %d",
                u_cols, z_cols, (int)U_COLS_MIN, (int)Z_COLS_MIN);
    PASTA_CHECK(width >= 1 && height >= 1 && channels >= 1 && hz_pad >= 0, "ada_matrices: bad image geometry");
    PASTA_CHECK(debug_percentile < 0.f || debug_percentile <= 1.f, "ada_matrices: debug percentile %g outside [0, 1]", debug_percentile);
    hipLaunchKernelGGL(ada_matrices_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, u, z, (int)n, u_cols, z_cols, p, *cfg, width,
                       height, channels, hz_pad, debug_percentile, g_inv, c, margins);
    return launch_status("ada_matrices");
}

extern "C" int pasta_ada_theta(const float* g_inv, int64_t n, const float* a, const float* b, float* theta, void* stream) {
    PASTA_CHECK(g_inv && a && b && theta, "ada_theta: null pointer");
    PASTA_CHECK(n >= 1 && n <= (1 << 24), "ada_theta: batch of %lld samples", (long long)n);
    ThetaParams prm;
    for (int k = 0; k < 9; k++) { prm.a[k] = a[k]; prm.b[k] = b[k]; }      // a, b: host arrays
    hipLaunchKernelGGL(ada_theta_kernel, dim3((unsigned)ceil_div64(n, 64)), dim3(64), 0, (hipStream_t)stream, g_inv, (int)n, prm, theta);
    return launch_status("ada_theta");
}

extern "C" int pasta_color_affine(const float* x, const float* c, float* out, int64_t n, int64_t hw, int mode, void* stream) {
    PASTA_CHECK(x && c && out, "color_affine: null pointer");
    PASTA_CHECK(mode >= 0 && mode <= 2, "color_affine: mode %d (0 affine, 1 adjoint, 2 linear part)", mode);
    PASTA_CHECK(n >= 1 && n <= 65535 && hw >= 1, "color_affine: %lld images of %lld pixels", (long long)n, (long long)hw);
    const bool v4 = hw % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int64_t per_block = 256 * (v4 ? 4 : 1) * 4;
    const unsigned gx = (unsigned)(ceil_div64(hw, per_block) < 1024 ? ceil_div64(hw, per_block) : 1024);
    if (v4) hipLaunchKernelGGL((color_affine_kernel<4>), dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, c, out, hw, mode);
    else    hipLaunchKernelGGL((color_affine_kernel<1>), dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, c, out, hw, mode);
    return launch_status("color_affine");
}

extern "C" int pasta_ada_grid(const float* theta, int64_t n, int H, int W, float* grid, void* stream) {
    PASTA_CHECK(theta && grid, "ada_grid: null pointer");
    PASTA_CHECK(n >= 1 && n <= 65535 && H >= 1 && H <= 65535 && W >= 1, "ada_grid: %lld grids of %d x %d", (long long)n, H, W);
    PASTA_CHECK(((uintptr_t)grid & 15) == 0, "ada_grid: output must be 16-byte aligned");
    const unsigned gx = (unsigned)ceil_div64(ceil_div64(W, 2), 256);
    hipLaunchKernelGGL(ada_grid_kernel, dim3(gx, (unsigned)H, (unsigned)n), dim3(256), 0, (hipStream_t)stream, theta, H, W, grid);
    return launch_status("ada_grid");
}

static int affine_sample_check(const char* who, const void* a, const void* b, const void* c, int64_t n, int C, int IH, int IW, int OH, int OW) {
    PASTA_CHECK(a && b && c, "%s: null pointer", who);
    PASTA_CHECK(n >= 1 && n <= 65535 && C >= 1 && IH >= 1 && IH <= 65535 && IW >= 1 && OH >= 1 && OH <= 65535 && OW >= 1,
                "%s: %lld images [%d, %d, %d] -> [%d, %d]", who, (long long)n, C, IH, IW, OH, OW);
    return 0;
}

extern "C" int pasta_affine_sample(const float* x, const float* theta, float* y, int64_t n, int C, int IH, int IW, int OH, int OW, void* stream) {
    if (int e = affine_sample_check("affine_sample", x, theta, y, n, C, IH, IW, OH, OW)) return e;
    hipLaunchKernelGGL(affine_sample_kernel, dim3((unsigned)ceil_div64(OW, 256), (unsigned)OH, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       x, theta, y, C, IH, IW, OH, OW);
    return launch_status("affine_sample");
}

extern "C" int pasta_affine_sample_adjoint(const float* dy, const float* theta, float* dx, int64_t n, int C, int IH, int IW, int OH, int OW,
                                           void* stream) {
    if (int e = affine_sample_check("affine_sample_adjoint", dy, theta, dx, n, C, IH, IW, OH, OW)) return e;
    hipLaunchKernelGGL(affine_sample_adjoint_kernel, dim3((unsigned)ceil_div64(IW, 256), (unsigned)IH, (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, dy, theta, dx, C, IH, IW, OH, OW);
    return launch_status("affine_sample_adjoint");
}
