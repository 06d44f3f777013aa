// Per-(sample, channel)-plane kernels: channel scaling (+ noise), plane dot products and the
// SPADE instance-norm modulation.  All HBM-bound; each reads/writes its operands exactly once.
//
//   scale_add    y = x * a[n,c] + b[n,hw]        modulation / demodulation + noise
//                                                 (training/networks.py:74, 77-79; fma.py:15)
//   plane_dot    out[n,c] = sum_hw p * q          gradients of the per-channel scales (fma.py:44-50)
//   spade_norm   out = (x-mean)*rstd*(1+gamma)+beta   (training/networks.py:4371-4379)
#include "common.h"

namespace pasta {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over a workgroup of NT threads; every thread gets the total. `red` holds NT/64 floats.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) t += red[i];
    return t;
}

//------------------------------------------------------------------------------------
// y[plane, hw] = x[plane, hw] * a[plane] + b[(per_sample ? n : 0), hw]
// grid.x = chunks of a plane, grid.y = planes (strided).  16-byte accesses when HW % 4 == 0.
// y_amax stays unused by the Python layer: a grid of a few workgroups per CU that walk the planes and commit one |max| each was
// measured (round 3) at 57.7 - 59.2 us against 42.2 on [16, 128, 128, 128] -- the 17 us are two thirds of the scan they replace.

// T = storage type of x, b and y (float, __half, __bf16); the scales a are fp32, the arithmetic is fp32.
template <class T, int V>
__global__ __launch_bounds__(256) void scale_add_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                        const T* __restrict__ b, T* __restrict__ y,
                                                        int64_t planes, int C, int64_t HW, int b_per_sample, float* __restrict__ y_amax) {
    const int64_t hwv = HW / V;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(y_amax);
    for (int64_t plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const float s = a ? a[plane] : 1.f;
        const T* xp = x + plane * HW;
        T* yp = y + plane * HW;
        const T* bp = b ? b + (b_per_sample ? (plane / C) * HW : 0) : nullptr;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hwv; i += (int64_t)gridDim.x * 256) {
            if constexpr (V == 4) {
                float4 v = ld4<T>(xp + 4 * i);
                const float4 w = bp ? ld4<T>(bp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 o = make_float4(fmaf(v.x, s, w.x), fmaf(v.y, s, w.y), fmaf(v.z, s, w.z), fmaf(v.w, s, w.w));
                st4<T>(yp + 4 * i, o);
                if (y_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
            } else {
                const float o = fmaf(ld<T>(xp + i), s, bp ? ld<T>(bp + i) : 0.f);
                st<T>(yp + i, o);
                if (y_amax) amax_take(am, o);
            }
        }
    }
    amax_commit(am, aslot);
}

//------------------------------------------------------------------------------------
// out[plane] = sum_hw p*q (q may be null).  One workgroup per plane, fixed summation order.

template <class T>
__global__ __launch_bounds__(256) void plane_dot_kernel(const T* __restrict__ p, const T* __restrict__ q,
                                                        float* __restrict__ out, int64_t planes, int64_t HW) {
    __shared__ float red[4];
    for (int64_t plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        const T* pp = p + plane * HW;
        const T* qp = q ? q + plane * HW : nullptr;
        float acc = 0.f;
        if ((HW & 3) == 0) {
            for (int64_t i = threadIdx.x; i < HW / 4; i += 256) {
                float4 a = ld4<T>(pp + 4 * i);
                if (qp) { float4 b = ld4<T>(qp + 4 * i); acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
                else acc += a.x + a.y + a.z + a.w;
            }
        } else {
            for (int64_t i = threadIdx.x; i < HW; i += 256) acc += qp ? ld<T>(pp + i) * ld<T>(qp + i) : ld<T>(pp + i);
        }
        float t = block_sum<256>(acc, red);
        if (threadIdx.x == 0) out[plane] = t;
    }
}

//------------------------------------------------------------------------------------
// SPADE normalisation.  One 1024-thread workgroup per plane; when the plane is exactly
// EPT*1024 floats (EPT = 16 is the 128x128 plane of the generator) it lives in registers
// between the statistics and the apply pass, otherwise later passes re-read it through L2.

template <class T, int EPT>   // EPT == 0: generic loops; T = storage type of x, gamma, beta, out (statistics and arithmetic: fp32)
__global__ __launch_bounds__(1024) void spade_norm_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                          const T* __restrict__ beta, T* __restrict__ out,
                                                          float* __restrict__ stats, int64_t planes, int64_t HW, float eps,
                                                          int act, float gain, float clamp, int C, int64_t gb_ns, float* __restrict__ y_amax) {
    __shared__ float red[16];
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(y_amax);
    // optional relu * gain with clamp on the way out (the activation Spade_Conv2dLayer applies before its convolution)
    auto post = [&](float v) {
        if (act == 2) { v = v > 0.f ? v * gain : 0.f; if (clamp >= 0.f && v > clamp) v = clamp; }
        return v;
    };
    const int tid = threadIdx.x;
    for (int64_t plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        const T* xp = x + plane * HW;
        const float inv = 1.f / (float)HW;
        // gamma / beta may be the two channel halves of one [N, 2C, H, W] tensor: sample stride gb_ns instead of C * HW
        const int64_t gb = (plane / C) * gb_ns + (plane % C) * HW;
        float mean, rstd;
        if constexpr (EPT > 0) {
            float4 r[EPT / 4], gq[EPT / 4], bq[EPT / 4];
            float s = 0.f;
            // gamma and beta are fetched WITH the plane, ahead of the two workgroup reductions: three operands in flight instead of one
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) r[k] = ld4<T>(xp + 4 * (k * 1024 + tid));
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const int64_t i = k * 1024 + tid;
                gq[k] = gamma ? ld4<T>(gamma + gb + 4 * i) : make_float4(0, 0, 0, 0);
                bq[k] = beta ? ld4<T>(beta + gb + 4 * i) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) s += r[k].x + r[k].y + r[k].z + r[k].w;
            mean = block_sum<1024>(s, red) * inv;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                float a = r[k].x - mean, b = r[k].y - mean, c = r[k].z - mean, d = r[k].w - mean;
                q += a * a + b * b + c * c + d * d;
            }
            rstd = rsqrtf(block_sum<1024>(q, red) * inv + eps);
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const int64_t i = k * 1024 + tid;
                const float4 g = gq[k], b = bq[k];
                float4 o;
                o.x = post(fmaf((r[k].x - mean) * rstd, 1.f + g.x, b.x));
                o.y = post(fmaf((r[k].y - mean) * rstd, 1.f + g.y, b.y));
                o.z = post(fmaf((r[k].z - mean) * rstd, 1.f + g.z, b.z));
                o.w = post(fmaf((r[k].w - mean) * rstd, 1.f + g.w, b.w));
                st4<T>(out + plane * HW + 4 * i, o);
                if (y_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
            }
        } else {
            float s = 0.f;
            for (int64_t i = tid; i < HW; i += 1024) s += ld<T>(xp + i);
            mean = block_sum<1024>(s, red) * inv;
            float q = 0.f;
            for (int64_t i = tid; i < HW; i += 1024) { float d = ld<T>(xp + i) - mean; q += d * d; }
            rstd = rsqrtf(block_sum<1024>(q, red) * inv + eps);
            for (int64_t i = tid; i < HW; i += 1024) {
                float g = gamma ? ld<T>(gamma + gb + i) : 0.f, b = beta ? ld<T>(beta + gb + i) : 0.f;
                const float o = post(fmaf((ld<T>(xp + i) - mean) * rstd, 1.f + g, b));
                st<T>(out + plane * HW + i, o);
                if (y_amax) amax_take(am, o);
            }
        }
        if (tid == 0 && stats) { stats[2 * plane] = mean; stats[2 * plane + 1] = rstd; }
    }
    __shared__ uint32_t amred[16];
    amax_commit_block<1024>(am, aslot, amred);        // one commit per workgroup (sixteen waves)
}

// Backward.  xhat = (x-mean)*rstd, t = dout*(1+gamma):
//   dgamma = dout*xhat, dbeta = dout, dx = rstd*(t - mean(t) - xhat*mean(t*xhat)).
// dgamma/dbeta/dx may each be null (not needed).
template <class T, int EPT>
__global__ __launch_bounds__(1024) void spade_norm_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                              const T* __restrict__ gamma, const float* __restrict__ stats,
                                                              T* __restrict__ dx, T* __restrict__ dgamma,
                                                              T* __restrict__ dbeta, int64_t planes, int64_t HW,
                                                              const T* __restrict__ beta, int act, float gain, float clamp,
                                                              int C, int64_t gb_ns, int64_t dgb_ns, float* __restrict__ dx_amax,
                                                              float* __restrict__ dgb_amax, const T* __restrict__ dx_add) {
    __shared__ float red[16];
    uint32_t am = 0, am2 = 0;                       // |max| of dx; of everything written to dgamma and dbeta (one tensor when they are halves)
    const AmaxSlot aslot = amax_begin(dx_amax);
    const AmaxSlot aslot2 = amax_begin(dgb_amax);
    // gradient through the optional relu * gain / clamp of the forward: v = x_hat * (1 + gamma) + beta is recomputed
    auto pre = [&](float d, float h, float g, float b) {
        if (act == 2) {
            const float v = fmaf(h, 1.f + g, b);
            d = (v > 0.f && (clamp < 0.f || v * gain < clamp)) ? d * gain : 0.f;
        }
        return d;
    };
    const int tid = threadIdx.x;
    for (int64_t plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
        const float inv = 1.f / (float)HW;
        const int64_t base = plane * HW;
        const int64_t gb = (plane / C) * gb_ns + (plane % C) * HW;        // gamma / beta (see the forward kernel)
        const int64_t dgb = (plane / C) * dgb_ns + (plane % C) * HW;      // dgamma / dbeta
        if constexpr (EPT > 0) {
            float xh[EPT], t[EPT];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const int64_t i = k * 1024 + tid;
                float4 xv = ld4<T>(x + base + 4 * i);
                float4 dv = ld4<T>(dout + base + 4 * i);
                float4 gv = gamma ? ld4<T>(gamma + gb + 4 * i) : make_float4(0, 0, 0, 0);
                float4 bv = (act == 2 && beta) ? ld4<T>(beta + gb + 4 * i) : make_float4(0, 0, 0, 0);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
                float ds[4] = {dv.x, dv.y, dv.z, dv.w};
                float dg[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float h = (xs[j] - mean) * rstd;
                    ds[j] = pre(ds[j], h, gs[j], bs[j]);
                    float tt = ds[j] * (1.f + gs[j]);
                    xh[k * 4 + j] = h; t[k * 4 + j] = tt;
                    dg[j] = ds[j] * h;
                    s1 += tt; s2 += tt * h;
                }
                if (dgamma) st4<T>(dgamma + dgb + 4 * i, make_float4(dg[0], dg[1], dg[2], dg[3]));
                if (dbeta) st4<T>(dbeta + dgb + 4 * i, make_float4(ds[0], ds[1], ds[2], ds[3]));
                if (dgb_amax) {
#pragma unroll
                    for (int j = 0; j < 4; j++) { if (dgamma) amax_take(am2, dg[j]); if (dbeta) amax_take(am2, ds[j]); }
                }
            }
            if (dx) {
                const float m1 = block_sum<1024>(s1, red) * inv, m2 = block_sum<1024>(s2, red) * inv;
#pragma unroll
                for (int k = 0; k < EPT / 4; k++) {
                    float4 o;
                    o.x = rstd * (t[k * 4 + 0] - m1 - xh[k * 4 + 0] * m2);
                    o.y = rstd * (t[k * 4 + 1] - m1 - xh[k * 4 + 1] * m2);
                    o.z = rstd * (t[k * 4 + 2] - m1 - xh[k * 4 + 2] * m2);
                    o.w = rstd * (t[k * 4 + 3] - m1 - xh[k * 4 + 3] * m2);
                    if (dx_add) { const float4 c = ld4<T>(dx_add + base + 4 * (k * 1024 + tid)); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }     // another consumer's gradient of x
                    st4<T>(dx + base + 4 * (k * 1024 + tid), o);
                    if (dx_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
                }
            }
        } else {
            float s1 = 0.f, s2 = 0.f;
            for (int64_t i = tid; i < HW; i += 1024) {
                float h = (ld<T>(x + base + i) - mean) * rstd;
                const float gi = gamma ? ld<T>(gamma + gb + i) : 0.f;
                float d = pre(ld<T>(dout + base + i), h, gi, (act == 2 && beta) ? ld<T>(beta + gb + i) : 0.f);
                float tt = d * (1.f + gi);
                s1 += tt; s2 += tt * h;
                if (dgamma) st<T>(dgamma + dgb + i, d * h);
                if (dbeta) st<T>(dbeta + dgb + i, d);
                if (dgb_amax) { if (dgamma) amax_take(am2, d * h); if (dbeta) amax_take(am2, d); }
            }
            if (dx) {
                const float m1 = block_sum<1024>(s1, red) * inv, m2 = block_sum<1024>(s2, red) * inv;
                for (int64_t i = tid; i < HW; i += 1024) {
                    float h = (ld<T>(x + base + i) - mean) * rstd;
                    const float gi = gamma ? ld<T>(gamma + gb + i) : 0.f;
                    float d = pre(ld<T>(dout + base + i), h, gi, (act == 2 && beta) ? ld<T>(beta + gb + i) : 0.f);
                    float tt = d * (1.f + gi);
                    float o = rstd * (tt - m1 - h * m2);
                    if (dx_add) o += ld<T>(dx_add + base + i);
                    st<T>(dx + base + i, o);
                    if (dx_amax) amax_take(am, o);
                }
            }
        }
    }
    __shared__ uint32_t amred[16];
    amax_commit_block<1024>(am, aslot, amred);        // one commit per workgroup (sixteen waves)
    __shared__ uint32_t amred2[16];
    amax_commit_block<1024>(am2, aslot2, amred2);
}


//------------------------------------------------------------------------------------
// Garment features of the SPADE stage (training/networks.py:5777-5800): where the predicted region is not covered by a patch
// (hole) the feature is replaced by the mean feature of the covered part (valid):
//   out[n,c,i] = x[n,c,i] * (1 - hole[n,i]) + hole[n,i] * inv_count[n] * sum_j x[n,c,j] * valid[n,j]
// The reference forms this with six element-wise / reduction passes over the 134 MB feature map and a torch.cat of the two
// garments' results; here one workgroup per (n, c) plane holds the plane in registers between the sum and the fill, and writes
// straight into its half of the concatenated tensor (out: sample stride out_ns).  Backward:
//   dx[n,c,i] = dout[n,c,i] * (1 - hole[n,i]) + valid[n,i] * inv_count[n] * sum_j dout[n,c,j] * hole[n,j].
// BWD = false: a = x, m1 = valid (weights of the sum), m2 = hole (where the mean goes); BWD = true: a = dout, m1 = hole, m2 = valid.
// The factor (1 - hole) multiplies a in both.
template <int EPT, bool BWD>       // EPT == 0: generic loops
__global__ __launch_bounds__(1024) void masked_mean_fill_kernel(const float* __restrict__ a, const float* __restrict__ valid,
                                                                const float* __restrict__ hole, const float* __restrict__ inv_count,
                                                                float* __restrict__ out, int64_t planes, int C, int64_t HW, int64_t a_ns,
                                                                int64_t out_ns, float* __restrict__ y_amax) {
    __shared__ float red[16];
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(y_amax);
    const int tid = threadIdx.x;
    for (int64_t plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        const int64_t n = plane / C, c = plane - n * C;
        const float* ap = a + n * a_ns + c * HW;
        const float* vp = valid + n * HW;
        const float* hp = hole + n * HW;
        float* op = out + n * out_ns + c * HW;
        const float ic = inv_count[n];
        if constexpr (EPT > 0) {
            float4 r[EPT / 4], v[EPT / 4], h[EPT / 4];
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const int64_t i = 4 * (k * 1024 + tid);
                r[k] = *(const float4*)(ap + i); v[k] = *(const float4*)(vp + i); h[k] = *(const float4*)(hp + i);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const float4 w = BWD ? h[k] : v[k];
                s += r[k].x * w.x + r[k].y * w.y + r[k].z * w.z + r[k].w * w.w;
            }
            const float t = block_sum<1024>(s, red) * ic;
#pragma unroll
            for (int k = 0; k < EPT / 4; k++) {
                const float4 w = BWD ? v[k] : h[k];
                float4 o;
                o.x = fmaf(r[k].x, 1.f - h[k].x, t * w.x); o.y = fmaf(r[k].y, 1.f - h[k].y, t * w.y);
                o.z = fmaf(r[k].z, 1.f - h[k].z, t * w.z); o.w = fmaf(r[k].w, 1.f - h[k].w, t * w.w);
                *(float4*)(op + 4 * (k * 1024 + tid)) = o;
                if (y_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
            }
        } else {
            float s = 0.f;
            for (int64_t i = tid; i < HW; i += 1024) s += ap[i] * (BWD ? hp[i] : vp[i]);
            const float t = block_sum<1024>(s, red) * ic;
            for (int64_t i = tid; i < HW; i += 1024) {
                const float o = fmaf(ap[i], 1.f - hp[i], t * (BWD ? vp[i] : hp[i]));
                op[i] = o;
                if (y_amax) amax_take(am, o);
            }
        }
    }
    __shared__ uint32_t amred[16];
    amax_commit_block<1024>(am, aslot, amred);
}

//------------------------------------------------------------------------------------
// Tail of a modulated convolution layer in one pass (SynthesisLayer, training/networks.py:72-82 + 313-314):
//   y = clamp(act(u * d[n,c] + noise[n|.,hw] * strength + b[c]) * gain),   act = linear (1) or leaky relu (3)
// and its backward: with dz = dy * act'(y) * gain (zero where |y| >= clamp)
//   du = dz * d[n,c],   partial[n,c][chunk] = (sum dz*u, sum dz*noise, sum dz)  -> dd[n,c], dstrength, db[c] on the host.
// One workgroup per (plane, chunk of 4096 elements); 16-byte accesses when HW % 4 == 0.
constexpr int MBA_CHUNK = 4096;

__device__ __forceinline__ float mba_fwd(float u, float d, float nz, float b, int act, float alpha, float gain, float clamp) {
    float v = fmaf(u, d, nz) + b;
    if (act == 3) v = v > 0.f ? v : v * alpha;
    v *= gain;
    if (clamp >= 0.f) v = (v > -clamp && v < clamp) ? v : (v >= 0.f ? clamp : -clamp);
    return v;
}

template <class T>      // storage type of u and y; d, noise, strength and b are fp32
__global__ __launch_bounds__(256) void mod_bias_act_kernel(const T* __restrict__ u, const float* __restrict__ d,
                                                           const float* __restrict__ noise, const float* __restrict__ strength,
                                                           const float* __restrict__ b, T* __restrict__ y, int C, int64_t HW,
                                                           int noise_per_sample, int act, float alpha, float gain, float clamp, float* __restrict__ y_amax) {
    const int64_t plane = blockIdx.x;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(y_amax);
    const int n = (int)(plane / C), c = (int)(plane - (int64_t)n * C);
    const float dv = d ? d[plane] : 1.f, bv = b ? b[c] : 0.f, ns = noise ? strength[0] : 0.f;
    const T* up = u + plane * HW;
    const float* np_ = noise ? noise + (noise_per_sample ? (int64_t)n * HW : 0) : nullptr;
    T* yp = y + plane * HW;
    const int64_t i0 = (int64_t)blockIdx.y * MBA_CHUNK, i1 = i0 + MBA_CHUNK < HW ? i0 + MBA_CHUNK : HW;
    if ((HW & 3) == 0) {
        // the chunk's four fetches per operand first, then the arithmetic and the stores: 64 - 128 bytes in flight per thread
        constexpr int U = MBA_CHUNK / 1024;
        float4 uv[U], nv[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t i = i0 + 4 * threadIdx.x + 1024 * k, ic = i < i1 ? i : i1 - 4;
            uv[k] = ld4<T>(up + ic);
            nv[k] = np_ ? *(const float4*)(np_ + ic) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t i = i0 + 4 * threadIdx.x + 1024 * k;
            if (i >= i1) continue;
            float4 o;
            o.x = mba_fwd(uv[k].x, dv, nv[k].x * ns, bv, act, alpha, gain, clamp);
            o.y = mba_fwd(uv[k].y, dv, nv[k].y * ns, bv, act, alpha, gain, clamp);
            o.z = mba_fwd(uv[k].z, dv, nv[k].z * ns, bv, act, alpha, gain, clamp);
            o.w = mba_fwd(uv[k].w, dv, nv[k].w * ns, bv, act, alpha, gain, clamp);
            st4<T>(yp + i, o);
            if (y_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float o = mba_fwd(ld<T>(up + i), dv, np_ ? np_[i] * ns : 0.f, bv, act, alpha, gain, clamp);
            st<T>(yp + i, o);
            if (y_amax) amax_take(am, o);
        }
    }
    amax_commit(am, aslot);
}

__device__ __forceinline__ float mba_dz(float dy, float y, int act, float alpha, float gain, float clamp) {
    float g = dy * gain;
    if (act == 3) g = y > 0.f ? g : g * alpha;
    if (clamp >= 0.f) g = (y > -clamp && y < clamp) ? g : 0.f;
    return g;
}

template <class T>      // storage type of dy, y, u and du
__global__ __launch_bounds__(256) void mod_bias_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                               const T* __restrict__ u, const float* __restrict__ d,
                                                               const float* __restrict__ noise, T* __restrict__ du,
                                                               float* __restrict__ partial, int C, int64_t HW, int chunks,
                                                               int noise_per_sample, int act, float alpha, float gain, float clamp, float* __restrict__ du_amax) {
    __shared__ float red[4];
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(du_amax);
    const int64_t plane = blockIdx.x;
    const int n = (int)(plane / C);
    const float dv = d ? d[plane] : 1.f;
    const T* dyp = dy + plane * HW; const T* yp = y + plane * HW; const T* up = u + plane * HW;
    const float* np_ = noise ? noise + (noise_per_sample ? (int64_t)n * HW : 0) : nullptr;
    T* dup = du + plane * HW;
    const int64_t i0 = (int64_t)blockIdx.y * MBA_CHUNK, i1 = i0 + MBA_CHUNK < HW ? i0 + MBA_CHUNK : HW;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if ((HW & 3) == 0) {
        constexpr int U = MBA_CHUNK / 1024;             // fetches first, as in the forward kernel
        float4 gvs[U], yvs[U], uvs[U], nvs[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t i = i0 + 4 * threadIdx.x + 1024 * k, ic = i < i1 ? i : i1 - 4;
            gvs[k] = ld4<T>(dyp + ic); yvs[k] = ld4<T>(yp + ic); uvs[k] = ld4<T>(up + ic);
            nvs[k] = np_ ? *(const float4*)(np_ + ic) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int64_t i = i0 + 4 * threadIdx.x + 1024 * k;
            if (i >= i1) continue;
            const float4 gv = gvs[k], yv = yvs[k], uv = uvs[k], nv = nvs[k];
            const float z0 = mba_dz(gv.x, yv.x, act, alpha, gain, clamp), z1 = mba_dz(gv.y, yv.y, act, alpha, gain, clamp);
            const float z2 = mba_dz(gv.z, yv.z, act, alpha, gain, clamp), z3 = mba_dz(gv.w, yv.w, act, alpha, gain, clamp);
            s0 += z0 * uv.x + z1 * uv.y + z2 * uv.z + z3 * uv.w;
            s1 += z0 * nv.x + z1 * nv.y + z2 * nv.z + z3 * nv.w;
            s2 += z0 + z1 + z2 + z3;
            const float4 o = make_float4(z0 * dv, z1 * dv, z2 * dv, z3 * dv);
            st4<T>(dup + i, o);
            if (du_amax) { amax_take(am, o.x); amax_take(am, o.y); amax_take(am, o.z); amax_take(am, o.w); }
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float z = mba_dz(ld<T>(dyp + i), ld<T>(yp + i), act, alpha, gain, clamp);
            s0 += z * ld<T>(up + i); s1 += np_ ? z * np_[i] : 0.f; s2 += z;
            st<T>(dup + i, z * dv);
            if (du_amax) amax_take(am, z * dv);
        }
    }
    __shared__ uint32_t amred[4];
    amax_commit_block<256>(am, aslot, amred);
    s0 = block_sum<256>(s0, red); s1 = block_sum<256>(s1, red); s2 = block_sum<256>(s2, red);
    if (threadIdx.x == 0) {
        float* o = partial + (plane * chunks + blockIdx.y) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
    }
}

//------------------------------------------------------------------------------------
// Demodulation coefficients of the modulated convolution (networks.py:65-68):
//   d[n,o] = rsqrt(sum_{i,k} (w[o,i,k] * s[n,i])^2 + eps) = rsqrt(sum_i s[n,i]^2 * W2[o,i] + eps),  W2[o,i] = sum_k w[o,i,k]^2.
// One workgroup per output channel: the tap-summed squared weights of the channel are formed once in LDS (the [N,O,I,k,k]
// per-sample weight tensor of the reference never exists), then each wave takes samples in turn and reduces over the
// input channels with wavefront shuffles.

constexpr int DEMOD_MAX_I = 4096;

__global__ __launch_bounds__(256) void demod_coefs_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ d,
                                                          int N, int O, int I, int KK, float eps) {
    __shared__ float w2[DEMOD_MAX_I];
    const int o = blockIdx.x;
    const float* wo = w + (int64_t)o * I * KK;
    for (int i = threadIdx.x; i < I; i += 256) {
        float t = 0.f;
        for (int k = 0; k < KK; k++) { const float v = wo[(int64_t)i * KK + k]; t += v * v; }
        w2[i] = t;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = wave; n < N; n += 4) {
        const float* sn = s + (int64_t)n * I;
        float t = 0.f;
        for (int i = lane; i < I; i += 64) { const float v = sn[i]; t += v * v * w2[i]; }
        t = wave_sum(t);
        if (lane == 0) d[(int64_t)n * O + o] = rsqrtf(t + eps);
    }
}

}  // namespace pasta

// runs LAUNCH_(T) with T the storage type of dtype code `dtype` (PASTA_F32 / _F16 / _BF16)
#define PASTA_BY_DTYPE(dtype, what, LAUNCH_)                                                             \
    switch (dtype) {                                                                                     \
        case PASTA_F32: { LAUNCH_(float); break; }                                                       \
        case PASTA_F16: { LAUNCH_(__half); break; }                                                      \
        case PASTA_BF16: { LAUNCH_(__bf16); break; }                                                     \
        default: return ::pasta::fail(what ": unsupported dtype code %d", dtype);                        \
    }

extern "C" int pasta_scale_add(const void* x, const float* a, const void* b, void* y, int dtype, int N, int C, int64_t HW,
                               int b_per_sample, void* stream, float* y_amax) {
    using namespace pasta;
    PASTA_CHECK(x && y, "scale_add: null pointer");
    PASTA_CHECK(N >= 1 && C >= 1 && HW >= 1, "scale_add: empty tensor");
    const int64_t planes = (int64_t)N * C;
    const bool vec = HW % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)b) & 15) == 0;
    const int64_t per_plane = vec ? HW / 4 : HW;
    int gx = (int)(ceil_div64(per_plane, 256) < 64 ? ceil_div64(per_plane, 256) : 64);
    int gy = (int)(planes < 65535 ? planes : 65535);
    hipStream_t s = (hipStream_t)stream;
#define PASTA_L(T)                                                                                                                         \
    if (vec) hipLaunchKernelGGL((scale_add_kernel<T, 4>), dim3(gx, gy), dim3(256), 0, s, (const T*)x, a, (const T*)b, (T*)y, planes, C, HW, b_per_sample, y_amax); \
    else     hipLaunchKernelGGL((scale_add_kernel<T, 1>), dim3(gx, gy), dim3(256), 0, s, (const T*)x, a, (const T*)b, (T*)y, planes, C, HW, b_per_sample, y_amax)
    PASTA_BY_DTYPE(dtype, "scale_add", PASTA_L)
#undef PASTA_L
    return launch_status("scale_add");
}

extern "C" int pasta_plane_dot(const void* p, const void* q, float* out, int dtype, int64_t planes, int64_t HW, void* stream) {
    using namespace pasta;
    PASTA_CHECK(p && out, "plane_dot: null pointer");
    PASTA_CHECK(planes >= 1 && HW >= 1, "plane_dot: empty tensor");
    PASTA_CHECK((((uintptr_t)p | (uintptr_t)q) & 15) == 0 || (HW & 3) != 0, "plane_dot: operands must be 16-byte aligned");
    int grid = (int)(planes < 65535 ? planes : 65535);
#define PASTA_L(T) hipLaunchKernelGGL(plane_dot_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)p, (const T*)q, out, planes, HW)
    PASTA_BY_DTYPE(dtype, "plane_dot", PASTA_L)
#undef PASTA_L
    return launch_status("plane_dot");
}

extern "C" int pasta_demod_coefs(const float* w, const float* styles, float* d, int N, int O, int I, int KK, float eps, void* stream) {
    using namespace pasta;
    PASTA_CHECK(w && styles && d, "demod_coefs: null pointer");
    PASTA_CHECK(N >= 1 && O >= 1 && I >= 1 && KK >= 1, "demod_coefs: empty tensor");
    PASTA_CHECK(I <= DEMOD_MAX_I, "demod_coefs: %d input channels, at most %d", I, DEMOD_MAX_I);
    hipLaunchKernelGGL(demod_coefs_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, styles, d, N, O, I, KK, eps);
    return launch_status("demod_coefs");
}

//------------------------------------------------------------------------------------
// nan_to_num over a list of tensors in one launch (training_loop_wo_flow_fullbody.py:513-515 applies it to every
// parameter gradient before the optimiser step: ~280 three-microsecond launches per iteration otherwise).  The tensor
// table travels in the kernel arguments; a workgroup owns one 16 K-element chunk of one tensor.

constexpr int NTN_MAX = 96;                 // tensors per launch (1.6 KB of kernel arguments)
constexpr int NTN_CHUNK = 16384;
struct NanToNumTable {
    float* ptr[NTN_MAX];
    int32_t numel[NTN_MAX];
    int32_t chunk0[NTN_MAX + 1];            // first chunk (= workgroup) of tensor t; chunk0[count] = grid size
    int32_t count;
};

__global__ __launch_bounds__(256) void nan_to_num_multi_kernel(NanToNumTable tab, float nan, float posinf, float neginf) {
    int lo = 0, hi = tab.count;             // largest t with chunk0[t] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tab.chunk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    float* const base = tab.ptr[lo];
    const int n = tab.numel[lo];
    const int begin = ((int)blockIdx.x - tab.chunk0[lo]) * NTN_CHUNK;
    const int end = begin + NTN_CHUNK < n ? begin + NTN_CHUNK : n;
    auto fix = [&](float v) { return v != v ? nan : v == INFINITY ? posinf : v == -INFINITY ? neginf : v; };
    const int tid = threadIdx.x;
    if (((uintptr_t)base & 15) == 0) {      // chunks start at multiples of 16 K elements: float4 accesses, then <= 3 tail elements
        const int vend = begin + ((end - begin) & ~3);
        int i = begin + 4 * tid;
        for (; i + 3 * 1024 < vend; i += 4 * 1024) {       // four 16-byte loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = *(float4*)(base + i + 1024 * k);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k].x = fix(v[k].x); v[k].y = fix(v[k].y); v[k].z = fix(v[k].z); v[k].w = fix(v[k].w);
                *(float4*)(base + i + 1024 * k) = v[k];
            }
        }
        for (; i < vend; i += 1024) {
            float4 v = *(float4*)(base + i);
            v.x = fix(v.x); v.y = fix(v.y); v.z = fix(v.z); v.w = fix(v.w);
            *(float4*)(base + i) = v;
        }
        if (tid < end - vend) base[vend + tid] = fix(base[vend + tid]);
    } else {
        for (int i = begin + tid; i < end; i += 256) base[i] = fix(base[i]);
    }
}

extern "C" int pasta_spade_norm(const void* x, const void* gamma, const void* beta, void* out, float* stats, int dtype,
                                int64_t planes, int64_t HW, float eps, int act, float gain, float clamp, int C, int64_t gb_stride,
                                void* stream, float* y_amax) {
    using namespace pasta;
    PASTA_CHECK(x && out, "spade_norm: null pointer");
    PASTA_CHECK(act == 0 || act == 1 || act == 2, "spade_norm: fused activation code %d (0/1 = none, 2 = relu)", act);
    PASTA_CHECK(planes >= 1 && HW >= 1, "spade_norm: empty tensor");
    PASTA_CHECK(C >= 1 && planes % C == 0, "spade_norm: %lld planes are not whole samples of %d channels", (long long)planes, C);
    const int64_t gb_ns = gb_stride > 0 ? gb_stride : (int64_t)C * HW;
    PASTA_CHECK(gb_ns >= (int64_t)C * HW, "spade_norm: gamma / beta sample stride %lld below C * HW", (long long)gb_ns);
    int grid = (int)(planes < 65535 ? planes : 65535);
    hipStream_t s = (hipStream_t)stream;
    const bool al = (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) == 0 && gb_ns % 4 == 0;
#define PASTA_ARGS(T) (const T*)x, (const T*)gamma, (const T*)beta, (T*)out, stats, planes, HW, eps, act, gain, clamp, C, gb_ns, y_amax
#define PASTA_L(T)                                                                                                        \
    if (HW == 16384 && al) hipLaunchKernelGGL((spade_norm_kernel<T, 16>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T));  \
    else if (HW == 4096 && al) hipLaunchKernelGGL((spade_norm_kernel<T, 4>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T)); \
    else hipLaunchKernelGGL((spade_norm_kernel<T, 0>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T))
    PASTA_BY_DTYPE(dtype, "spade_norm", PASTA_L)
#undef PASTA_L
#undef PASTA_ARGS
    return launch_status("spade_norm");
}

extern "C" int pasta_spade_norm_bwd(const void* dout, const void* x, const void* gamma, const float* stats, void* dx,
                                    void* dgamma, void* dbeta, int dtype, int64_t planes, int64_t HW, const void* beta, int act, float gain,
                                    float clamp, int C, int64_t gb_stride, int64_t dgb_stride, void* stream, float* dx_amax, float* dgb_amax,
                                    const void* dx_add) {
    using namespace pasta;
    PASTA_CHECK(dout && x && stats, "spade_norm_bwd: null pointer");
    PASTA_CHECK(!dx_add || dx, "spade_norm_bwd: dx_add without dx");
    PASTA_CHECK(act == 0 || act == 1 || act == 2, "spade_norm_bwd: fused activation code %d (0/1 = none, 2 = relu)", act);
    PASTA_CHECK(act != 2 || dbeta || !dgamma, "spade_norm_bwd: dbeta buffer required with a fused activation");
    PASTA_CHECK(planes >= 1 && HW >= 1, "spade_norm_bwd: empty tensor");
    PASTA_CHECK(C >= 1 && planes % C == 0, "spade_norm_bwd: %lld planes are not whole samples of %d channels", (long long)planes, C);
    const int64_t gb_ns = gb_stride > 0 ? gb_stride : (int64_t)C * HW, dgb_ns = dgb_stride > 0 ? dgb_stride : (int64_t)C * HW;
    PASTA_CHECK(gb_ns >= (int64_t)C * HW && dgb_ns >= (int64_t)C * HW, "spade_norm_bwd: sample stride below C * HW");
    int grid = (int)(planes < 65535 ? planes : 65535);
    hipStream_t s = (hipStream_t)stream;
    const bool al = (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)beta | (uintptr_t)dx_add) & 15) == 0 &&
                    gb_ns % 4 == 0 && dgb_ns % 4 == 0;
#define PASTA_ARGS(T) (const T*)dout, (const T*)x, (const T*)gamma, stats, (T*)dx, (T*)dgamma, (T*)dbeta, planes, HW, (const T*)beta, act, gain, clamp, C, gb_ns, dgb_ns, dx_amax, dgb_amax, (const T*)dx_add
#define PASTA_L(T)                                                                                                            \
    if (HW == 16384 && al) hipLaunchKernelGGL((spade_norm_bwd_kernel<T, 16>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T));  \
    else if (HW == 4096 && al) hipLaunchKernelGGL((spade_norm_bwd_kernel<T, 4>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T)); \
    else hipLaunchKernelGGL((spade_norm_bwd_kernel<T, 0>), dim3(grid), dim3(1024), 0, s, PASTA_ARGS(T))
    PASTA_BY_DTYPE(dtype, "spade_norm_bwd", PASTA_L)
#undef PASTA_L
#undef PASTA_ARGS
    return launch_status("spade_norm_bwd");
}

extern "C" int pasta_masked_mean_fill(const float* a, const float* valid, const float* hole, const float* inv_count, float* out,
                                      int N, int C, int64_t HW, int64_t a_sample_stride, int64_t out_sample_stride, int backward,
                                      void* stream, float* y_amax) {
    using namespace pasta;
    PASTA_CHECK(a && valid && hole && inv_count && out, "masked_mean_fill: null pointer");
    PASTA_CHECK(N >= 1 && C >= 1 && HW >= 1, "masked_mean_fill: empty tensor");
    const int64_t a_ns = a_sample_stride > 0 ? a_sample_stride : (int64_t)C * HW, out_ns = out_sample_stride > 0 ? out_sample_stride : (int64_t)C * HW;
    PASTA_CHECK(a_ns >= (int64_t)C * HW && out_ns >= (int64_t)C * HW, "masked_mean_fill: sample stride below C * HW");
    const int64_t planes = (int64_t)N * C;
    const int grid = (int)(planes < 65535 ? planes : 65535);
    hipStream_t s = (hipStream_t)stream;
    const bool al = (((uintptr_t)a | (uintptr_t)valid | (uintptr_t)hole | (uintptr_t)out) & 15) == 0 && a_ns % 4 == 0 && out_ns % 4 == 0;
#define PASTA_MMF(E, B) hipLaunchKernelGGL((masked_mean_fill_kernel<E, B>), dim3(grid), dim3(1024), 0, s, a, valid, hole, inv_count, out, planes, C, HW, a_ns, out_ns, y_amax)
    if (HW == 16384 && al) { if (backward) PASTA_MMF(16, true); else PASTA_MMF(16, false); }
    else if (HW == 4096 && al) { if (backward) PASTA_MMF(4, true); else PASTA_MMF(4, false); }
    else { if (backward) PASTA_MMF(0, true); else PASTA_MMF(0, false); }
#undef PASTA_MMF
    return launch_status("masked_mean_fill");
}

extern "C" int pasta_mod_bias_act(const void* u, const float* d, const float* noise, const float* strength, const float* b, void* y,
                                  int dtype, int N, int C, int64_t HW, int noise_per_sample, int act, float alpha, float gain, float clamp,
                                  void* stream, float* y_amax) {
    using namespace pasta;
    PASTA_CHECK(u && y, "mod_bias_act: null pointer");
    PASTA_CHECK(N >= 1 && C >= 1 && HW >= 1, "mod_bias_act: empty tensor");
    PASTA_CHECK(act == 1 || act == 3, "mod_bias_act: activation code %d (linear = 1 and lrelu = 3 are supported)", act);
    PASTA_CHECK(!noise || strength, "mod_bias_act: noise without strength");
    PASTA_CHECK(((HW & 3) != 0) || ((((uintptr_t)u | (uintptr_t)y | (uintptr_t)noise) & 15) == 0), "mod_bias_act: pointers must be 16-byte aligned");
    const int64_t chunks = (HW + MBA_CHUNK - 1) / MBA_CHUNK;
    PASTA_CHECK(chunks <= 65535 && (int64_t)N * C <= INT32_MAX, "mod_bias_act: tensor too large");
#define PASTA_L(T) hipLaunchKernelGGL(mod_bias_act_kernel<T>, dim3((unsigned)(N * C), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, (const T*)u, d, noise, \
                                      strength, b, (T*)y, C, HW, noise_per_sample, act, alpha, gain, clamp, y_amax)
    PASTA_BY_DTYPE(dtype, "mod_bias_act", PASTA_L)
#undef PASTA_L
    return launch_status("mod_bias_act");
}

extern "C" int64_t pasta_mod_bias_act_bwd_workspace(int N, int C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return (int64_t)N * C * ((HW + pasta::MBA_CHUNK - 1) / pasta::MBA_CHUNK) * 3 * (int64_t)sizeof(float);
}

extern "C" int pasta_mod_bias_act_bwd(const void* dy, const void* y, const void* u, const float* d, const float* noise, void* du,
                                      float* partial, int dtype, int N, int C, int64_t HW, int noise_per_sample, int act, float alpha, float gain,
                                      float clamp, void* stream, float* du_amax) {
    using namespace pasta;
    PASTA_CHECK(dy && y && u && du && partial, "mod_bias_act_bwd: null pointer");
    PASTA_CHECK(N >= 1 && C >= 1 && HW >= 1, "mod_bias_act_bwd: empty tensor");
    PASTA_CHECK(act == 1 || act == 3, "mod_bias_act_bwd: activation code %d (linear = 1 and lrelu = 3 are supported)", act);
    PASTA_CHECK(((HW & 3) != 0) || ((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)u | (uintptr_t)du | (uintptr_t)noise) & 15) == 0),
                "mod_bias_act_bwd: pointers must be 16-byte aligned");
    const int64_t chunks = (HW + MBA_CHUNK - 1) / MBA_CHUNK;
    PASTA_CHECK(chunks <= 65535 && (int64_t)N * C <= INT32_MAX, "mod_bias_act_bwd: tensor too large");
#define PASTA_L(T) hipLaunchKernelGGL(mod_bias_act_bwd_kernel<T>, dim3((unsigned)(N * C), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)y, \
                                      (const T*)u, d, noise, (T*)du, partial, C, HW, (int)chunks, noise_per_sample, act, alpha, gain, clamp, du_amax)
    PASTA_BY_DTYPE(dtype, "mod_bias_act_bwd", PASTA_L)
#undef PASTA_L
    return launch_status("mod_bias_act_bwd");
}

extern "C" int pasta_nan_to_num_multi(float* const* ptrs, const int64_t* numels, int n, float nan, float posinf, float neginf,
                                      void* stream) {
    using namespace pasta;
    PASTA_CHECK(n == 0 || (ptrs && numels), "nan_to_num_multi: null table");
    int done = 0;
    while (done < n) {
        NanToNumTable tab;
        int cnt = 0, chunks = 0;
        while (done < n && cnt < NTN_MAX) {
            const int64_t ne = numels[done];
            PASTA_CHECK(ne >= 0 && ne < (1ll << 31), "nan_to_num_multi: tensor %d has %lld elements", done, (long long)ne);
            if (ne > 0) {
                PASTA_CHECK(ptrs[done], "nan_to_num_multi: tensor %d is null", done);
                tab.ptr[cnt] = ptrs[done]; tab.numel[cnt] = (int32_t)ne; tab.chunk0[cnt] = chunks;
                chunks += (int)ceil_div64(ne, NTN_CHUNK);
                cnt++;
            }
            done++;
        }
        if (cnt == 0) break;
        tab.chunk0[cnt] = chunks; tab.count = cnt;
        hipLaunchKernelGGL(nan_to_num_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, tab, nan, posinf, neginf);
    }
    return launch_status("nan_to_num_multi");
}
