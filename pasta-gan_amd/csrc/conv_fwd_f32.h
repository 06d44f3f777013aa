// fp32-MFMA forward-type kernels (v_mfma_f32_32x32x2_f32): weight packing, conv_fwd_kernel and its launcher.  Instantiated by conv_tu_pack_f32.hip.
#pragma once
#include "conv_common.h"

namespace pasta {

//------------------------------------------------------------------------------------
// Weight packing: PyTorch layout -> [G][kh*kw][I_pad][O_pad] (O contiguous), zero padded so
// the GEMM's A-operand staging needs no bounds checks.

#ifdef PASTA_TU_PACK       // defined by conv_tu_pack_f32.hip only (conv_launch.h)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int G, int Ig,
                                                           int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                                                           int flip, float wscale, const float* __restrict__ mod_s,
                                                           const float* __restrict__ mod_d) {
    const int64_t total = (int64_t)G * kh * kw * Ig_pad * Og_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int o = (int)(idx % Og_pad);
        int64_t r = idx / Og_pad;
        int i = (int)(r % Ig_pad); r /= Ig_pad;
        int t = (int)(r % (kh * kw));
        int g = (int)(r / (kh * kw));
        float v = 0.f;
        if (i < Ig && o < Og) {
            int ty = t / kw, tx = t - ty * kw;
            if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
            // mod_s: ONE weight of a single group, shared by all groups, modulated per group (= sample) on the way:
            // w[o,i] * s[g,i] (* d[g,o]) rounded in this order, as networks.py:65-68, 84-86 forms its per-sample weights
            const int gs = mod_s ? 0 : g;
            const int64_t src = transposed ? (((int64_t)(gs * Ig + i) * Og + o) * kh + ty) * kw + tx
                                           : (((int64_t)(gs * Og + o) * Ig + i) * kh + ty) * kw + tx;
            v = w[src] * wscale;
            if (mod_s) { v *= mod_s[(int64_t)g * Ig + i]; if (mod_d) v *= mod_d[(int64_t)g * Og + o]; }
        }
        wp[idx] = v;
    }
}
#endif  // PASTA_TU_PACK


template <int BM, int BN, int WMT, int WNT, int KC, int OCC = 1>   // OCC = minimum waves per SIMD asked of the register allocator
__global__ __launch_bounds__(256, OCC) void conv_fwd_kernel(ConvFwdParams p) {
    constexpr int WAVES_N = BN / (32 * WNT);
    static_assert((BM / (32 * WMT)) * WAVES_N == 4, "four waves per workgroup");
    constexpr int RSTEP = 256 / BN > 0 ? 256 / BN : 1;      // k-rows covered by one pass of the workgroup (B tile)
    constexpr int BPT = KC * BN / 256;                       // B elements per thread per chunk
    constexpr int A4_PER_ROW = BM / 4;
    constexpr int APT = (KC * A4_PER_ROW + 255) / 256;       // float4 A loads per thread per chunk
    static_assert(BN <= 256 && BPT >= 1, "tile/thread mapping");

    __shared__ float As[2][KC][BM];
    __shared__ float Bs[2][KC][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    const int64_t npix = (int64_t)p.N * p.P * p.Q;
    const int64_t pix_blk = (int64_t)blockIdx.x * BN;
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int chunks_all = p.T * NC;
    const int c_first = (int)(((unsigned)chunks_all * (unsigned)ks) / (unsigned)p.ksplit);
    const int nchunks = (int)(((unsigned)chunks_all * (unsigned)(ks + 1)) / (unsigned)p.ksplit) - c_first;

    // ---- B staging: this thread's pixel column is fixed for the whole K loop.
    const int bcol = tid % BN, brow0 = tid / BN;
    const int64_t mypix = pix_blk + bcol;
    const bool pix_ok = mypix < npix;
    int n_in = 0, py = 0, px = 0;
    if (pix_ok) {
        n_in = (int)(mypix / (p.P * p.Q));
        int rem = (int)(mypix - (int64_t)n_in * p.P * p.Q);
        py = rem / p.Q; px = rem - py * p.Q;
    }
    const float* xb = p.x + ((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW;
    const float* isb = p.iscale ? p.iscale + (int64_t)n_in * p.Cin + (int64_t)g * p.Ig : nullptr;
    const int iy_base = py * p.isy, ix_base = px * p.isx;

    // ---- A staging.
    const float* wb = p.wp + (int64_t)g * p.KK * p.Ig_pad * p.Og_pad + o_blk;

    float  breg[BPT];
    float4 areg[APT];
    unsigned bmask = 0;         // bit j: breg[j] is a real element (inside the image, channel < Ig)

    // Loader state: tap index and channel offset of the NEXT chunk to fetch, plus the per-tap
    // quantities derived from them (recomputed only when the tap changes: T times, not per chunk).
    int ld_t = c_first / NC, ld_c0 = (c_first - ld_t * NC) * KC;
    bool ld_ok = false;
    const float* ld_xp = xb;
    const float* ld_wt = wb;
    auto set_tap = [&](int t) {
        const int iy = iy_base + p.tap_dy[t], ix = ix_base + p.tap_dx[t];
        ld_ok = pix_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        ld_xp = ld_ok ? xb + iy * p.W + ix : xb;       // always readable; out-of-image taps are zeroed by the select below
        ld_wt = wb + (int64_t)p.tap_slab[t] * p.Ig_pad * p.Og_pad;
    };
    if (ld_t < p.T) set_tap(ld_t);
    // Unconditional loads from a clamped channel index: no branch and no wait sits between a load and the MFMAs that
    // cover its latency.
    auto load_chunk = [&]() {
        const int last = p.Ig - 1;
        bmask = 0;
#pragma unroll
        for (int j = 0; j < BPT; j++) {
            const int c = ld_c0 + brow0 + j * RSTEP;
            if (ld_ok && c < p.Ig) bmask |= 1u << j;         // validity is known now; the select waits until the store
        }
        if (isb) {
#pragma unroll
            for (int j = 0; j < BPT; j++) {
                const int c = ld_c0 + brow0 + j * RSTEP, cs = c < last ? c : last;
                breg[j] = ld_xp[(int64_t)cs * HW] * isb[cs];
            }
        } else {
#pragma unroll
            for (int j = 0; j < BPT; j++) {
                const int c = ld_c0 + brow0 + j * RSTEP, cs = c < last ? c : last;
                breg[j] = ld_xp[(int64_t)cs * HW];
            }
        }
        const float* wt = ld_wt + (int64_t)ld_c0 * p.Og_pad;
#pragma unroll
        for (int j = 0; j < APT; j++) {
            const int e = tid + j * 256, row = e / A4_PER_ROW, c4 = e - row * A4_PER_ROW;
            if (row < KC) areg[j] = *(const float4*)(wt + (int64_t)row * p.Og_pad + c4 * 4);
        }
        ld_c0 += KC;
        if (ld_c0 >= p.Ig_pad) {
            ld_c0 = 0;
            if (++ld_t < p.T) set_tap(ld_t);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BPT; j++) Bs[buf][brow0 + j * RSTEP][bcol] = (bmask >> j & 1u) ? breg[j] : 0.f;
#pragma unroll
        for (int j = 0; j < APT; j++) {
            const int e = tid + j * 256, row = e / A4_PER_ROW, c4 = e - row * A4_PER_ROW;
            if (row < KC) *(float4*)&As[buf][row][c4 * 4] = areg[j];
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    if (nchunks > 0) {          // an empty K slice (more slices than chunks) contributes zeros
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    const int kl = lane >> 5, jl = lane & 31;
    for (int ch = 0; ch < nchunks; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk();
#pragma unroll
        for (int kk = 0; kk < KC / 2; kk++) {
            float af[WMT], bf[WNT];
#pragma unroll
            for (int a = 0; a < WMT; a++) af[a] = As[buf][kk * 2 + kl][(wm * WMT + a) * 32 + jl];
#pragma unroll
            for (int b = 0; b < WNT; b++) bf[b] = Bs[buf][kk * 2 + kl][(wn * WNT + b) * 32 + jl];
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int b = 0; b < WNT; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: registers -> NCHW, rows = channels, lanes = consecutive pixels.
    const int OHW = p.OH * p.OW;
    const bool has_noise = p.noise && p.ksplit == 1;
    const float nstr = has_noise ? p.noise_strength[0] : 0.f;
    // the stores: instantiated per (activation, clamp, whole tile of rows) and chosen once per workgroup (conv_common.h: a store loop with
    // three or four wave-uniform branches per element cost the split kernels 5 %)
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, p.ksplit == 1);
    conv_epilogue_dispatch<true>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
    const bool FULL = full_c;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        if (pix >= npix) continue;
        const int n = (int)(pix / (p.P * p.Q));
        const int rem = (int)(pix - (int64_t)n * p.P * p.Q);
        const int pp = rem / p.Q, qq = rem - pp * p.Q;
        const int plane_off = (p.oy0 + pp * p.osy) * p.OW + p.ox0 + qq * p.osx;
        const int64_t yoff = ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
        const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n * OHW : 0) + plane_off] * nstr : 0.f;
        float* yb = (p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW : p.y) + yoff;
        const float* rb = (p.res && p.ksplit == 1) ? p.res + yoff : nullptr;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
        const float* bsb = (ea.on && p.bias) ? p.bias + g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++) {
            float tv[16];                               // output scale, residual, bias through the same registers: sixteen loads in a row, then their use
            if (osb) {
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = osb[(FULL || o < p.Og) ? o : p.Og - 1]; }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = fmaf(acc[a][b][r], tv[r], nz);
            } else if (has_noise) {
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += nz;
            }
            if (rb) {
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = (FULL || o < p.Og) ? rb[(int64_t)o * OHW] : 0.f; }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += tv[r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = bsb ? bsb[(FULL || o < p.Og) ? o : p.Og - 1] : 0.f; }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                const float v = conv_epilogue_c(acc[a][b][r], tv[r], ea, case_c);
                if (FULL || o < p.Og) yb[(int64_t)o * OHW] = v;
            }
        }
    }
    });
}

template <int BM, int BN, int WMT, int WNT, int KC, int OCC = 1>
static void launch_fwd(const ConvFwdParams& p, hipStream_t s) {
    const int64_t npix = (int64_t)p.N * p.P * p.Q;
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    dim3 grid((unsigned)ceil_div64(npix, BN), q.o_tiles * q.ksplit, p.G);
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WMT, WNT, KC, OCC>), grid, dim3(256), 0, s, q);
}



}  // namespace pasta
