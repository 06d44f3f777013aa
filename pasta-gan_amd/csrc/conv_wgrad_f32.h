// fp32-MFMA weight-gradient kernels: conv_wgrad_kernel, the small-Cin variant, their slab reductions and the small-Cin plan.  Instantiated by conv_tu_wgrad_f32.hip (conv_launch.h).
#pragma once
#include "conv_common.h"

namespace pasta {

//------------------------------------------------------------------------------------
// Weight gradient.
//   dW[g*Ag + a][b][r][s] = sum_{n,p,q} S[n, g*Ag + a, p, q] * L[n, g*Bg + b, p*st + r - pad_h, q*st + s - pad_w]
// Workgroup: 64 (a) x 64 (b) x TR*TS taps, over a slice of K = pixels.  K is walked in chunks of
// CHH x CW = 32 lattice pixels (CW a power of two <= 32 chosen from Q).


// TR x TS taps per workgroup; each wave owns WA x WB 32x32 tiles per tap; PIPE = prefetch the next chunk into
// registers behind the MFMAs (needs a halo of at most 128 positions)
template <int TR, int TS, int WA, int WB, int PIPE, int KP>   // PIPE: 0 = none, 1 = halo <= 128 positions; KP pixels per chunk
__global__ __launch_bounds__(256, PIPE ? 2 : 1) void conv_wgrad_kernel(WgradParams p) {
    constexpr int NT = TR * TS;
    constexpr int BA = 64 * WA, BB = 64 * WB;  // workgroup tile: 2 x 2 waves
    constexpr int SPITCH = KP + 1;             // odd pitch: column-of-channels reads hit 32 banks
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int CW = 1 << p.cw_log2, CHH = KP >> p.cw_log2;
    const int LWID = (CW - 1) * p.st + TS;     // halo width per chunk row
    const int LPITCH = LWID | 1;
    const int LROWS = CHH * TR;                // halo rows per channel: one set of TR rows per chunk row
    const int LCH = (LROWS * LPITCH) | 1;      // odd per-channel pitch
    float* Ss = smem;                          // [BA][SPITCH]
    float* Ls = smem + BA * SPITCH;            // [BB][LCH]

    // block coordinates
    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int tgs = bid % p.tap_groups_s; bid /= p.tap_groups_s;
    const int tgr = bid % p.tap_groups_r; bid /= p.tap_groups_r;
    const int bt = bid % p.b_tiles; bid /= p.b_tiles;
    const int at = bid % p.a_tiles; bid /= p.a_tiles;
    const int g = bid;
    const int r0 = tgr * TR, s0 = tgs * TS;
    const int a_blk = at * BA, b_blk = bt * BB;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;

    // ---- staging roles, fixed for the whole K loop.
    // S: this thread's pixel of the chunk and every 8th channel.
    constexpr int SROWS = 256 / KP;                        // channels covered by one pass of the workgroup
    constexpr int SPT = BA / SROWS;
    const int s_k = tid & (KP - 1), s_a0 = tid / KP;
    const int s_dr = s_k >> p.cw_log2, s_dq = s_k & (CW - 1);
    // L: one halo position and every `lgroups`-th channel (the host guarantees NPOS <= 256).
    const int NPOS = LROWS * LWID;
    int npos_pad = 64;
    while (npos_pad < NPOS) npos_pad <<= 1;
    const int lgroups = 256 / npos_pad;                    // 4, 2 or 1 channel groups
    const int l_pos = tid & (npos_pad - 1), l_cg = tid / npos_pad;
    const bool l_act = l_pos < NPOS;
    const int l_lr = l_act ? l_pos / LWID : 0, l_lc = l_pos - l_lr * LWID;
    const int l_cr = l_lr / TR, l_tr = l_lr - l_cr * TR;
    float* const l_dst = Ls + l_lr * LPITCH + l_lc;
    const bool l_tap_ok = l_act && r0 + l_tr < p.kh;
    constexpr int LPT = PIPE ? BB / 2 : 1;                 // prefetch registers per thread (PIPE needs lgroups >= 2)

    f32x16 acc[NT][WA][WB];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < WA; i++)
#pragma unroll
            for (int j = 0; j < WB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][i][j][r] = 0.f;

    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);
    const int kl = lane >> 5, jl = lane & 31;
    const float* const Sg = p.S + ((int64_t)g * p.Ag + a_blk) * PQ;
    const float* const Lg = p.L + ((int64_t)g * p.Bg + b_blk) * LHW;

    float sreg[SPT];
    float lreg[LPT];

    // Per-chunk source offsets of this thread's S pixel and L halo position (element offsets fit in 31 bits).
    auto s_source = [&](int ch, bool& ok) -> unsigned {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row = rb * CHH + s_dr, q = qb * CW + s_dq;
        ok = row < p.rows_total && q < p.Q;
        const int n = ok ? row / p.P : 0, pp = row - n * p.P;
        return (unsigned)(n * p.SC) * (unsigned)PQ + (unsigned)(pp * p.Q + q);
    };
    auto l_source = [&](int ch, bool& ok) -> unsigned {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row = rb * CHH + l_cr;
        const bool rok = l_tap_ok && row < p.rows_total;
        const int n = rok ? row / p.P : 0, pp = row - n * p.P;
        const int ly = pp * p.st + r0 + l_tr - p.pad_h, lx = qb * CW * p.st + s0 + l_lc - p.pad_w;
        ok = rok && (unsigned)ly < (unsigned)p.LH && (unsigned)lx < (unsigned)p.LW;
        return ok ? (unsigned)(n * p.LC) * (unsigned)LHW + (unsigned)(ly * p.LW + lx) : 0u;
    };

    const bool full_a = a_blk + BA <= p.Ag, full_b = b_blk + BB <= p.Bg;   // uniform: no per-channel bound checks
    auto fetch = [&](int ch) {            // global -> registers (PIPE only)
        bool ok;
        const unsigned so = s_source(ch, ok);
        {
            const float* sp = Sg + so + (unsigned)s_a0 * (unsigned)PQ;
            const unsigned step = (unsigned)SROWS * (unsigned)PQ;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                sreg[j] = (ok && (full_a || a_blk + s_a0 + SROWS * j < p.Ag)) ? *sp : 0.f;
                sp += step;
            }
        }
        const unsigned lo = l_source(ch, ok);
        {
            const float* lp = Lg + lo + (unsigned)l_cg * (unsigned)LHW;
            const unsigned step = (unsigned)lgroups * (unsigned)LHW;
#pragma unroll
            for (int j = 0; j < LPT; j++) {
                const int b = l_cg + lgroups * j;
                lreg[j] = (ok && b < BB && (full_b || b_blk + b < p.Bg)) ? *lp : 0.f;
                lp += step;
            }
        }
    };
    auto stash = [&]() {                  // registers -> LDS (PIPE only)
#pragma unroll
        for (int j = 0; j < SPT; j++) Ss[(s_a0 + SROWS * j) * SPITCH + s_k] = sreg[j];
        if (l_act) {
#pragma unroll
            for (int j = 0; j < LPT; j++) {
                const int b = l_cg + lgroups * j;
                if (b < BB) l_dst[b * LCH] = lreg[j];
            }
        }
    };
    auto stage_direct = [&](int ch) {     // global -> LDS without the register stage (!PIPE)
        bool ok;
        const unsigned so = s_source(ch, ok);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int a = s_a0 + SROWS * j;
            Ss[a * SPITCH + s_k] = (ok && a_blk + a < p.Ag) ? Sg[so + (unsigned)a * (unsigned)PQ] : 0.f;
        }
        const unsigned lo = l_source(ch, ok);
        if (l_act) {
#pragma unroll 8
            for (int b = l_cg; b < BB; b += lgroups)
                l_dst[b * LCH] = (ok && b_blk + b < p.Bg) ? Lg[lo + (unsigned)b * (unsigned)LHW] : 0.f;
        }
    };

    if (PIPE && c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        if (PIPE) stash(); else stage_direct(ch);
        __syncthreads();
        if (PIPE && ch + 1 < c_end) fetch(ch + 1);   // in flight behind the MFMAs below
        // ---- KP/2 k-steps of 2 pixels; per step WA A-fragments feed NT*WA*WB MFMAs
#pragma unroll 2
        for (int kk = 0; kk < KP / 2; kk++) {
            const int k = kk * 2 + kl;
            float af[WA];
#pragma unroll
            for (int i = 0; i < WA; i++) af[i] = Ss[((wa * WA + i) * 32 + jl) * SPITCH + k];
            const int cr = k >> p.cw_log2, cc = k & (CW - 1);
            const float* lb = Ls + cr * TR * LPITCH + cc * p.st;
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const float* lbj = lb + ((wb * WB + j) * 32 + jl) * LCH;
#pragma unroll
                for (int tr = 0; tr < TR; tr++)
#pragma unroll
                    for (int ts = 0; ts < TS; ts++) {
                        const float bf = lbj[tr * LPITCH + ts];
#pragma unroll
                        for (int i = 0; i < WA; i++)
                            acc[tr * TS + ts][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf, acc[tr * TS + ts][i][j], 0, 0, 0);
                    }
            }
        }
    }

    // ---- partial slab: [ksplit][G][kh*kw][Ag_pad][Bg_pad], b contiguous
    const int Ag_pad = p.a_tiles * BA, Bg_pad = p.b_tiles * BB;
    float* out = p.slab + ((int64_t)ks * p.G + g) * p.kh * p.kw * Ag_pad * Bg_pad;
#pragma unroll
    for (int tr = 0; tr < TR; tr++)
#pragma unroll
        for (int ts = 0; ts < TS; ts++) {
            if (r0 + tr >= p.kh || s0 + ts >= p.kw) continue;
            float* ot = out + (int64_t)((r0 + tr) * p.kw + s0 + ts) * Ag_pad * Bg_pad;
#pragma unroll
            for (int i = 0; i < WA; i++)
#pragma unroll
                for (int j = 0; j < WB; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int a = a_blk + (wa * WA + i) * 32 + acc_row(r, lane), b = b_blk + (wb * WB + j) * 32 + jl;
                        ot[(int64_t)a * Bg_pad + b] = acc[tr * TS + ts][i][j][r];
                    }
        }
}

// dW[(g*Ag + a)][b][ty][tx] = sum_ks slab[ks][g][t][a][b]   (tap index optionally mirrored)
#ifdef PASTA_TU_WGRAD_F32  // defined by conv_tu_wgrad_f32.hip only (conv_launch.h)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ksplit,
                                                           int G, int Ag, int Bg, int Ag_pad, int Bg_pad, int kh, int kw,
                                                           int flip, float wscale) {
    const int KK = kh * kw;
    const int64_t total = (int64_t)G * KK * Ag * Bg;
    const int64_t slab_stride = (int64_t)G * KK * Ag_pad * Bg_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int b = (int)(idx % Bg);
        int64_t r = idx / Bg;
        const int a = (int)(r % Ag); r /= Ag;
        const int t = (int)(r % KK);
        const int g = (int)(r / KK);
        const float* src = slab + (((int64_t)g * KK + t) * Ag_pad + a) * Bg_pad + b;
        // sixteen slabs in flight per thread; four partial sums combined in a fixed order (bitwise reproducible)
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int k = 0;
        for (; k + 16 <= ksplit; k += 16) {
            float r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = src[(int64_t)(k + j) * slab_stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { v0 += r[j]; v1 += r[j + 1]; v2 += r[j + 2]; v3 += r[j + 3]; }
        }
        for (; k < ksplit; k++) v0 += src[(int64_t)k * slab_stride];
        const float v = ((v0 + v1) + (v2 + v3)) * wscale;
        int ty = t / kw, tx = t - ty * kw;
        if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
        dw[(((int64_t)(g * Ag + a) * Bg + b) * kh + ty) * kw + tx] = v;
    }
}
#endif  // PASTA_TU_WGRAD_F32

//------------------------------------------------------------------------------------
// Weight gradient AND style gradient of a modulated convolution y = conv(x * s[n, i], w) (networks.py:72-76) from the weight-gradient kernels run on
// the UNMODULATED x (pasta_conv2d_wgrad_modulated): the K slices are sample-aligned -- slices [n m, (n + 1) m), m = ksplit / N, sum to Dw_n, sample n's
// gradient with respect to the weight it saw, w s[n, i] -- so
//     dw[o, i, t] = wscale sum_n s[n, i] Dw_n[o, i, t]            ds[n, i] = sum_{o, t} wscale w[o, i, t] Dw_n[o, i, t]
// and neither x * s (a pass over the activation to form it, one to scan it, one to scale the input gradient back) nor the plane products
// sum_hw dx x (two reads) exist.  Workgroup = (64 b, 16 or 4 a, one tap); wave = four rows a or one, lanes along b (the slab's contiguous index).
// MOD_A: the modulated (input) channel is the a index (conv_transpose2d: weight [I, O, kh, kw]).  Partial ds per workgroup row:
// dsp[(a block * KK + t)][n][b] (MOD_A: dsp[(b tile * KK + t)][n][a]), summed in a fixed order by sum_blocks_kernel.  N <= 32.
// rows a per workgroup: 16, or 4 where 16 would leave the grid below 512 workgroups (the 64- and 128-channel layers: 75 MB of slabs each)
static int wgrad_mod_rows(int Ap, int Bp, int KK) { return (int64_t)(Bp / 64) * (Ap / 16) * KK >= 512 ? 16 : 4; }
template <bool MOD_A>
__global__ __launch_bounds__(256) void wgrad_reduce_modulated_kernel(const float* __restrict__ slab, const float* __restrict__ sty, const float* __restrict__ w,
                                                                     float* __restrict__ dw, float* __restrict__ dsp, int ksplit, int N,
                                                                     int Ag, int Bg, int Ap, int Bp, int kh, int kw, int flip, float wscale, int wg_rows) {
    constexpr int MAXN = 32;
    const int ROWS = wg_rows / 4;                                    // rows a per wave
    const int KK = kh * kw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int t = blockIdx.z;
    const int m = ksplit / N;
    const int C = MOD_A ? Ag : Bg;                                   // modulated channels
    const int64_t slab_stride = (int64_t)KK * Ap * Bp;
    int ty = t / kw, tx = t - ty * kw;
    if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
    float dsacc[MAXN];
#pragma unroll
    for (int n = 0; n < MAXN; n++) dsacc[n] = 0.f;
    __shared__ float red[4][MAXN][64];
    for (int j = 0; j < ROWS; j++) {
        const int a = blockIdx.y * wg_rows + wave * ROWS + j;
        const bool live = a < Ag && b < Bg;
        const float* src = slab + ((int64_t)t * Ap + a) * Bp + b;   // a < Ap, b < Bp: inside the padded slab
        const int64_t widx = (((int64_t)a * Bg + b) * kh + ty) * kw + tx;
        const float wv = live ? w[widx] * wscale : 0.f;
        float dwsum = 0.f;
        // four samples at a time, their slices four at a time: sixteen independent loads in flight, every sum in a fixed order
#pragma unroll
        for (int n0 = 0; n0 < MAXN; n0 += 4) {
            if (n0 < N) {
                float pn[4] = {0.f, 0.f, 0.f, 0.f};
                const float* sp[4];
#pragma unroll
                for (int q = 0; q < 4; q++) sp[q] = src + (int64_t)((n0 + q < N ? n0 + q : n0) * m) * slab_stride;
                int k = 0;
                for (; k + 4 <= m; k += 4) {
                    float r[4][4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int u = 0; u < 4; u++) r[q][u] = sp[q][(int64_t)(k + u) * slab_stride];
#pragma unroll
                    for (int q = 0; q < 4; q++) pn[q] += (r[q][0] + r[q][1]) + (r[q][2] + r[q][3]);
                }
                for (; k < m; k++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) pn[q] += sp[q][(int64_t)k * slab_stride];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = n0 + q;
                    const float v = (live && n < N) ? pn[q] : 0.f;
                    const float sc = (live && n < N) ? sty[n * C + (MOD_A ? a : b)] : 0.f;
                    dwsum = fmaf(sc, v, dwsum);
                    if constexpr (MOD_A) {
                        float z = wv * v;                             // this row's share of ds[n, a]: summed over the b lanes now
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) z += __shfl_xor(z, off, 64);
                        if (lane == 0 && a < Ag && n < N) dsp[((int64_t)(blockIdx.x * KK + t) * N + n) * C + a] = z;
                    } else {
                        dsacc[n] = fmaf(wv, v, dsacc[n]);
                    }
                }
            }
        }
        if (live) dw[widx] = dwsum * wscale;
    }
    if constexpr (!MOD_A) {
#pragma unroll
        for (int n = 0; n < MAXN; n++) if (n < N) red[wave][n][lane] = dsacc[n];
        __syncthreads();
        // four waves' partial sums in a fixed order; thread -> (n, b): 64 lanes along b, waves stride n
        for (int n = wave; n < N; n += 4) {
            const float v = (red[0][n][lane] + red[1][n][lane]) + (red[2][n][lane] + red[3][n][lane]);
            if (b < Bg) dsp[((int64_t)(blockIdx.y * KK + t) * N + n) * C + b] = v;
        }
    }
}

// out[i] = sum_k blocks[k][i] in the order of k (bitwise reproducible)
#ifdef PASTA_TU_WGRAD_F32  // defined by conv_tu_wgrad_f32.hip only (conv_launch.h)
__global__ __launch_bounds__(256) void sum_blocks_kernel(const float* __restrict__ blocks, float* __restrict__ out, int nblocks, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int k = 0;
    for (; k + 4 <= nblocks; k += 4) {
        v0 += blocks[(int64_t)k * n + i]; v1 += blocks[(int64_t)(k + 1) * n + i]; v2 += blocks[(int64_t)(k + 2) * n + i]; v3 += blocks[(int64_t)(k + 3) * n + i];
    }
    for (; k < nblocks; k++) v0 += blocks[(int64_t)k * n + i];
    out[i] = (v0 + v1) + (v2 + v3);
}
#endif  // PASTA_TU_WGRAD_F32

//------------------------------------------------------------------------------------
// Weight gradient when the input has very few channels (RGB / pose stems: 3 or 6 channels, up to 7x7):
// the (channel, tap) pairs become the GEMM's column index b' = (i*kh + r)*kw + s, so a 7x7x3 kernel fills
// 147 of 160 MFMA columns instead of 3 of 64.  conv2d, stride 1, groups 1 only.
//   dW[o][b'] = sum_{n,p,q} dy[n,o,p,q] * x[n, i, p + r - pad_h, q + s - pad_w]

struct WgradSmallParams {
    const float* S; const float* L; float* slab;
    int N, Ag, P, Q;        // S = dy: [N, Ag, P, Q]
    int Bg, LH, LW;         // L = x : [N, Bg, LH, LW]
    int kh, kw, pad_h, pad_w;
    int bprime, nb;         // Bg*kh*kw and its number of 32-column tiles (<= 5)
    int cw_log2, rows_total, qblocks, chunks_total, ksplit, a_tiles;
};

#ifdef PASTA_TU_WGRAD_F32  // defined by conv_tu_wgrad_f32.hip only (conv_launch.h)
__global__ __launch_bounds__(256) void conv_wgrad_smallcin_kernel(WgradSmallParams p) {
    constexpr int KP = 32, SPITCH = KP + 1, MAXT = 3;     // each wave owns column tiles wb, wb+2, wb+4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int CW = 1 << p.cw_log2, CHH = KP >> p.cw_log2;
    const int HW_ = CW + p.kw - 1;           // halo width
    const int RH = CHH * p.kh;               // halo rows per channel: kh rows for each chunk row (rows may straddle images)
    float* Ss = smem;                        // [64][SPITCH]
    float* Ls = smem + 64 * SPITCH;          // [Bg][RH][HW_], then one zero word
    const int halo_elems = p.Bg * RH * HW_;

    int bid = blockIdx.x;
    const int ks = bid % p.ksplit; bid /= p.ksplit;
    const int a_blk = bid * 64;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;
    const int kl = lane >> 5, jl = lane & 31;

    // this lane's (channel, tap) column in each of its tiles -> LDS offset of its halo element for chunk pixel (0,0);
    // columns beyond Bg*kh*kw read the zero word
    int boff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int bp = (wb + 2 * t) * 32 + jl;
        if (bp < p.bprime) {
            const int i = bp / (p.kh * p.kw), rs = bp - i * p.kh * p.kw, r = rs / p.kw, sx = rs - r * p.kw;
            boff[t] = (i * RH + r) * HW_ + sx;
        } else boff[t] = -1;
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int s_k = tid & 31, s_a0 = tid >> 5;
    const int s_dr = s_k >> p.cw_log2, s_dq = s_k & (CW - 1);
    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);

    // Register prefetch: chunk ch+1 is fetched while chunk ch is multiplied out of LDS.  Every thread owns the same 8 S
    // elements (channel s_a0 + 8j, pixel s_k) and up to 4 halo slots of every chunk; the slot -> (channel, kernel row,
    // chunk row, halo column) decomposition is fixed, only the chunk origin moves.
    constexpr int HSLOTS = 4;                          // halo_elems <= 1024 (checked by the host)
    int h_off[HSLOTS], h_r[HSLOTS], h_cr[HSLOTS], h_hx[HSLOTS], h_i[HSLOTS];
#pragma unroll
    for (int j = 0; j < HSLOTS; j++) {
        int rem = tid + 256 * j;
        h_off[j] = rem < halo_elems ? rem : -1;
        h_hx[j] = rem % HW_; rem /= HW_;
        h_r[j] = rem % p.kh; rem /= p.kh;
        h_cr[j] = rem % CHH; h_i[j] = rem / CHH;
    }
    float sreg[8], hreg[HSLOTS];
    auto fetch = [&](int ch) {
        const int rb = ch / p.qblocks, qb = ch - rb * p.qblocks;
        const int row0 = rb * CHH, q0 = qb * CW;
        {   // S: 64 channels x 32 pixels
            const int row = row0 + s_dr, q = q0 + s_dq;
            const bool ok = row < p.rows_total && q < p.Q;
            const int n = ok ? row / p.P : 0, pp = row - n * p.P;
            const float* sp = p.S + ((int64_t)n * p.Ag + a_blk) * PQ + pp * p.Q + q;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int a = s_a0 + 8 * j;
                sreg[j] = (ok && a_blk + a < p.Ag) ? sp[(int64_t)a * PQ] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < HSLOTS; j++) {              // L halo, [i][cr*kh + r][hx]
            float v = 0.f;
            const int row = row0 + h_cr[j];
            if (h_off[j] >= 0 && row < p.rows_total) {
                const int n = row / p.P, pp = row - n * p.P;
                const int ly = pp + h_r[j] - p.pad_h, lx = q0 + h_hx[j] - p.pad_w;
                if ((unsigned)ly < (unsigned)p.LH && (unsigned)lx < (unsigned)p.LW)
                    v = p.L[((int64_t)n * p.Bg + h_i[j]) * LHW + ly * p.LW + lx];
            }
            hreg[j] = v;
        }
    };
    if (c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                                // the previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < 8; j++) Ss[(s_a0 + 8 * j) * SPITCH + s_k] = sreg[j];
#pragma unroll
        for (int j = 0; j < HSLOTS; j++)
            if (h_off[j] >= 0) Ls[h_off[j]] = hreg[j];
        if (tid == 0) Ls[halo_elems] = 0.f;
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll 4
        for (int kk = 0; kk < KP / 2; kk++) {
            const int k = kk * 2 + kl;
            const float af = Ss[(wa * 32 + jl) * SPITCH + k];
            const int koff = (k >> p.cw_log2) * p.kh * HW_ + (k & (CW - 1));
#pragma unroll
            for (int t = 0; t < MAXT; t++) {
                if (wb + 2 * t >= p.nb) continue;      // uniform per wave
                const float bf = Ls[boff[t] >= 0 ? boff[t] + koff : halo_elems];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
            }
        }
    }

    // slab [ksplit][a_pad][nb*32]
    const int bpad = p.nb * 32;
    float* out = p.slab + (int64_t)ks * p.a_tiles * 64 * bpad;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (wb + 2 * t >= p.nb) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = (wb + 2 * t) * 32 + jl;
            out[(int64_t)a * bpad + b] = acc[t][r];
        }
    }
}
#endif  // PASTA_TU_WGRAD_F32

// dw[o][b'] = sum_ks slab[ks][o][b']   (b' already in PyTorch's [i][r][s] order)
#ifdef PASTA_TU_WGRAD_F32  // defined by conv_tu_wgrad_f32.hip only (conv_launch.h)
__global__ __launch_bounds__(256) void wgrad_smallcin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ksplit,
                                                                    int Ag, int bprime, int a_pad, int bpad, float wscale) {
    const int total = Ag * bprime;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int a = idx / bprime, b = idx - a * bprime;
        const float* src = slab + (int64_t)a * bpad + b;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;          // as wgrad_reduce_kernel: sixteen slabs in flight, fixed order
        const int64_t stride = (int64_t)a_pad * bpad;
        int k = 0;
        for (; k + 16 <= ksplit; k += 16) {
            float r[16];
#pragma unroll
            for (int j = 0; j < 16; j++) r[j] = src[(k + j) * stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { v0 += r[j]; v1 += r[j + 1]; v2 += r[j + 2]; v3 += r[j + 3]; }
        }
        for (; k < ksplit; k++) v0 += src[k * stride];
        dw[idx] = ((v0 + v1) + (v2 + v3)) * wscale;
    }
}
#endif  // PASTA_TU_WGRAD_F32

struct WgradSmallPlan { bool use; int nb, bprime, cw_log2, qblocks, chunks_total, ksplit, a_tiles, rows_total; int64_t slab_floats; size_t lds_bytes; };

static WgradSmallPlan plan_wgrad_small(const pasta_conv_desc* d) {
    WgradSmallPlan w; w.use = false;
    const int Ig = d->C_in / d->groups;
    if (d->transposed || d->groups != 1 || d->stride != 1 || d->flip || Ig > 8 || Ig * d->kh * d->kw > 160) return w;
    w.use = true;
    w.bprime = Ig * d->kh * d->kw; w.nb = (w.bprime + 31) / 32;
    int cw = 32, lg = 5;
    while (cw > 1 && cw / 2 >= d->OW) { cw /= 2; lg--; }
    const int chh = 32 / cw;
    w.cw_log2 = lg; w.rows_total = d->N * d->OH;
    w.qblocks = (d->OW + cw - 1) / cw;
    w.chunks_total = ((w.rows_total + chh - 1) / chh) * w.qblocks;
    w.a_tiles = (d->C_out + 63) / 64;
    int64_t ks = (1024 + w.a_tiles - 1) / w.a_tiles;        // four workgroups per CU: one chunk in flight each
    if (ks > w.chunks_total / 8) ks = w.chunks_total / 8;
    if (ks < 1) ks = 1;
    w.ksplit = (int)ks;
    w.slab_floats = (int64_t)w.ksplit * w.a_tiles * 64 * w.nb * 32;
    w.lds_bytes = (size_t)(64 * 33 + Ig * chh * d->kh * (cw + d->kw - 1) + 4) * sizeof(float);
    if (Ig * chh * d->kh * (cw + d->kw - 1) > 1024) w.use = false;      // four halo slots per thread in the kernel
    return w;
}

//------------------------------------------------------------------------------------
// Pointwise weight gradient with very few input channels (round 4: the discriminator's fromrgb, 3 -> 64 over 48 stacked images, the pose /
// parsing stems): dW[o][i] = sum_{n, pixel} dy[n, o, pixel] x[n, i, pixel] is ONE pass over dy (805 MB for the fromrgb layer) and 3 .. 8
// multiply-adds per element -- bandwidth-bound, and the (channel, tap)-pair MFMA kernel above, built for 7 x 7 x 3 = 147 columns, runs it at
// 1.3 TB/s with three of 32 columns filled (0.67 ms; profiles/r4_byshape_classes.txt).  Here: plain fp32 FMAs; a thread takes four
// consecutive pixels (16-byte loads) of eight output channels and all CI input channels per trip -- CI + 8 loads in flight -- and keeps
// 8 x CI sums; a workgroup walks one K slice (pixel quads of the whole batch) for one group of eight output channels; sums are combined by
// wave shuffles and a fixed-order pass over the four waves (bitwise reproducible) and land in the slab layout of the kernel above, whose
// reduction kernel finishes the job.
template <int CI, int IO = IO_F32>      // IO (round 5): storage type of dy and x (16-bit storage: converted on the way in; fp32 sums)
__global__ __launch_bounds__(256) void wgrad1x1_fewcin_kernel(const void* __restrict__ dyv, const void* __restrict__ xv_, float* __restrict__ slab,
                                                              int N, int Co, int HW, int64_t quads_per_slice, int a_pad, int bpad) {
    constexpr int ES = io_size<IO>::value;
    const char* const dy = (const char*)dyv; const char* const x = (const char*)xv_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int og = blockIdx.y * 8;
    const int hwq = HW >> 2;
    const int64_t total = (int64_t)N * hwq;
    const int64_t q0 = (int64_t)blockIdx.x * quads_per_slice;
    const int64_t q1 = q0 + quads_per_slice < total ? q0 + quads_per_slice : total;
    float acc[8][CI];
#pragma unroll
    for (int o = 0; o < 8; o++)
#pragma unroll
        for (int i = 0; i < CI; i++) acc[o][i] = 0.f;
    for (int64_t q = q0 + tid; q < q1; q += 256) {
        const int n = (int)(q / hwq);
        const int off = (int)(q - (int64_t)n * hwq) * 4;
        float4 xv[CI], dv[8];
#pragma unroll
        for (int i = 0; i < CI; i++) xv[i] = io_ld4<IO>(x + (((int64_t)n * CI + i) * HW + off) * ES);
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int oc = og + o < Co ? og + o : Co - 1;          // rows beyond C_out re-read the last one; their sums are not stored
            dv[o] = io_ld4<IO>(dy + (((int64_t)n * Co + oc) * HW + off) * ES);
        }
#pragma unroll
        for (int o = 0; o < 8; o++)
#pragma unroll
            for (int i = 0; i < CI; i++) {
                float a = acc[o][i];
                a = fmaf(dv[o].x, xv[i].x, a); a = fmaf(dv[o].y, xv[i].y, a); a = fmaf(dv[o].z, xv[i].z, a); a = fmaf(dv[o].w, xv[i].w, a);
                acc[o][i] = a;
            }
    }
    __shared__ float part[4][8 * CI];
#pragma unroll
    for (int o = 0; o < 8; o++)
#pragma unroll
        for (int i = 0; i < CI; i++) {
            float v = acc[o][i];
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
            if (lane == 0) part[wave][o * CI + i] = v;
        }
    __syncthreads();
    if (tid < 8 * CI) {
        const int o = tid / CI, i = tid - o * CI;
        if (og + o < Co)
            slab[((int64_t)blockIdx.x * a_pad + og + o) * bpad + i] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    }
}

// Does the few-channel pointwise kernel take this weight gradient, and with how many K slices (<= the slab the small-cin plan reserved)?
static int plan_wgrad1x1_fewcin(const pasta_conv_desc* d, const WgradSmallPlan& ws) {
    static const bool enabled = !(getenv("PASTA_WGRAD_FEWCIN") && getenv("PASTA_WGRAD_FEWCIN")[0] == '0');         // A/B switch
    if (!enabled || !ws.use || d->kh != 1 || d->kw != 1 || d->pad_h || d->pad_w || d->C_in > 8) return 0;      // (any storage type: round 5)
    const int64_t hw = (int64_t)d->H * d->W;
    if (hw % 4 || d->OH != d->H || d->OW != d->W) return 0;
    int64_t ks = (int64_t)d->N * (hw / 4) / (256 * 8);          // at least eight trips per thread
    ks = ks < 1 ? 1 : ks > 256 ? 256 : ks;
    return (int)(ks < ws.ksplit ? ks : ws.ksplit);
}

}  // namespace pasta
