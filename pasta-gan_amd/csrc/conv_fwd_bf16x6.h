// Split-bf16 forward-type kernels (v_mfma_f32_32x32x16_bf16, fp32-equivalent products): weight packing, conv_fwd_bf16x6_kernel, conv_fwd_rows_bf16x6_kernel and their launchers.  Instantiated by conv_tu_pack_f32.hip (packing), conv_tu_fwd_base_*.hip, conv_tu_fwd_rows_*.hip.
#pragma once
#include "conv_common.h"

namespace pasta {

//------------------------------------------------------------------------------------
// Forward-type implicit GEMM on the bf16 matrix cores with fp32-equivalent products ("split-bf16").
//
// Every fp32 operand v is written as v = v1 + v2 + v3 with v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2):
// three bf16 pieces of 8 significand bits each carry the 24 bits of an fp32 significand, and a product of two bf16
// values is exact in fp32.  a*b is evaluated as a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (the three dropped terms are
// below 2^-24 |ab|), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs replace eight fp32 MFMAs of a
// quarter of the rate each: 2.67x the fp32-MFMA throughput at fp32 accuracy.
// Tile 128 x 128, K chunks of 16 channels of one tap; weights are split once by the packing kernel, activations by
// the staging code (after the optional modulation scale).  Layouts in LDS (per piece and per k-half of 8):
// [piece][half][row or pixel][8 bf16] so that a fragment is one conflict-free 16-byte read.



__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)v;
    float r = v - (float)a;
    b = (__bf16)r;
    r -= (float)b;
    c = (__bf16)r;
}

#ifdef PASTA_TU_PACK       // the packing kernels are defined by conv_tu_pack_f32.hip only (conv_launch.h)
// [g][tap][chunk of 16 channels][piece 3][half 2][O_pad][8]
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int G, int Ig,
                                                                int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed,
                                                                int flip, float wscale, int f16, const float* __restrict__ mod_s,
                                                                const float* __restrict__ mod_d) {
    // f16: 0 = three bf16 pieces, 1 = fp16 storage (leading piece = the fp16 operand).  PASTA_MATH_F16X3: pack_weights_f16x3_kernel
    const int64_t total = (int64_t)G * kh * kw * Ig_pad * Og_pad;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx & 7);
        int64_t r = idx >> 3;
        const int o = (int)(r % Og_pad); r /= Og_pad;
        const int half = (int)(r & 1); r >>= 1;
        const int cc = (int)(r % (Ig_pad / 16)); r /= (Ig_pad / 16);
        const int t = (int)(r % (kh * kw));
        const int g = (int)(r / (kh * kw));
        const int i = cc * 16 + half * 8 + j;
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
        __bf16 p1, p2, p3;
        split3(v, p1, p2, p3);
        if (f16 == 1) p1 = __builtin_bit_cast(__bf16, (_Float16)v);      // fp16 storage: the leading piece is the fp16 operand (the others are unused)
        const int64_t chunk = (((int64_t)g * kh * kw + t) * (Ig_pad / 16) + cc) * 6 * Og_pad * 8;
        const int64_t within = ((int64_t)half * Og_pad + o) * 8 + j;
        wp[chunk + within] = p1;
        wp[chunk + 2 * Og_pad * 8 + within] = p2;
        wp[chunk + 4 * Og_pad * 8 + within] = p3;
    }
}

// PASTA_MATH_F16X3 (conv_common.h): the same layout with the fp16 pieces h = fp16(v S), l = fp16(v S - h), h'' = h 2^-11 and the power-of-two scale S taken
// PER OUTPUT ROW from the row's own largest magnitude, found here: one workgroup owns one row o of one group over the whole K
// range (C_in kh kw values: at most a few thousand), first pass = fetch the row (every load of a thread in flight at once), form
// v = w wscale (s[g,i] d[g,o]), keep it in LDS and reduce |v| over the workgroup (non-finite values skipped, as the tensor scan
// does); second pass = split and store sixteen-byte units out of LDS, plus rowinv[g][o] = 1 / S for the convolution's epilogue.
// Hundreds of short workgroups: 5 - 8 us per layer like the element-parallel kernel it replaces (a first version with eight rows
// per workgroup and serial loads took 50 - 100 us: 341 launches per training step).  Nothing about w is cached between launches: a
// weight written behind autograd's back (`p.data.mul_()`, an optimiser that works on `.data`) cannot meet a stale scale, and
// per-sample modulated weights (pasta_conv2d_modulated) get their own scale per (sample, output channel).
constexpr int PACK_ROW_LDS = 8192;          // floats of a row kept in LDS; longer rows are re-read from global memory (L2)
__device__ __forceinline__ void pack_row_f16x3(const float* __restrict__ w, __bf16* __restrict__ wp, float* __restrict__ rowinv,
                                               int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                                               float wscale, const float* __restrict__ mod_s, const float* __restrict__ mod_d, int pack_xcd_rows, const unsigned bx) {
    // one packed row.  Eight consecutive rows share every 128-byte line of the packed layout ([...][O_pad][8 x 2 bytes]) and workgroups go round-robin
    // to the eight XCDs: XCD x takes the rows [x O_pad / 8, (x + 1) O_pad / 8), so that the sixteen-byte stores of a line meet in ONE L2 and leave it
    // as a whole line (O_pad is a multiple of 64)
    const int g = blockIdx.y, o = pack_xcd_rows ? (int)(bx & 7u) * (Og_pad >> 3) + (int)(bx >> 3) : (int)bx;
    const int taps = kh * kw, NC = Ig_pad / 16, K = Ig * taps;
    const int tid = threadIdx.x;
    const int gs = mod_s ? 0 : g;
    __shared__ float row[PACK_ROW_LDS];                     // v[i * taps + t], t = the tap index of the WEIGHT tensor (before the flip)
    __shared__ float wmax[4];
    const bool real = o < Og;
    const float md = (real && mod_d) ? mod_d[(int64_t)g * Og + o] : 1.f;
    // element k = i * taps + t of the row, as it lies in the weight tensor: the taps of one input channel are contiguous in both layouts
    auto tap_run = [&](int i) -> const float* {
        return w + (transposed ? ((int64_t)(gs * Ig + i) * Og + o) * taps : ((int64_t)(gs * Og + o) * Ig + i) * taps);
    };
    auto fetch = [&](int k) -> float {                       // rows longer than PACK_ROW_LDS: the second pass re-reads the tail
        const int i = k / taps, t = k - i * taps;
        float v = tap_run(i)[t] * wscale;
        if (mod_s) { v *= mod_s[(int64_t)g * Ig + i]; v *= md; }
        return v;
    };
    float m = 0.f;
    if (real) {
        if (taps <= 9) {
            // a thread owns input channels tid, tid + 256, ...: up to nine contiguous loads in flight, no division
            for (int i = tid; i < Ig; i += 256) {
                const float* const src = tap_run(i);
                const float ms = mod_s ? mod_s[(int64_t)g * Ig + i] : 1.f;
                float v[9];
#pragma unroll
                for (int t = 0; t < 9; t++) v[t] = t < taps ? src[t] : 0.f;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    if (t < taps) {
                        float x = v[t] * wscale;
                        if (mod_s) { x *= ms; x *= md; }
                        const int k = i * taps + t;
                        if (k < PACK_ROW_LDS) row[k] = x;
                        const float a = fabsf(x);
                        m = (a < __builtin_inff() && a > m) ? a : m;
                    }
                }
            }
        } else {
            for (int k0 = tid; k0 < K; k0 += 256 * 8) {          // eight loads in flight per thread and trip
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const int k = k0 + 256 * j; v[j] = k < K ? fetch(k) : 0.f; }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = k0 + 256 * j;
                    if (k < K && k < PACK_ROW_LDS) row[k] = v[j];
                    const float a = fabsf(v[j]);
                    m = (a < __builtin_inff() && a > m) ? a : m;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float S, inv;
    scale_from_amax(m, S, inv);
    if (tid == 0) rowinv[(int64_t)g * Og_pad + o] = inv;
    // second pass: unit u = (t_packed NC + cc) 2 + half -> sixteen bytes of h and of l: channels cc 16 + half 8 + 0..7 of packed tap t_packed
    const int units = taps * NC * 2;
    for (int u = tid; u < units; u += 256) {
        const int half = u & 1, k2 = u >> 1;
        const int cc = k2 % NC, tp = k2 / NC;
        int ty = tp / kw, tx = tp - ty * kw;
        if (flip) { ty = kh - 1 - ty; tx = kw - 1 - tx; }
        const int t = ty * kw + tx;
        uint32_t hq[4], lq[4], sq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = cc * 16 + half * 8 + 2 * j;
            float v0 = 0.f, v1 = 0.f;
            if (real) {
                const int k = i * taps + t;
                if (i < Ig) v0 = k < PACK_ROW_LDS ? row[k] : fetch(k);
                if (i + 1 < Ig) v1 = k + taps < PACK_ROW_LDS ? row[k + taps] : fetch(k + taps);
            }
            f16_split2_direct(v0 * S, v1 * S, hq[j], lq[j]);
            sq[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, hq[j]) * f16x2_t{(_Float16)0.00048828125f, (_Float16)0.00048828125f});
        }
        __bf16* const d = wp + (((int64_t)g * taps + tp) * NC + cc) * 6 * Og_pad * 8 + ((int64_t)half * Og_pad + o) * 8;
        *(uint4*)d = make_uint4(hq[0], hq[1], hq[2], hq[3]);
        *(uint4*)(d + 2 * Og_pad * 8) = make_uint4(lq[0], lq[1], lq[2], lq[3]);
        *(uint4*)(d + 4 * Og_pad * 8) = make_uint4(sq[0], sq[1], sq[2], sq[3]);
    }
}
__global__ __launch_bounds__(256) void pack_weights_f16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, float* __restrict__ rowinv,
                                                                 int Ig, int Og, int Ig_pad, int Og_pad, int kh, int kw, int transposed, int flip,
                                                                 float wscale, const float* __restrict__ mod_s, const float* __restrict__ mod_d, int pack_xcd_rows) {
    pack_row_f16x3(w, wp, rowinv, Ig, Og, Ig_pad, Og_pad, kh, kw, transposed, flip, wscale, mod_s, mod_d, pack_xcd_rows, blockIdx.x);
}
// Round 5: ONE weight tensor packed for TWO launches at once -- the forward convolution and its input gradient (the same weights transposed
// and mirrored) -- so that the backward pass finds its operand packed (pasta_conv2d_pack_pair; 120 of the 310 packing launches of a step).
// The first a.rows workgroups of the grid's x axis pack for a, the others for b.
__global__ __launch_bounds__(256) void pack_weights_f16x3_pair_kernel(const float* __restrict__ w, PackJob a, PackJob b) {
    const bool first = blockIdx.x < (unsigned)a.Og_pad;
    const PackJob& j = first ? a : b;
    pack_row_f16x3(w, (__bf16*)j.wp, j.rowinv, j.Ig, j.Og, j.Ig_pad, j.Og_pad, j.kh, j.kw, j.transposed, j.flip, j.wscale, nullptr, nullptr, j.pack_xcd_rows,
                   first ? blockIdx.x : blockIdx.x - (unsigned)a.Og_pad);
}
#endif  // PASTA_TU_PACK

// NP = bf16 pieces kept per operand: 3 = six products (fp32-equivalent, the default), 2 = three products (hi*hi, hi*mid,
// mid*hi: ~2^-16 relative, PASTA_MATH_BF16X3), 1 = one product (plain bf16 operands, PASTA_MATH_BF16).  The packed
// weights always hold three pieces; NP < 3 fetches and stages the leading ones only.
// IO = storage type of x / y / res (conv_common.h): 16-bit storage runs NP = 1 on the matching matrix-core type.
// ISC: the activations are multiplied by p.iscale[n, channel] (the styles of a modulated convolution, networks.py:74) on
// their way from the fetch registers to the split -- x * s rounded to fp32 exactly as a separate scaling pass would, so
// the modulated activation tensor never exists in HBM.
// KT (round 3, few input channels -- the RGB stems: 3 -> 64 at 7x7, 3x3, 1x1): the K loop runs over (input channel, tap) PAIRS
// instead of taps x 16-channel chunks, of which an RGB tensor fills three lanes in sixteen: one pseudo-tap, Ig = C_in kh kw "channels",
// channel k at byte offset p.koff[k] (scalar loads) from the pixel's base in a zero-padded copy of the input, so no bound is checked
// per element.  The weights are the tensor as it lies ([O][C_in kh kw] is a 1x1 weight over those channels).
template <int BM, int BN, int OCC, int NP, int IO = IO_F32, bool ISC = false, bool KT = false>      // (128, 128): waves 2 x 2;  (64, 256): waves 1 x 4; each wave 64 rows x 64 pixels
__global__ __launch_bounds__(256, OCC) void conv_fwd_bf16x6_kernel(ConvFwdParams p) {
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: the element is the operand, one product");
    static_assert(!KT || (IO == IO_F32 && !ISC), "the packed-K mode serves fp32 tensors without an input scale");
    constexpr unsigned ES = io_size<IO>::value;
    constexpr bool HX = Arith<NP>::f16x3;               // PASTA_MATH_F16X3 (conv_common.h)
    constexpr int NPA = Arith<NP>::npa, NPB = Arith<NP>::npb;
    constexpr int WMT = 2, WNT = 2, KC = 16;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "four waves per workgroup");
    constexpr int ASEG = BM * 8, BSEG = BN * 8;         // bf16 elements of one (piece, half) segment
    constexpr int AUNITS = 2 * NPA * BM;                // sixteen-byte units of the A chunk
    constexpr int APT = (AUNITS + 255) / 256;           // per thread: 3 (BM 128) or 2 (BM 64, second one guarded)
    constexpr int BPT = BN * 2 / 256;                   // (pixel, k-half) pairs per thread: 1 or 2
    // A buffers are rounded up to APT * 256 units: every thread copies APT units without a guard (see load_chunk)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][APT * 256 * 8];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][2 * NPB * BSEG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    // Workgroup -> (lattice class, pixel tile), class-major: the grid holds gridDim.x / ncls tiles for every class.
    int cls = 0;
    unsigned tile_x = blockIdx.x;
    if (p.ncls > 1) {
        const unsigned per = gridDim.x / (unsigned)p.ncls;
        cls = (int)(blockIdx.x / per);
        tile_x = blockIdx.x - (unsigned)cls * per;
    }
    const int P = p.cls[cls].P, Q = p.cls[cls].Q, oy0 = p.cls[cls].oy0, ox0 = p.cls[cls].ox0, T = p.cls[cls].T, tap0 = p.cls[cls].tap0;
    const int64_t npix = (int64_t)p.N * P * Q;
    const int64_t pix_blk = (int64_t)tile_x * BN;
    if (pix_blk >= npix) return;                   // grid is sized for the largest class
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int chunks_all = T * NC;
    const int c_first = (int)(((unsigned)chunks_all * (unsigned)ks) / (unsigned)p.ksplit);
    const int nchunks = (int)(((unsigned)chunks_all * (unsigned)(ks + 1)) / (unsigned)p.ksplit) - c_first;

    // B staging: this thread's pixel column (fixed for the whole K loop) and its k-halves:
    //   BN 128: one half, tid >> 7 (uniform per wave);  BN 256: both halves of pixel tid.
    const int bcol = tid & (BN - 1);
    const int64_t mypix = pix_blk + bcol;
    const bool pix_ok = mypix < npix;
    int n_in = 0, py = 0, px = 0;
    if (pix_ok) {
        n_in = (int)(mypix / (P * Q));
        const int rem = (int)(mypix - (int64_t)n_in * P * Q);
        py = rem / Q; px = rem - py * Q;
    }
    // Activation addressing: byte offset = (per-thread pixel part, VGPR) + (per-wave channel part, SGPR); the host only
    // selects this kernel for tensors below 2^30 elements, so 32-bit byte offsets suffice.
    const unsigned xb_off = (unsigned)(((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW) * ES;
    const char* const xbytes = (const char*)p.x;
    const int iy_base = py * p.isy, ix_base = px * p.isx;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int half0 = BPT == 1 ? __builtin_amdgcn_readfirstlane(tid >> 7) : 0;

    // Two register sets for the activations: while chunk c is multiplied out of LDS, chunk c+1 (already in registers)
    // is split into bf16 pieces and written to the other LDS buffer between the MFMAs, and chunk c+2 is being fetched.
    // The weights of chunk c+1 (already split, L2-resident: every workgroup reads the same ones) are fetched at the
    // start of step c and copied to LDS at its end.
    // The loop body is free of data-dependent control flow around its memory operations: every step issues the same
    // loads and stores (past the end of the K range they re-read valid addresses and the activations are zeroed), so
    // that the s_waitcnt counters the compiler derives let a fetch stay in flight for a whole step.
    float x_scale = 1.f, out_scale = 1.f;          // PASTA_MATH_F16X3: operand scales from the tensors' partial maxima
    if constexpr (HX) {
        float sx, isx;
        scale_from_amax(amax_of_parts(p.x_amax), sx, isx);
        x_scale = sx; out_scale = isx;                 // the weight rows carry their own scales: p.w_rowinv, applied per output row in the epilogue
    }
    struct Stage { float b[8 * BPT]; float sc[ISC ? 8 * BPT : 1]; int nvalid[BPT]; };
    Stage st0, st1;
    const float* const isb = ISC ? p.iscale + (int64_t)n_in * p.Cin + (int64_t)g * p.Ig : nullptr;
    float4 areg0, areg1, areg2;         // APT of them are used (scalars: an array here is not kept in registers)
    int ld_t = c_first / NC, ld_cc = c_first - ld_t * NC, ld_left = nchunks;
    bool ld_ok = false;
    unsigned ld_pix = xb_off;          // byte offset of this thread's tap pixel in channel 0 (a readable address also when the tap is outside)
    const __bf16* a_wt = wb;           // weight fetch position: tap slab and chunk within it
    int a_t = ld_t, a_cc = ld_cc;
    auto set_tap = [&](int t_in) {
        const int t = __builtin_amdgcn_readfirstlane(tap0 + t_in);   // the tap tables are read with scalar loads
        const int iy = iy_base + p.tap_dy[t], ix = ix_base + p.tap_dx[t];
        ld_ok = pix_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        ld_pix = ld_ok ? xb_off + (unsigned)(iy * p.W + ix) * ES : xb_off;
    };
    if (ld_t >= T) { ld_t = T - 1; ld_cc = 0; a_t = ld_t; a_cc = 0; }      // empty K slice: nothing is accumulated, addresses stay valid
    set_tap(ld_t);
    a_wt = wb + (int64_t)p.tap_slab[tap0 + a_t] * NC * 6 * p.Og_pad * 8;
    auto load_chunk = [&](Stage& st) {
        const int cc = __builtin_amdgcn_readfirstlane(ld_cc);
        const int last = p.Ig - 1;
        const bool real = ld_left > 0;              // chunks past the end of this K slice contribute zeros
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int c0 = cc * KC + (half0 + i) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                unsigned coff;                                                                           // scalar
                if constexpr (KT) coff = p.koff[c0 + j < last ? c0 + j : last];
                else coff = (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * ES;
                st.b[8 * i + j] = io_ld<IO>(xbytes, ld_pix + coff);
                if constexpr (ISC) st.sc[8 * i + j] = isb[c0 + j < last ? c0 + j : last];
            }
            st.nvalid[i] = (ld_ok && real) ? p.Ig - c0 : 0;     // elements j < nvalid are real
        }
        --ld_left;
        if (++ld_cc >= NC) {
            ld_cc = 0;
            if (ld_t + 1 < T) set_tap(++ld_t);
        }
    };
    auto load_a = [&]() {
        const __bf16* wt = a_wt + (int64_t)a_cc * 6 * p.Og_pad * 8;
        // 6 (piece, half) segments of BM sixteen-byte units each; this thread copies units tid, tid + 256, ...
        // (BM 64: the last 128 threads repeat unit AUNITS - 1 into the padding of the LDS buffer)
        auto unit = [&](int j) {
            int e = tid + 256 * j;
            if (256 * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;
            const int seg = e / BM, within = e - seg * BM;
            return *(const float4*)(wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8);
        };
        areg0 = unit(0);
        if (APT > 1) areg1 = unit(1);
        if (APT > 2) areg2 = unit(2);
        if (++a_cc >= NC) {
            a_cc = 0;
            if (a_t + 1 < T) a_wt = wb + (int64_t)p.tap_slab[__builtin_amdgcn_readfirstlane(tap0 + ++a_t)] * NC * 6 * p.Og_pad * 8;
        }
    };
    uint32_t q1[BPT][4], q2[BPT][4], q3[BPT][4];          // 8 bf16 per piece, packed two per dword
    // Two elements at a time: v_cvt_pk_bf16_f32 yields the packed pair, the residuals come from its halves.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_pair = [&](const Stage& st, int i, int j) {
        float v0 = st.b[8 * i + 2 * j], v1 = st.b[8 * i + 2 * j + 1];
        if constexpr (ISC) { v0 *= st.sc[8 * i + 2 * j]; v1 *= st.sc[8 * i + 2 * j + 1]; }
        if (st.nvalid[i] < 8) {                  // border pixel or channel tail
            v0 = 2 * j < st.nvalid[i] ? v0 : 0.f;
            v1 = 2 * j + 1 < st.nvalid[i] ? v1 : 0.f;
        }
        if constexpr (HX) {
            f16_split2(v0 * x_scale, v1 * x_scale, q1[i][j], q2[i][j]);
            return;
        }
        f32x2 v = {v0, v1};
        uint32_t w = io_pack2<IO>(v0, v1);
        q1[i][j] = w;
        if constexpr (NP >= 2) {
            // the two residual subtractions stay scalar: packed f32 VALU next to MFMAs costs more than it saves
            // (MI355X_MICROARCH.md, cycle table), and the empty asm keeps the SLP vectoriser from pairing them
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q2[i][j] = w;
        }
        if constexpr (NP >= 3) {
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            q3[i][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            __bf16* bd = &Bs[buf][((half0 + i) * BN + bcol) * 8];
            *(uint4*)(bd) = make_uint4(q1[i][0], q1[i][1], q1[i][2], q1[i][3]);
            if constexpr (NPB >= 2) *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[i][0], q2[i][1], q2[i][2], q2[i][3]);
            if constexpr (NPB >= 3) *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[i][0], q3[i][1], q3[i][2], q3[i][3]);
        }
    };
    auto store_a = [&](int buf) {
        *(float4*)&As[buf][tid * 8] = areg0;
        if (APT > 1) *(float4*)&As[buf][(tid + 256) * 8] = areg1;
        if (APT > 2) *(float4*)&As[buf][(tid + 512) * 8] = areg2;
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int hl = lane >> 5, jl = lane & 31;
    uint32_t y_am = 0;
    const AmaxSlot y_slot = amax_begin(p.y_amax);      // the slot's present value arrives behind the K loop
    // Fragments of one chunk: [tile][piece], read in the order the MFMA groups consume them.
    struct Frag { bf16x8 a[WMT][3], b[WNT][3]; };
    auto read_frag = [&](Frag& f, int buf) {
#define PASTA_LDA(PC) if constexpr ((PC) < NPA) { _Pragma("unroll") for (int a = 0; a < WMT; a++) f.a[a][PC] = *(const bf16x8*)&As[buf][(((PC) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8]; }
#define PASTA_LDB(PC) if constexpr ((PC) < NPB) { _Pragma("unroll") for (int b = 0; b < WNT; b++) f.b[b][PC] = *(const bf16x8*)&Bs[buf][(((PC) * 2 + hl) * BN + (wn * WNT + b) * 32 + jl) * 8]; }
        // in the order the product groups consume them, so that the first group starts when ITS operands have landed (the LDS
        // returns reads in order: counted lgkmcnt waits): three-product arithmetic (h'' l'), (l h), (h h)
        if constexpr (HX) { PASTA_LDA(2) PASTA_LDB(1) PASTA_LDA(1) PASTA_LDB(0) PASTA_LDA(0) }
        else { PASTA_LDA(2) PASTA_LDB(0) PASTA_LDA(0) PASTA_LDB(2) PASTA_LDA(1) PASTA_LDB(1) }
#undef PASTA_LDA
#undef PASTA_LDB
    };
    // One K chunk: 24 MFMAs in six groups of four; the staging work for the next chunk is slotted between the groups.
    auto step = [&](int buf, Stage& cur_next, Stage& fetch_into) {
        load_a();                   // weights of the next chunk first: they are waited for with the activation fetch still in flight
        load_chunk(fetch_into);
        Frag f;
        read_frag(f, buf);
#define PASTA_MM(PA, PB)                                                                                       \
        if constexpr (mm_on<NP>(PA, PB)) {                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            acc[a][b] = mfma16<IO, NP>(f.a[a][PA], f.b[b][PB], acc[a][b]); }
#define PASTA_SPLIT(J) _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(cur_next, i, J);
        // smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1
        if constexpr (HX) {
            PASTA_MM(2, 1)
            PASTA_SPLIT(0)
            PASTA_SPLIT(1)
            PASTA_MM(1, 0)
            PASTA_SPLIT(2)
            PASTA_SPLIT(3)
        } else {
        PASTA_MM(2, 0)
        PASTA_SPLIT(0)
        PASTA_MM(0, 2)
        PASTA_SPLIT(1)
        PASTA_MM(1, 1)
        PASTA_SPLIT(2)
        PASTA_MM(1, 0)
        PASTA_SPLIT(3)
        PASTA_MM(0, 1)
        }
        store_b(buf ^ 1); store_a(buf ^ 1);
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        __syncthreads();
    };

    load_a();
    load_chunk(st0);
#pragma unroll
    for (int i = 0; i < BPT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) split_pair(st0, i, j);
    store_b(0); store_a(0);
    load_chunk(st0);                            // chunk 1 waits in registers
    __syncthreads();
    // two chunks per trip so that the register sets swap roles without copies; an odd count runs one all-zero chunk
    for (int ch = 0; ch < nchunks; ch += 2) {
        step(0, st0, st1);                      // st0 holds chunk ch+1, chunk ch+2 is fetched into st1
        step(1, st1, st0);
    }

    const int OHW = p.OH * p.OW;
    if constexpr (HX) {
        // back to the operands' units: 1 / S_x for the tile, 1 / S_w per weight row (p.w_rowinv, written by the packing kernel).  The 32 row
        // scales of this lane are fetched in one go in front of the stores (a load in front of every store cost 9 % of the kernel).
        const float* const wri = p.w_rowinv + (int64_t)g * p.Og_pad + o_blk;
        float ws[WMT][16];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) ws[a][r] = wri[(wm * WMT + a) * 32 + acc_row(r, lane)];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int b = 0; b < WNT; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    acc[a][b][r] = (acc[a][b][r] * out_scale) * ws[a][r];
                    
                }
    }
    const bool has_noise = p.noise && p.ksplit == 1;
    const float nstr = has_noise ? p.noise_strength[0] : 0.f;
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, p.ksplit == 1);
    conv_epilogue_dispatch<(NP == NP_F16X3 || IO != IO_F32)>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
    const bool FULL = full_c;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        if (pix >= npix) continue;
        const int n = (int)(pix / (P * Q));
        const int rem = (int)(pix - (int64_t)n * P * Q);
        const int pp = rem / Q, qq = rem - pp * Q;
        const int plane_off = (oy0 + pp * p.osy) * p.OW + ox0 + qq * p.osx;
        const int64_t yoff = ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
        const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n * OHW : 0) + plane_off] * nstr : 0.f;
        float* pb = p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW + yoff : nullptr;      // K slices: fp32 partial sums
        const bool has_res = p.res && p.ksplit == 1;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++) {
            // what the sixteen stores of a 32 x 32 sub-tile need from memory -- output scale, residual, bias -- is fetched in front of them
            // (a load in front of every store serialises on the memory counter: conv_fwd_rows2d_bf16x6.h)
            // (one operand kind at a time through the same sixteen registers: three arrays side by side spill on the three-workgroup tiles)
            float tv[16];
            const bool has_bias = p.act && p.ksplit == 1 && p.bias;
            if (osb) {
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = osb[o < p.Og ? o : p.Og - 1]; }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = fmaf(acc[a][b][r], tv[r], nz);
            } else if (has_noise) {
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += nz;
            }
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    tv[r] = o < p.Og ? io_ld1<IO>((const char*)p.res + (yoff + (int64_t)o * OHW) * ES) : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += tv[r];
            }
            if (pb) {                                   // K slices: fp32 partial sums, nothing else (scale, residual, bias, activation ride in the reduction)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    if (o < p.Og) pb[(int64_t)o * OHW] = acc[a][b][r];
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = has_bias ? p.bias[g * p.Og + (o < p.Og ? o : p.Og - 1)] : 0.f; }
            // the stores: instantiated per (activation, clamp, whole tile of rows), no wave-uniform branch per element (conv_common.h)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                const float v = conv_epilogue_c(acc[a][b][r], tv[r], ea, case_c);
                if (FULL || o < p.Og) { io_st<IO>(p.y, yoff + (int64_t)o * OHW, v); amax_take(y_am, v); }
            }
        }
    }
    });
    if (p.ksplit == 1) amax_commit(y_am, y_slot);
}


//------------------------------------------------------------------------------------
// Row-reuse variant of conv_fwd_bf16x6_kernel for stride-1 lattices whose taps form kh rows of three horizontally
// adjacent offsets (every 3x3 stride-1 convolution and its input gradient: 85 % of the forward-type FLOPs of the step).
// A pixel tile is R = BN / SEG row segments of SEG = min(Q, BN) consecutive pixels.  The activations of one input row
// (kernel row dy) and one 16-channel chunk are fetched, split into bf16 pieces and stored to LDS ONCE, with one halo
// pixel on either side of every segment ([piece][k-half][slot][8 bf16], slot = pixel + 2 * segment + 1), and the three
// horizontal taps read their B fragments from that image at slot offsets 0, 1, 2: per 72 MFMAs one activation fetch
// and split instead of three.  The weights are fetched per tap as in the base kernel.
// K loop: "stages" (dy, chunk) of three steps (the taps of the row).  Step 0 of a stage issues the loads of the next
// stage (halo pixels first: one wave, the waves take turns), step 1 splits and stores its main pixels, step 2 its halo
// pixels; the B
// image is double-buffered per stage, the A image per step.  Control flow around memory operations is static as in
// the base kernel.
// Schedules (PASTA_ROWS_PIPE selects; all bit-compatible in their results up to the order of the six products):
//   1 (default)  below.   0: weights fetched and stored within one step.
//   2  register diet, three workgroups per CU (157 VGPRs): +2-4 % on the 128^2 SPADE layers, -4 % at 64^2 and 32^2.
//   3  weights straight to registers, one barrier per stage instead of per step, half the LDS traffic: +-0 %.
// Round-2 finding (profiles/r2_ablation_rows.txt): three workgroups per CU, a third of the barriers,
// half of the LDS traffic -- none of it moves the kernel, while removing either operand's global loads in a timing-only
// build gains 20-24 %.  The kernel behaves as power / clock limited at ~1.25-1.3 PFLOP/s of executed bf16 MFMA (the
// guide's tuned 256^2 GEMM template reaches 1.32-1.47 on random operands): what remains is energy per product, not schedule.
// PIPE = 1 (default): the weights of a step are fetched TWO steps ahead into one of two register sets and stored to LDS a
// full step later, so the store never waits for an L2 round trip (+1.5 % on the 128 x 128 x 128 layers; PIPE = 0 fetches
// and stores within one step).  Also measured and dropped: reading the next step's fragments during the current step's
// MFMAs (two fragment sets, 236-256 VGPRs) -- no change, the other workgroup of the CU already covers that latency.
// PAIR (stride-2 conv_transpose2d, 3x3: every upsampling layer and the input gradient of every stride-2 convolution): the
// lattice is the INPUT plane, a workgroup owns one vertical output parity a and BOTH horizontal parities of its pixels.
// The kernel row r (r = a + pad mod 2: one or two rows per a) is a stage as above; of its three taps, c = 0 and c = 2 belong
// to the output column 2q + (pad & 1) (input offsets dx, dx - 1), c = 1 to the other column -- two accumulator sets, the same
// B image (offsets p.pair_off[c]).  Against one launch per parity class: the K loop of a workgroup is three or six taps
// long instead of one to four, the input is fetched and split once per row instead of once per tap, and an output row
// leaves as whole 8-byte pairs (2q, 2q + 1) instead of every second float.  Runs on the register-diet schedule (PIPE 2),
// which leaves room for the second accumulator set.  Outputs beyond 2H x 2W (one row / column when OH = 2H + 1) are a
// small launch of conv_fwd_bf16x6_kernel over remainder lattices.
template <int BM, int BN, int OCC, int PIPE, int NP, int IO = IO_F32, bool ISC = false, bool PAIR = false>     // ISC: as in conv_fwd_bf16x6_kernel
__global__ __launch_bounds__(256, OCC) void conv_fwd_rows_bf16x6_kernel(ConvFwdParams p) {
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: the element is the operand, one product");
    static_assert(!ISC || PIPE <= 1, "the input scale is staged by the default schedules only");
    static_assert(!PAIR || (PIPE == 2 && !ISC), "the parity-pair mode runs on the register-diet schedule");
    constexpr unsigned ES = io_size<IO>::value;
    constexpr bool HX = Arith<NP>::f16x3;               // PASTA_MATH_F16X3 (conv_common.h): default schedules only
    static_assert(!HX || PIPE <= 1 || (PIPE == 2 && PAIR), "the three-product fp16 arithmetic runs on the default schedules and in the parity-pair mode");
    constexpr int NPA = Arith<NP>::npa, NPB = Arith<NP>::npb;
    constexpr int WMT = 2, WNT = 2, KC = 16;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "four waves per workgroup");
    constexpr int AUNITS = 2 * NPA * BM;
    constexpr int APT = (AUNITS + 255) / 256;
    constexpr int BPT = BN * 2 / 256;                   // (pixel, k-half) pairs per thread: 1 or 2
    constexpr int SLOTS = BN + 16;                      // up to 8 segments with two halo slots each
    constexpr int ABUF = APT * 256 * 8, BSEG = SLOTS * 8, BBUF = 2 * NPB * BSEG;     // 16-bit elements
    extern __shared__ __attribute__((aligned(16))) __bf16 rows_smem[];
    __bf16* const As = rows_smem;                       // [2][ABUF]  (PIPE 3: the weights never enter LDS)
    __bf16* const Bs = rows_smem + (PIPE == 3 ? 0 : 2 * ABUF);            // [2][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    int cls_i = 0;
    unsigned tile_x = blockIdx.x;
    if constexpr (PAIR) {                               // first half of the grid: the vertical parity with two kernel rows
        const unsigned per = gridDim.x >> 1;
        cls_i = blockIdx.x >= per ? 1 : 0;
        tile_x = blockIdx.x - (unsigned)cls_i * per;
    }
    const int P = p.cls[cls_i].P, Q = p.cls[cls_i].Q, oy0 = p.cls[cls_i].oy0, ox0 = p.cls[cls_i].ox0, T = p.cls[cls_i].T;
    const int tap0 = PAIR ? p.cls[cls_i].tap0 : 0;
    const int KH = T / 3;
    const int64_t pix_blk = (int64_t)tile_x * BN;       // the host guarantees full tiles inside one image
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int stages_all = KH * NC;
    const int s_first = (int)(((unsigned)stages_all * (unsigned)ks) / (unsigned)p.ksplit);
    const int nstages = (int)(((unsigned)stages_all * (unsigned)(ks + 1)) / (unsigned)p.ksplit) - s_first;
    const int seg_log2 = 31 - __builtin_clz(Q < BN ? Q : BN);
    const int SEG = 1 << seg_log2, R = BN >> seg_log2;
    const int d0 = p.rows_d0;

    // main pixel of this thread (fixed): tile pixel bcol -> image (n, py, px); its slot keeps one halo slot per segment free
    const int bcol = tid & (BN - 1);
    const int n_in = (int)((pix_blk + bcol) / (P * Q));
    const int rem_in = (int)(pix_blk + bcol - (int64_t)n_in * P * Q);
    const int py = rem_in / Q, px = rem_in - py * Q;
    const int m_slot = bcol + 2 * (bcol >> seg_log2) + 1;
    const int m_cx = px + d0 + 1;
    const unsigned xb_off = (unsigned)(((int64_t)n_in * p.Cin + (int64_t)g * p.Ig) * HW) * ES;
    // halo pixels (one wave per stage, in turn): BN 128: lane = segment * 4 + side * 2 + k-half; BN 256: lane = segment * 2 + side, both halves.
    // Lanes beyond the last segment repeat it (identical data to the identical slot).
    int h_r = BPT == 1 ? lane >> 2 : lane >> 1;
    h_r = h_r < R ? h_r : R - 1;
    const int h_side = BPT == 1 ? (lane >> 1) & 1 : lane & 1;
    const int h_half = BPT == 1 ? lane & 1 : 0;
    const int64_t h_pixel = pix_blk + ((int64_t)h_r << seg_log2);
    const int h_n = (int)(h_pixel / (P * Q));
    const int h_rem = (int)(h_pixel - (int64_t)h_n * P * Q);
    const int h_py = h_rem / Q, h_qs = h_rem - h_py * Q;
    const int h_slot = h_r * (SEG + 2) + (h_side ? SEG + 1 : 0);
    const int h_cx = h_side ? h_qs + SEG + d0 + 1 : h_qs + d0;
    const unsigned hb_off = (unsigned)(((int64_t)h_n * p.Cin + (int64_t)g * p.Ig) * HW) * ES;

    const char* const xbytes = (const char*)p.x;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int half0 = BPT == 1 ? __builtin_amdgcn_readfirstlane(tid >> 7) : 0;
    const int64_t a_chunk = (int64_t)6 * p.Og_pad * 8;             // bf16 elements of one packed 16-channel chunk

    // ---- fetch state of the activations: the stage (kernel row b_dy, chunk b_cc) that the next load_b() fetches
    int b_dy = s_first / NC, b_cc = s_first - b_dy * NC, b_left = nstages;
    if (b_dy >= KH) { b_dy = KH - 1; b_cc = 0; }
    bool m_ok = false, h_ok = false;
    unsigned m_pix = xb_off, h_pix = hb_off;
    auto set_row = [&](int dyi) {
        const int dy = p.tap_dy[__builtin_amdgcn_readfirstlane(tap0 + 3 * dyi)];
        const int iy = py + dy, hy = h_py + dy;
        m_ok = (unsigned)iy < (unsigned)p.H && (unsigned)m_cx < (unsigned)p.W;
        m_pix = m_ok ? xb_off + (unsigned)(iy * p.W + m_cx) * ES : xb_off;
        h_ok = (unsigned)hy < (unsigned)p.H && (unsigned)h_cx < (unsigned)p.W;
        h_pix = h_ok ? hb_off + (unsigned)(hy * p.W + h_cx) * ES : hb_off;
    };
    set_row(b_dy);
    float x_scale = 1.f, out_scale = 1.f;          // PASTA_MATH_F16X3: operand scales from the tensors' partial maxima
    if constexpr (HX) {
        float sx, isx;
        scale_from_amax(amax_of_parts(p.x_amax), sx, isx);
        x_scale = sx; out_scale = isx;                 // the weight rows carry their own scales: p.w_rowinv, applied per output row in the epilogue
    }
    float mb[8 * BPT], hb[8 * BPT];
    float msc[ISC ? 8 * BPT : 1], hsc[ISC ? 8 * BPT : 1];           // ISC: the input scales of the channels in mb / hb
    const float* const m_isb = ISC ? p.iscale + (int64_t)n_in * p.Cin + (int64_t)g * p.Ig : nullptr;
    const float* const h_isb = ISC ? p.iscale + (int64_t)h_n * p.Cin + (int64_t)g * p.Ig : nullptr;
    int m_nvalid[BPT], h_nvalid[BPT];
    int h_owner = 0;                                 // the wave that stages the halo pixels of the stage in flight
    auto load_b = [&]() {
        const int cc = __builtin_amdgcn_readfirstlane(b_cc);
        const int last = p.Ig - 1;
        const bool real = b_left > 0;
        h_owner = b_left & 3;
        if (wave == h_owner) {                       // oldest loads of the step: every later wait covers them
#pragma unroll
            for (int i = 0; i < BPT; i++) {
                const int c0 = cc * KC + (h_half + i) * 8;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    hb[8 * i + j] = io_ld<IO>(xbytes, h_pix + (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * ES);
                    if constexpr (ISC) hsc[8 * i + j] = h_isb[c0 + j < last ? c0 + j : last];
                }
                h_nvalid[i] = (h_ok && real) ? p.Ig - c0 : 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int c0 = cc * KC + (half0 + i) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned coff = (unsigned)(c0 + j < last ? c0 + j : last) * (unsigned)HW * ES;    // scalar
                mb[8 * i + j] = io_ld<IO>(xbytes, m_pix + coff);
                if constexpr (ISC) msc[8 * i + j] = m_isb[c0 + j < last ? c0 + j : last];
            }
            m_nvalid[i] = (m_ok && real) ? p.Ig - c0 : 0;
        }
        --b_left;
        if (++b_cc >= NC) {
            b_cc = 0;
            if (b_dy + 1 < KH) set_row(++b_dy);
        }
    };
    // ---- fetch state of the weights: one step ahead of the multiplication
    int a_dy = b_dy, a_cc = b_cc;
    const __bf16* a_w0 = wb; const __bf16* a_w1 = wb; const __bf16* a_w2 = wb;
    auto set_a_row = [&](int dyi) {
        const int t = __builtin_amdgcn_readfirstlane(tap0 + 3 * dyi);
        a_w0 = wb + (int64_t)p.tap_slab[t] * NC * a_chunk;
        a_w1 = wb + (int64_t)p.tap_slab[t + 1] * NC * a_chunk;
        a_w2 = wb + (int64_t)p.tap_slab[t + 2] * NC * a_chunk;
    };
    set_a_row(a_dy);
    float4 areg0, areg1, areg2;                      // weights in flight; the pipelined variant alternates with a second set
    float4 breg0, breg1, breg2;
    auto load_a = [&](int tap_i, int set = 0) {      // tap_i and set are compile-time constants at every call
        const __bf16* wt = (tap_i == 0 ? a_w0 : tap_i == 1 ? a_w1 : a_w2) + (int64_t)a_cc * a_chunk;
        auto unit = [&](int j) {
            int e = tid + 256 * j;
            if (256 * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;
            const int seg = e / BM, within = e - seg * BM;
            return *(const float4*)(wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8);
        };
        if (set == 0) {
            areg0 = unit(0);
            if (APT > 1) areg1 = unit(1);
            if (APT > 2) areg2 = unit(2);
        } else {
            breg0 = unit(0);
            if (APT > 1) breg1 = unit(1);
            if (APT > 2) breg2 = unit(2);
        }
    };
    auto next_a_stage = [&]() {
        if (++a_cc >= NC) {
            a_cc = 0;
            if (a_dy + 1 < KH) set_a_row(++a_dy);
        }
    };
    auto store_a = [&](int buf, int set = 0) {
        __bf16* d = As + buf * ABUF;
        *(float4*)&d[tid * 8] = set ? breg0 : areg0;
        if (APT > 1) *(float4*)&d[(tid + 256) * 8] = set ? breg1 : areg1;
        if (APT > 2) *(float4*)&d[(tid + 512) * 8] = set ? breg2 : areg2;
    };

    uint32_t q1[BPT][4], q2[BPT][4], q3[BPT][4];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_pair = [&](const float* b, const int* nvalid, int i, int j, const float* sc = nullptr) {
        float v0 = b[8 * i + 2 * j], v1 = b[8 * i + 2 * j + 1];
        if constexpr (ISC) { v0 *= sc[8 * i + 2 * j]; v1 *= sc[8 * i + 2 * j + 1]; }
        if (nvalid[i] < 8) {
            v0 = 2 * j < nvalid[i] ? v0 : 0.f;
            v1 = 2 * j + 1 < nvalid[i] ? v1 : 0.f;
        }
        if constexpr (HX) {
            f16_split2(v0 * x_scale, v1 * x_scale, q1[i][j], q2[i][j]);
            return;
        }
        f32x2 v = {v0, v1};
        uint32_t w = io_pack2<IO>(v0, v1);
        q1[i][j] = w;
        if constexpr (NP >= 2) {
            // the two residual subtractions stay scalar: packed f32 VALU next to MFMAs costs more than it saves
            // (MI355X_MICROARCH.md, cycle table), and the empty asm keeps the SLP vectoriser from pairing them
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q2[i][j] = w;
        }
        if constexpr (NP >= 3) {
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            q3[i][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        }
    };
    auto store_q = [&](int buf, int slot, int hbase) {
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            __bf16* bd = Bs + buf * BBUF + ((hbase + i) * SLOTS + slot) * 8;
            *(uint4*)(bd) = make_uint4(q1[i][0], q1[i][1], q1[i][2], q1[i][3]);
            if constexpr (NPB >= 2) *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[i][0], q2[i][1], q2[i][2], q2[i][3]);
            if constexpr (NPB >= 3) *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[i][0], q3[i][1], q3[i][2], q3[i][3]);
        }
    };

    f32x16 acc[WMT][WNT];
    f32x16 acc2[WMT][WNT];                           // PAIR: the output column of tap c = 1
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[a][b][r] = 0.f; if constexpr (PAIR) acc2[a][b][r] = 0.f; }

    const int hl = lane >> 5, jl = lane & 31;
    uint32_t y_am = 0;
    const AmaxSlot y_slot = amax_begin(p.y_amax);      // the slot's present value arrives behind the K loop
    int fslot[WNT];                                  // slot of this lane's pixel of B fragment b, for tap offset 0
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int t = (wn * WNT + b) * 32 + jl;
        fslot[b] = t + 2 * (t >> seg_log2);
    }
    struct Frag { bf16x8 a[WMT][3], b[WNT][3]; };
    auto read_frag = [&](Frag& f, int abuf, int bbuf, int off) {
        const __bf16* A_ = As + abuf * ABUF;
        const __bf16* B_ = Bs + bbuf * BBUF;
#define PASTA_LDA(PC) if constexpr ((PC) < NPA) { _Pragma("unroll") for (int a = 0; a < WMT; a++) f.a[a][PC] = *(const bf16x8*)&A_[(((PC) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8]; }
#define PASTA_LDB(PC) if constexpr ((PC) < NPB) { _Pragma("unroll") for (int b = 0; b < WNT; b++) f.b[b][PC] = *(const bf16x8*)&B_[(((PC) * 2 + hl) * SLOTS + fslot[b] + off) * 8]; }
        // in the order the product groups consume them, so that the first group starts when ITS operands have landed (the LDS
        // returns reads in order: counted lgkmcnt waits): three-product arithmetic (h'' l'), (l h), (h h)
        if constexpr (HX) { PASTA_LDA(2) PASTA_LDB(1) PASTA_LDA(1) PASTA_LDB(0) PASTA_LDA(0) }
        else { PASTA_LDA(2) PASTA_LDB(0) PASTA_LDA(0) PASTA_LDB(2) PASTA_LDA(1) PASTA_LDB(1) }
#undef PASTA_LDA
#undef PASTA_LDB
    };
    // One tap = one step: 24 MFMAs in six groups; TAP (0, 1, 2: position in the kernel row), ABUF_ and BBUF_ are literals.
    auto step = [&](const int TAP, const int abuf, const int bbuf) {
        if (PIPE == 1) {
            // weights of step t + 2 into the register set of this step's parity (abuf = parity); stored by step t + 1
            if (TAP == 1) next_a_stage();
            if (TAP == 0) load_b();
            load_a(TAP == 0 ? 2 : TAP == 1 ? 0 : 1, abuf);
        } else {
        if (TAP == 2) next_a_stage();
        if (TAP == 0) {
            // halo loads (wave 0) are issued inside load_b ahead of everything else of this step
            load_b();
            load_a(1);
        } else {
            load_a(TAP == 1 ? 2 : 0);
        }
        }
        Frag f;
        read_frag(f, abuf, bbuf, p.rows_rev ? 2 - TAP : TAP);
#define PASTA_MM(PA, PB)                                                                                       \
        if constexpr (mm_on<NP>(PA, PB)) {                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            acc[a][b] = mfma16<IO, NP>(f.a[a][PA], f.b[b][PB], acc[a][b]); }
#define PASTA_SPLIT(J)                                                                                         \
        if (TAP == 1) { _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(mb, m_nvalid, i, J, msc); }   \
        if (TAP == 2 && wave == h_owner) { _Pragma("unroll") for (int i = 0; i < BPT; i++) split_pair(hb, h_nvalid, i, J, hsc); }
        if constexpr (HX) {
            PASTA_MM(2, 1)
            PASTA_SPLIT(0)
            PASTA_SPLIT(1)
            PASTA_MM(1, 0)
            PASTA_SPLIT(2)
            PASTA_SPLIT(3)
        } else {
        PASTA_MM(2, 0)
        PASTA_SPLIT(0)
        PASTA_MM(0, 2)
        PASTA_SPLIT(1)
        PASTA_MM(1, 1)
        PASTA_SPLIT(2)
        PASTA_MM(1, 0)
        PASTA_SPLIT(3)
        PASTA_MM(0, 1)
        }
        if (TAP == 1) store_q(bbuf ^ 1, m_slot, half0);
        if (TAP == 2 && wave == h_owner) store_q(bbuf ^ 1, h_slot, h_half);
        store_a(abuf ^ 1, PIPE == 1 ? abuf ^ 1 : 0);        // PIPE 1: the set fetched by the previous step
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        __syncthreads();
    };

    // Register diet (PIPE == 2, launched at THREE workgroups per CU: <= 168 VGPRs):
    //  * fragments are read one operand at a time, in an order in which consecutive product groups share an operand:
    //    (a3,b1) (a2,b1) (a2,b2) (a1,b2) (a1,b3) (a1,b1): seven operand reads (fourteen ds_read_b128) instead of six pairs
    //    held at once -- 24 fragment registers live instead of 48.  The order of the six products within a chunk is free:
    //    the accumulator already holds the sum over all earlier chunks;
    //  * the split runs piece by piece with the residuals kept IN the staging registers, each piece stored as soon as it is
    //    complete: 4 packed registers live instead of 12;
    //  * the weights are fetched and stored within one step (one register set: PIPE 0's schedule).
    auto split_piece = [&](float* b, const int* nvalid, uint32_t (&q)[BPT][4], bool first) {
#pragma unroll
        for (int i = 0; i < BPT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v0 = b[8 * i + 2 * j], v1 = b[8 * i + 2 * j + 1];
                if (first && nvalid[i] < 8) {
                    v0 = 2 * j < nvalid[i] ? v0 : 0.f;
                    v1 = 2 * j + 1 < nvalid[i] ? v1 : 0.f;
                }
                const uint32_t w = io_pack2<IO>(v0, v1);
                q[i][j] = w;
                v0 -= __builtin_bit_cast(float, w << 16);
                v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
                PASTA_KEEP_SCALAR(v0);
                b[8 * i + 2 * j] = v0; b[8 * i + 2 * j + 1] = v1;
            }
    };
    auto store_piece = [&](int buf, int slot, int hbase, int piece, const uint32_t (&q)[BPT][4]) {
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            __bf16* bd = Bs + buf * BBUF + ((hbase + i) * SLOTS + slot) * 8 + piece * 2 * BSEG;
            *(uint4*)(bd) = make_uint4(q[i][0], q[i][1], q[i][2], q[i][3]);
        }
    };
    auto step_diet = [&](const int TAP, const int abuf, const int bbuf) {
        if (TAP == 2) next_a_stage();
        if (TAP == 0) { load_b(); load_a(1); } else { load_a(TAP == 1 ? 2 : 0); }
        const int off = PAIR ? p.pair_off[TAP] : p.rows_rev ? 2 - TAP : TAP;
        const __bf16* A_ = As + abuf * ABUF;
        const __bf16* B_ = Bs + bbuf * BBUF;
        bf16x8 fa[WMT], fb[WNT];
        auto lda = [&](int pc) {
#pragma unroll
            for (int a = 0; a < WMT; a++) fa[a] = *(const bf16x8*)&A_[((pc * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8];
        };
        auto ldb = [&](int pc) {
#pragma unroll
            for (int b = 0; b < WNT; b++) fb[b] = *(const bf16x8*)&B_[((pc * 2 + hl) * SLOTS + fslot[b] + off) * 8];
        };
        auto mm = [&]() {
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int b = 0; b < WNT; b++) {
                    if (PAIR && TAP == 1) acc2[a][b] = mfma16<IO, NP>(fa[a], fb[b], acc2[a][b]);
                    else acc[a][b] = mfma16<IO, NP>(fa[a], fb[b], acc[a][b]);
                }
        };
        const bool mine = TAP == 1 || (TAP == 2 && wave == h_owner);
        float* const sb = TAP == 1 ? mb : hb;
        const int* const sv = TAP == 1 ? m_nvalid : h_nvalid;
        const int sslot = TAP == 1 ? m_slot : h_slot, shalf = TAP == 1 ? half0 : h_half;
        uint32_t q[BPT][4];
        if constexpr (NP == 3) {
            lda(2); ldb(0); mm();                                   // a3 b1
            if (mine) { split_piece(sb, sv, q, true); store_piece(bbuf ^ 1, sslot, shalf, 0, q); }
            lda(1); mm();                                           // a2 b1
            ldb(1); mm();                                           // a2 b2
            if (mine) { split_piece(sb, sv, q, false); store_piece(bbuf ^ 1, sslot, shalf, 1, q); }
            lda(0); mm();                                           // a1 b2
            ldb(2); mm();                                           // a1 b3
            if (mine) { split_piece(sb, sv, q, false); store_piece(bbuf ^ 1, sslot, shalf, 2, q); }
            store_a(abuf ^ 1, 0);
            ldb(0); mm();                                           // a1 b1
        }
        if constexpr (HX) {
            // three products, five operand reads: (h'' l') (l h) (h h); the split yields both pieces at once (no residual chain)
            const float* const ssc = TAP == 1 ? msc : hsc;
            lda(2); ldb(1); mm();
            if (mine) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < BPT; i++) split_pair(sb, sv, i, j, ssc);
            }
            lda(1); ldb(0); mm();
            if (mine) {
#pragma unroll
                for (int j = 2; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < BPT; i++) split_pair(sb, sv, i, j, ssc);
                store_q(bbuf ^ 1, sslot, shalf);
            }
            store_a(abuf ^ 1, 0);
            lda(0); mm();
        }
        __syncthreads();
    };

    // Weights straight to registers (PIPE == 3).  The packed weight chunk is already in fragment order -- [piece][k-half]
    // [row][8 bf16]: the 64 lanes of a fragment read 2 x 512 contiguous bytes -- so each wave fetches ITS OWN A fragments
    // from global memory (L2 / L1 resident: every workgroup of an output-channel tile reads the same 885 KB per layer) one
    // step ahead, into the register set the previous step has just finished with.  The A operand never touches LDS: no
    // ds_write / ds_read for it, half the LDS footprint, and -- the point -- the workgroup barrier is needed only where
    // the B image changes hands, once per STAGE (72 MFMAs) instead of once per step (24).
    bf16x8 ar0[WMT][3], ar1[WMT][3];
    auto load_a_frags = [&](int tap_i, bf16x8 (&dst)[WMT][3]) {
        const __bf16* wt = (tap_i == 0 ? a_w0 : tap_i == 1 ? a_w1 : a_w2) + (int64_t)a_cc * a_chunk;
#pragma unroll
        for (int pc = 0; pc < NP; pc++)
#pragma unroll
            for (int a = 0; a < WMT; a++)
                dst[a][pc] = *(const bf16x8*)(wt + ((int64_t)(pc * 2 + hl) * p.Og_pad + o_blk + (wm * WMT + a) * 32 + jl) * 8);
    };
#ifndef PASTA_ABLATE
#define PASTA_ABLATE 0          // timing-only builds (results are garbage): 1 = no activation loads, 2 = no split / LDS stores of B, 4 = no weight loads, 8 = no MFMAs
#endif
    auto step_areg = [&](const int TAP, const int bbuf, bf16x8 (&cur)[WMT][3], bf16x8 (&nxt)[WMT][3]) {
        if (TAP == 2) next_a_stage();
        if (TAP == 0 && !(PASTA_ABLATE & 1)) load_b();
        if (!(PASTA_ABLATE & 4)) load_a_frags(TAP == 0 ? 1 : TAP == 1 ? 2 : 0, nxt);
        // The fetches above are for the NEXT step.  Left alone, the scheduler sinks each of them to just in front of its
        // first use to save registers (s_waitcnt vmcnt(1) / vmcnt(0) between the MFMAs of the next step: an L2 round trip
        // per fragment); the scheduling barrier keeps them here, a whole step ahead.
        __builtin_amdgcn_sched_barrier(0);
        const int off = p.rows_rev ? 2 - TAP : TAP;
        const __bf16* B_ = Bs + bbuf * BBUF;
        bf16x8 fb[WNT];
        auto ldb = [&](int pc) {
#pragma unroll
            for (int b = 0; b < WNT; b++) fb[b] = *(const bf16x8*)&B_[((pc * 2 + hl) * SLOTS + fslot[b] + off) * 8];
        };
        auto mm = [&](int pa) {
            if (PASTA_ABLATE & 8) {
#pragma unroll
                for (int b = 0; b < WNT; b++) asm volatile("" :: "v"(fb[b]));
#pragma unroll
                for (int a = 0; a < WMT; a++) asm volatile("" :: "v"(cur[a][pa]));
                return;
            }
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int b = 0; b < WNT; b++) acc[a][b] = io_mfma<IO>(cur[a][pa], fb[b], acc[a][b]);
        };
        const bool mine = !(PASTA_ABLATE & 2) && (TAP == 1 || (TAP == 2 && wave == h_owner));
        float* const sb = TAP == 1 ? mb : hb;
        const int* const sv = TAP == 1 ? m_nvalid : h_nvalid;
        const int sslot = TAP == 1 ? m_slot : h_slot, shalf = TAP == 1 ? half0 : h_half;
        uint32_t q[BPT][4];
        if constexpr (NP == 3) {
            ldb(0); mm(2);                                          // a3 b1
            if (mine) { split_piece(sb, sv, q, true); store_piece(bbuf ^ 1, sslot, shalf, 0, q); }
            mm(1);                                                  // a2 b1
            ldb(1); mm(1);                                          // a2 b2
            if (mine) { split_piece(sb, sv, q, false); store_piece(bbuf ^ 1, sslot, shalf, 1, q); }
            mm(0);                                                  // a1 b2
            ldb(2); mm(0);                                          // a1 b3
            if (mine) { split_piece(sb, sv, q, false); store_piece(bbuf ^ 1, sslot, shalf, 2, q); }
            ldb(0); mm(0);                                          // a1 b1
        }
        if (TAP == 2) __syncthreads();                              // the B image of the next stage is complete, this one is free
    };

    // prologue: stage 0 of this K slice entirely, and the weights of its first tap
    load_b();
    if (PIPE != 3) load_a(0);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < BPT; i++) split_pair(mb, m_nvalid, i, j, msc);
    store_q(0, m_slot, half0);
    if (wave == h_owner) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < BPT; i++) split_pair(hb, h_nvalid, i, j, hsc);
        store_q(0, h_slot, h_half);
    }
    if (PIPE != 3) store_a(0);
    if (PIPE == 1) load_a(1, 1);                     // weights of step 1: stored by step 0
    if (PIPE == 3) load_a_frags(0, ar0);             // fragments of step 0
    __syncthreads();
    if constexpr (PIPE == 3) {
        static_assert(NP == 3, "the weights-in-registers schedule exists for the six-product arithmetic");
        for (int s = 0; s < nstages; s += 2) {
            step_areg(0, 0, ar0, ar1); step_areg(1, 0, ar1, ar0); step_areg(2, 0, ar0, ar1);
            step_areg(0, 1, ar1, ar0); step_areg(1, 1, ar0, ar1); step_areg(2, 1, ar1, ar0);
        }
    } else if constexpr (PIPE == 2) {
        static_assert(NP == 3 || HX, "the register-diet schedule exists for the six- and the three-product arithmetic");
        for (int s = 0; s < nstages; s += 2) {
            step_diet(0, 0, 0); step_diet(1, 1, 0); step_diet(2, 0, 0);
            step_diet(0, 1, 1); step_diet(1, 0, 1); step_diet(2, 1, 1);
        }
    } else
    // two stages (six steps) per trip: the B image alternates per stage, the A image per step; an odd stage count runs
    // one all-zero stage
    for (int s = 0; s < nstages; s += 2) {
        step(0, 0, 0); step(1, 1, 0); step(2, 0, 0);
        step(0, 1, 1); step(1, 0, 1); step(2, 1, 1);
    }

    const int OHW = p.OH * p.OW;
    if constexpr (HX) {
        // back to the operands' units: 1 / S_x for the tile, 1 / S_w per weight row (p.w_rowinv, written by the packing kernel).  The 32 row
        // scales of this lane are fetched in one go in front of the stores (a load in front of every store cost 9 % of the kernel).
        const float* const wri = p.w_rowinv + (int64_t)g * p.Og_pad + o_blk;
        float ws[WMT][16];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) ws[a][r] = wri[(wm * WMT + a) * 32 + acc_row(r, lane)];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int b = 0; b < WNT; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    acc[a][b][r] = (acc[a][b][r] * out_scale) * ws[a][r];
                    if constexpr (PAIR) acc2[a][b][r] = (acc2[a][b][r] * out_scale) * ws[a][r];
                }
    }
    if constexpr (PAIR) {
        // (p, q) of the input lattice -> output row 2p + a, columns 2q and 2q + 1: one 8-byte store per lane
        struct __attribute__((packed, aligned(4))) Pair { float even, odd; };
#pragma unroll
        for (int b = 0; b < WNT; b++) {
            const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
            const int n = (int)(pix / (P * Q));
            const int rem = (int)(pix - (int64_t)n * P * Q);
            const int pp = rem / Q, qq = rem - pp * Q;
            const int64_t yoff = ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + (oy0 + 2 * pp) * p.OW + 2 * qq;
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    if (o < p.Og) {
                        Pair v;
                        v.even = p.pair_bx ? acc2[a][b][r] : acc[a][b][r];
                        v.odd = p.pair_bx ? acc[a][b][r] : acc2[a][b][r];
                        *(Pair*)(p.y + yoff + (int64_t)o * OHW) = v;
                    }
                }
        }
        return;
    }
    const bool has_noise = p.noise && p.ksplit == 1;
    const float nstr = has_noise ? p.noise_strength[0] : 0.f;
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, p.ksplit == 1);
    conv_epilogue_dispatch<(NP == NP_F16X3 || IO != IO_F32)>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
    const bool FULL = full_c;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int64_t pix = pix_blk + (wn * WNT + b) * 32 + jl;
        const int n = (int)(pix / (P * Q));
        const int rem = (int)(pix - (int64_t)n * P * Q);
        const int pp = rem / Q, qq = rem - pp * Q;
        const int plane_off = (oy0 + pp * p.osy) * p.OW + ox0 + qq * p.osx;
        const int64_t yoff = ((int64_t)n * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
        const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n * OHW : 0) + plane_off] * nstr : 0.f;
        float* pb = p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW + yoff : nullptr;      // K slices: fp32 partial sums
        const bool has_res = p.res && p.ksplit == 1;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++) {
            // what the sixteen stores of a 32 x 32 sub-tile need from memory -- output scale, residual, bias -- is fetched in front of them
            // (a load in front of every store serialises on the memory counter: conv_fwd_rows2d_bf16x6.h)
            // (one operand kind at a time through the same sixteen registers: three arrays side by side spill on the three-workgroup tiles)
            float tv[16];
            const bool has_bias = p.act && p.ksplit == 1 && p.bias;
            if (osb) {
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = osb[o < p.Og ? o : p.Og - 1]; }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = fmaf(acc[a][b][r], tv[r], nz);
            } else if (has_noise) {
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += nz;
            }
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    tv[r] = o < p.Og ? io_ld1<IO>((const char*)p.res + (yoff + (int64_t)o * OHW) * ES) : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] += tv[r];
            }
            if (pb) {                                   // K slices: fp32 partial sums, nothing else (scale, residual, bias, activation ride in the reduction)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    if (o < p.Og) pb[(int64_t)o * OHW] = acc[a][b][r];
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane); tv[r] = has_bias ? p.bias[g * p.Og + (o < p.Og ? o : p.Og - 1)] : 0.f; }
            // the stores: instantiated per (activation, clamp, whole tile of rows), no wave-uniform branch per element (conv_common.h)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                const float v = conv_epilogue_c(acc[a][b][r], tv[r], ea, case_c);
                if (FULL || o < p.Og) { io_st<IO>(p.y, yoff + (int64_t)o * OHW, v); amax_take(y_am, v); }
            }
        }
    }
    });
    if (p.ksplit == 1) amax_commit(y_am, y_slot);
}

// Pixel tiles of the row-reuse kernel: full tiles of BN pixels made of whole row segments inside one image.
static bool rows_tile_ok(int P, int Q, int BN) {
    const int seg = Q < BN ? Q : BN;
    return Q % 32 == 0 && (seg & (seg - 1)) == 0 && BN % seg == 0 && Q % seg == 0 && ((int64_t)P * Q) % BN == 0;
}

// The row-reuse kernel of one arithmetic / storage type, if the lattice is made of whole row segments (conv_tu_fwd_rows_*.hip).
template <int BM, int BN, int NP, int IO>
static bool launch_fwd_rows_np(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if (!(q.rows && q.ncls == 1 && rows_tile_ok(q.cls[0].P, q.cls[0].Q, BN))) return false;
    // full tiles made of whole row segments inside one image
    constexpr int APT = (2 * Arith<NP>::npa * BM + 255) / 256;
    constexpr size_t lds = (size_t)(2 * APT * 256 * 8 + 2 * 2 * Arith<NP>::npb * (BN + 16) * 8) * sizeof(__bf16);
    static const int pipe = getenv("PASTA_ROWS_PIPE") ? getenv("PASTA_ROWS_PIPE")[0] - '0' : 1;        // read once (thread-safe initialisation), never written again
    if constexpr (IO == IO_F32 && NP != NP_F16X3)
        PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 0, NP, IO>), lds);
    if constexpr (IO == IO_F32 && NP == 3)
        PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 3, 2, NP, IO>), lds);
    PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 1, NP, IO>), lds);
    if constexpr (IO == IO_F32 && NP == 3) {
        if (pipe == 2) { hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 3, 2, NP, IO>), grid, dim3(256), lds, s, q); return true; }
        if (pipe == 3) {
            constexpr size_t lds3 = (size_t)(2 * 2 * NP * (BN + 16) * 8) * sizeof(__bf16);       // the B images only
            PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 3, NP, IO>), lds3);
            hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 3, NP, IO>), grid, dim3(256), lds3, s, q);
            return true;
        }
    }
    if constexpr (IO == IO_F32 && NP != NP_F16X3) {
        if (pipe == 0) { hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 0, NP, IO>), grid, dim3(256), lds, s, q); return true; }
    }
    hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 1, NP, IO>), grid, dim3(256), lds, s, q);
    return true;
}

// ... and with the input scale in the staging (fp32 storage, fp32-equivalent products: the forward of a modulated convolution;
// other arithmetics keep the separate scaling pass).
template <int BM, int BN, int NP>
static bool launch_fwd_rows_isc(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if (!(q.rows && q.ncls == 1 && rows_tile_ok(q.cls[0].P, q.cls[0].Q, BN))) return false;
    constexpr int APT = (2 * Arith<NP>::npa * BM + 255) / 256;
    constexpr size_t lds = (size_t)(2 * APT * 256 * 8 + 2 * 2 * Arith<NP>::npb * (BN + 16) * 8) * sizeof(__bf16);
    PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 1, NP, IO_F32, true>), lds);
    hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 1, NP, IO_F32, true>), grid, dim3(256), lds, s, q);
    return true;
}

// q.bf16x6 = pieces per operand (3: six products, 2: three, 1: one, NP_F16X3); q.iscale implies fp32 storage and fp32-equivalent products (the caller checked)
template <int BM, int BN>
static bool launch_fwd_rows_any(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if (q.iscale && q.bf16x6 == NP_F16X3) return launch_fwd_rows_isc<BM, BN, NP_F16X3>(q, grid, s);
    if (q.iscale)                  return launch_fwd_rows_isc<BM, BN, 3>(q, grid, s);
    if (q.io == IO_BF16)           return launch_fwd_rows_np<BM, BN, 1, IO_BF16>(q, grid, s);       // 16-bit storage: always one product
    if (q.io == IO_F16)            return launch_fwd_rows_np<BM, BN, 1, IO_F16>(q, grid, s);
    if (q.bf16x6 == 1)             return launch_fwd_rows_np<BM, BN, 1, IO_F32>(q, grid, s);
    if (q.bf16x6 == 2)             return launch_fwd_rows_np<BM, BN, 2, IO_F32>(q, grid, s);
    if (q.bf16x6 == NP_F16X3)      return launch_fwd_rows_np<BM, BN, NP_F16X3, IO_F32>(q, grid, s);
    return launch_fwd_rows_np<BM, BN, 3, IO_F32>(q, grid, s);
}

// The base kernel (conv_tu_fwd_base_*.hip): any lattice, the packed-K mode, the input scale.
template <int BM, int BN, int NP, int IO>
static void launch_fwd_base_np(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if constexpr (IO == IO_F32 && (NP == 3 || NP == NP_F16X3)) {
        if (q.koff) { hipLaunchKernelGGL((conv_fwd_bf16x6_kernel<BM, BN, (BN == 256 ? 2 : 3), NP, IO, false, true>), grid, dim3(256), 0, s, q); return; }
    }
    hipLaunchKernelGGL((conv_fwd_bf16x6_kernel<BM, BN, (BN == 256 ? 2 : 3), NP, IO>), grid, dim3(256), 0, s, q);     // <= 64 KB of LDS: two or three workgroups per CU
}

template <int BM, int BN>
static void launch_fwd_base_any(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if (q.iscale && q.bf16x6 == NP_F16X3) hipLaunchKernelGGL((conv_fwd_bf16x6_kernel<BM, BN, 2, NP_F16X3, IO_F32, true>), grid, dim3(256), 0, s, q);
    else if (q.iscale)             hipLaunchKernelGGL((conv_fwd_bf16x6_kernel<BM, BN, 2, 3, IO_F32, true>), grid, dim3(256), 0, s, q);
    else if (q.io == IO_BF16)      launch_fwd_base_np<BM, BN, 1, IO_BF16>(q, grid, s);
    else if (q.io == IO_F16)       launch_fwd_base_np<BM, BN, 1, IO_F16>(q, grid, s);
    else if (q.bf16x6 == 1)        launch_fwd_base_np<BM, BN, 1, IO_F32>(q, grid, s);
    else if (q.bf16x6 == 2)        launch_fwd_base_np<BM, BN, 2, IO_F32>(q, grid, s);
    else if (q.bf16x6 == NP_F16X3) launch_fwd_base_np<BM, BN, NP_F16X3, IO_F32>(q, grid, s);
    else                           launch_fwd_base_np<BM, BN, 3, IO_F32>(q, grid, s);
}

// Parity-pair launch of the row-reuse kernel (see its PAIR note): p.cls[0..1] are the two vertical parities over the input
// lattice, tap tables per class, p.pair_off / p.pair_bx / p.rows_d0 set by the caller.
template <int BM, int BN, int NP = 3>
static void launch_fwd_pair(const ConvFwdParams& p, hipStream_t s) {
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    const int64_t tiles = (int64_t)p.N * p.cls[0].P * p.cls[0].Q / BN;
    dim3 grid((unsigned)(2 * tiles), q.o_tiles, p.G);
    constexpr int APT = (2 * Arith<NP>::npa * BM + 255) / 256;
    constexpr size_t lds = (size_t)(2 * APT * 256 * 8 + 2 * 2 * Arith<NP>::npb * (BN + 16) * 8) * sizeof(__bf16);
    PASTA_SET_LDS((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 2, NP, IO_F32, false, true>), lds);
    hipLaunchKernelGGL((conv_fwd_rows_bf16x6_kernel<BM, BN, 2, 2, NP, IO_F32, false, true>), grid, dim3(256), lds, s, q);
}

// Do the T taps at table positions [0, T) form rows of three horizontally adjacent offsets (ascending or descending)?
static bool detect_tap_rows(ConvFwdParams& p, int T) {
    p.rows = 0;
    if (T % 3 != 0 || p.isx != 1 || p.isy != 1 || p.osx != 1 || p.osy != 1) return false;
    const int step = p.tap_dx[1] - p.tap_dx[0];
    if (step != 1 && step != -1) return false;
    const int d0 = step == 1 ? p.tap_dx[0] : p.tap_dx[2];
    for (int j = 0; j < T; j += 3)
        for (int i = 0; i < 3; i++)
            if (p.tap_dy[j + i] != p.tap_dy[j] || p.tap_dx[j + i] != p.tap_dx[0] + i * step) return false;
    p.rows = 1; p.rows_d0 = d0; p.rows_rev = step == -1 ? 1 : 0;
    return true;
}


}  // namespace pasta
