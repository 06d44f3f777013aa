// 3x3 stride-2 forward convolutions of the three-product fp16 arithmetic (the down path of the discriminator and of the
// encoders: conv2d_resample.py:119-122 after the blur; 9.9 ms of the training step at 136 TFLOP/s on the one-tap base kernel,
// profiles/r4_byshape_classes.txt).  Instantiated by conv_tu_fwd_small.hip (conv_launch.h).
#pragma once
#include "conv_fwd_bf16x6.h"

namespace pasta {

//------------------------------------------------------------------------------------
// conv_fwd_bf16x6_kernel stages every tap on its own: for a stride-2 lattice each lane gathers every second pixel of an input row
// (half of every 64-byte sector fetched is dropped) and the staged image serves 12 MFMAs per wave.  Stride 2 has a quarter of the
// reuse of stride 1 (an output pixel's 3 x 3 window overlaps its neighbour's in one column only), so the 2-D tiles of
// conv_fwd_rows2d_bf16x6_kernel do not carry over (a de-interleaved (2R + 1) x (2 SEG + 1) footprint does not fit LDS with double
// buffering) -- but the three taps of a kernel ROW do share their data: with E[j] = X[2 (q0 + j) - pad] and
// O[j] = X[2 (q0 + j) + 1 - pad] the taps dx = 0, 1, 2 of output pixel q0 + j read E[j], O[j], E[j + 1].  Here:
//   * a round = (kernel row dy, 16-channel chunk): the input row segments of the tile -- 2 SEG + 1 CONSECUTIVE pixels per tile row, fetched
//     by consecutive lanes (whole sectors) -- are split once and written to the E or the O image by the parity of the pixel; the three taps
//     of the row read their fragments there: 36 MFMAs per wave and round, two staging tasks per thread where the base kernel needs three;
//   * one LDS buffer, two barriers per round, the next round's fetches (activations and the three taps' weights) in flight during the
//     MFMAs, two workgroups per CU: while one stages the other multiplies (the scheme of conv1x1_f16x3_kernel);
//   * tile = BM output channels x 128 output pixels = R = 128 / SEG rows of SEG = min(OW, 128) columns.
// Weights: the standard packed layout, one scale per output row (p.w_rowinv).  Planes: OW a power of two >= 16 (the live shapes: 128, 64, 32, 16).
// XP (round 5): x is the producer-written operand (PASTA_LAYOUT_PIECES16, pieces.hip: [N][C / 8][H][2][W] units of 16 bytes, fp16 h[8] / l'[8] of v S with
// S from p.x_amax, the producer's bound row) -- a staging task is then two sixteen-byte loads and two LDS stores: no channel-strided dword gathers,
// no split (the eight loads and ~25 VALU instructions of a task were a third of this kernel's issue slots: profiles/r4_pmc_summary.txt, issue 0.35).
template <int BM, bool XP = false>
__global__ __launch_bounds__(256, 2) void conv3x3s2_f16x3_kernel(ConvFwdParams p) {
    constexpr int NP = NP_F16X3, BN = 128;
    constexpr int WMT = 2, WNT = 2;
    constexpr int WAVES_N = BM == 128 ? 2 : 4;          // BM 128: waves 2 x 2 of 64 x 64; BM 64: waves 1 x 4 of 64 x 32
    constexpr int WN_PIX = BN / WAVES_N;                // pixels per wave: 64 or 32
    constexpr int WNT_ = WN_PIX / 32;                   // 32-pixel MFMA tiles per wave: 2 or 1
    static_assert((BM / 64) * WAVES_N == 4 && WNT_ >= 1, "four waves per workgroup");
    constexpr int MAXR = 8;                             // tile rows (OW >= 16)
    constexpr int SLOTS = 2 * BN + MAXR;                // E image (BN + R slots) then O image (BN slots)
    constexpr int AUNITS = 3 * 6 * BM, APT = (AUNITS + 255) / 256;  // sixteen-byte units of a round's weights: 3 taps x 3 pieces x 2 halves x BM rows -> 9 or 4.5 per thread
    static_assert(APT == 9 || APT == 5, "nine or five units per thread (the last one of BM = 64 repeats the final unit in half of the threads)");
    __shared__ __attribute__((aligned(16))) __bf16 As[3 * 6 * BM * 8];          // [tap dx][piece * 2 + half][row][8]
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2 * 2 * SLOTS * 8];       // [piece][half][slot][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int o_blk = blockIdx.y * BM;
    const int OW = p.OW, OH = p.OH;
    const int seg_log2 = 31 - __builtin_clz(OW < BN ? OW : BN);
    const int SEG = 1 << seg_log2, R = BN >> seg_log2;
    const int cblocks = OW >> seg_log2, tpi = (OH / R) * cblocks;
    const int n_img = blockIdx.x / tpi, t_in = blockIdx.x - n_img * tpi;
    const int p0 = (t_in / cblocks) * R, q0 = (t_in % cblocks) << seg_log2;
    const int pad = -p.tap_dy[0];                       // 0 or 1 (tap (r, c) reads input (2 p + r - pad, 2 q + c - pad))
    const int NC = p.Ig_pad / 16, rounds = 3 * NC;
    const int ESLOTS = BN + R, NSLOT = 2 * BN + R, NTASK = 2 * NSLOT;      // tasks = (slot, k-half)

    float sx, isx;
    scale_from_amax(amax_of_parts(p.x_amax), sx, isx);

    // ---- this thread's staging tasks u = tid + 256 k (k = 0, 1 for every thread; k = 2 for the first NTASK - 512 threads)
    int t_lds[3], t_half[3], t_iy0[3], t_ix[3];
    bool t_on[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int u = tid + 256 * k;
        t_on[k] = u < NTASK;
        const int uu = t_on[k] ? u : 0;
        const int half = uu >= NSLOT ? 1 : 0, s = uu - half * NSLOT;
        int r, x;
        if (s < ESLOTS) { r = s / (SEG + 1); x = 2 * (s - r * (SEG + 1)); }
        else { const int so = s - ESLOTS; r = so >> seg_log2; x = 2 * (so & (SEG - 1)) + 1; }
        t_half[k] = half;
        t_lds[k] = (half * SLOTS + s) * 8;
        t_iy0[k] = 2 * (p0 + r) - pad;
        t_ix[k] = 2 * q0 + x - pad;
    }
    const float* const xb = p.x + (int64_t)n_img * p.Cin * p.H * p.W;
    const int HW = p.H * p.W;
    constexpr int XV = XP ? 1 : 8;
    float xv[3][XV];
    u32x4 xq[3][XP ? 2 : 1];                            // XP: the unit's h and l' pieces as they lie
    int xvalid[3];                                      // channels of the task that exist and lie inside the plane (0: a zero pixel)
    auto load_x = [&](int rnd) {
        const int dy = rnd / NC, cc = rnd - dy * NC;    // scalar
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k == 2 && !t_on[2]) continue;           // the third task exists for the first few threads only
            const int iy = t_iy0[k] + dy;
            const bool in = t_on[k] && (unsigned)iy < (unsigned)p.H && (unsigned)t_ix[k] < (unsigned)p.W;
            const int off = in ? iy * p.W + t_ix[k] : 0;
            const int c0 = cc * 16 + t_half[k] * 8;
            xvalid[k] = in ? p.Ig - c0 : 0;
            if constexpr (XP) {
                const int c8 = c0 < p.Ig ? (c0 >> 3) : 0;                   // Ig is a multiple of 8: an octet exists whole or not at all
                // [N][C / 8][H][2 pieces][W] units of sixteen bytes: consecutive lanes (pixels of a row) fetch consecutive units
                const int offp = in ? 2 * iy * p.W + t_ix[k] : 0;
                const u32x4* const src = (const u32x4*)((const char*)p.x + ((int64_t)n_img * (p.Cin >> 3) + c8) * HW * 32) + offp;
                xq[k][0] = src[0]; xq[k][1] = src[p.W];
                continue;
            }
#pragma unroll
            for (int j = 0; j < XV; j++) {
                const int c = c0 + j < p.Ig ? c0 + j : p.Ig - 1;
                xv[k][j] = xb[(int64_t)c * HW + off];
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k == 2 && !t_on[2]) continue;
            if constexpr (XP) {
                const bool on = xvalid[k] > 0;
                const u32x4 z = {0u, 0u, 0u, 0u};
                __bf16* const d = &Bs[t_lds[k]];
                *(u32x4*)d = on ? xq[k][0] : z;
                *(u32x4*)(d + 2 * SLOTS * 8) = on ? xq[k][1] : z;
                continue;
            }
            uint32_t h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float v0 = xv[k][(2 * q) % XV], v1 = xv[k][(2 * q + 1) % XV];
                if (xvalid[k] < 8) { v0 = 2 * q < xvalid[k] ? v0 : 0.f; v1 = 2 * q + 1 < xvalid[k] ? v1 : 0.f; }
                f16_split2(v0 * sx, v1 * sx, h[q], l[q]);
            }
            __bf16* const d = &Bs[t_lds[k]];
            *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
            *(uint4*)(d + 2 * SLOTS * 8) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    // ---- weights of a round: taps (dy, 0..2) of chunk cc; unit u -> (tap dx, segment, row)
    float4 av0, av1, av2, av3, av4, av5, av6, av7, av8;          // APT of them are used (scalars: an array here is not kept in registers)
    auto a_unit = [&](int k, int rnd) -> float4 {
        const int dy = rnd / NC, cc = rnd - dy * NC;
        int u = tid + 256 * k;
        if (256 * (k + 1) > AUNITS) u = u < AUNITS ? u : AUNITS - 1;      // BM 64: 1152 units = 4.5 per thread
        const int dx = u / (6 * BM), rem = u - dx * 6 * BM;
        const int sg = rem / BM, row = rem - sg * BM;
        const int slab = dy * 3 + dx;                   // conv2d: the packed tap order is the kernel's own (a flip is applied by the packing kernel)
        return *(const float4*)((const __bf16*)p.wp + (((int64_t)slab * NC + cc) * 6 * p.Og_pad + (int64_t)sg * p.Og_pad + o_blk + row) * 8);
    };
    auto a_store = [&](int k, float4 v) {
        int u = tid + 256 * k;
        if (256 * (k + 1) > AUNITS) u = u < AUNITS ? u : AUNITS - 1;
        *(float4*)&As[u * 8] = v;                       // the unit order IS the LDS order: [dx][segment][row]
    };
    auto load_a = [&](int rnd) {
        av0 = a_unit(0, rnd); av1 = a_unit(1, rnd); av2 = a_unit(2, rnd); av3 = a_unit(3, rnd);
        if constexpr (APT >= 5) av4 = a_unit(4, rnd);
        if constexpr (APT >= 9) { av5 = a_unit(5, rnd); av6 = a_unit(6, rnd); av7 = a_unit(7, rnd); av8 = a_unit(8, rnd); }
    };
    auto store_a = [&]() {
        a_store(0, av0); a_store(1, av1); a_store(2, av2); a_store(3, av3);
        if constexpr (APT >= 5) a_store(4, av4);
        if constexpr (APT >= 9) { a_store(5, av5); a_store(6, av6); a_store(7, av7); a_store(8, av8); }
    };

    f32x16 acc[WMT][WNT_];
#pragma unroll
    for (int a = 0; a < WMT; a++)
#pragma unroll
        for (int b = 0; b < WNT_; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int hl = lane >> 5, jl = lane & 31;
    uint32_t y_am = 0;
    const AmaxSlot y_slot = amax_begin(p.y_amax);
    // slots of this lane's pixel of B fragment b: E image (tap dx = 0; dx = 2: the next slot) and O image
    int eslot[WNT_], oslot[WNT_];
#pragma unroll
    for (int b = 0; b < WNT_; b++) {
        const int t = wn * WN_PIX + b * 32 + jl;
        eslot[b] = t + (t >> seg_log2);                 // r (SEG + 1) + c
        oslot[b] = ESLOTS + t;
    }

    load_x(0);
    load_a(0);
    for (int rnd = 0; rnd < rounds; rnd++) {
        store_x();
        store_a();
        __syncthreads();
        const int rn = rnd + 1 < rounds ? rnd + 1 : rnd;        // the last round re-reads itself (static control flow around the loads)
        load_x(rn);
        load_a(rn);
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            bf16x8 fa[WMT][3], fb[WNT_][2];
#pragma unroll
            for (int pc = 0; pc < 3; pc++)
#pragma unroll
                for (int a = 0; a < WMT; a++) fa[a][pc] = *(const bf16x8*)&As[(((dx * 3 + pc) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8];
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int b = 0; b < WNT_; b++) {
                    const int slot = dx == 1 ? oslot[b] : eslot[b] + (dx >> 1);
                    fb[b][pc] = *(const bf16x8*)&Bs[((pc * 2 + hl) * SLOTS + slot) * 8];
                }
#define PASTA_MM2(PA, PB)                                                                                          \
            _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT_; b++)         \
                acc[a][b] = mfma16<IO_F32, NP>(fa[a][PA], fb[b][PB], acc[a][b]);
            PASTA_MM2(2, 1)     // h'' l', l h, h h: smallest terms first
            PASTA_MM2(1, 0)
            PASTA_MM2(0, 0)
#undef PASTA_MM2
        }
        __syncthreads();
    }

    {   // back to the operands' units
        const float* const wri = p.w_rowinv + o_blk;
        float ws[WMT][16];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) ws[a][r] = wri[(wm * WMT + a) * 32 + acc_row(r, lane)];
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int b = 0; b < WNT_; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = (acc[a][b][r] * isx) * ws[a][r];
    }
    const int OHW = OH * OW;
    // the stores: instantiated per (activation, clamp, whole tile of rows) and chosen once per workgroup (conv_common.h)
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, true);
    conv_epilogue_dispatch<true>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
        const bool FULL = full_c;
#pragma unroll
        for (int b = 0; b < WNT_; b++) {
            const int t = wn * WN_PIX + b * 32 + jl;
            const int r = t >> seg_log2, c = t & (SEG - 1);
            const int64_t yoff = (int64_t)n_img * p.Cout * OHW + (p0 + r) * OW + q0 + c;
#pragma unroll
            for (int a = 0; a < WMT; a++) {
                float tv[16];                           // residual, then bias, through the same registers
                if (p.res) {
#pragma unroll
                    for (int rr = 0; rr < 16; rr++) {
                        const int o = o_blk + (wm * WMT + a) * 32 + acc_row(rr, lane);
                        tv[rr] = (FULL || o < p.Og) ? p.res[yoff + (int64_t)o * OHW] : 0.f;
                    }
#pragma unroll
                    for (int rr = 0; rr < 16; rr++) acc[a][b][rr] += tv[rr];
                }
#pragma unroll
                for (int rr = 0; rr < 16; rr++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(rr, lane);
                    tv[rr] = (ea.on && p.bias) ? p.bias[(FULL || o < p.Og) ? o : p.Og - 1] : 0.f;
                }
#pragma unroll
                for (int rr = 0; rr < 16; rr++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(rr, lane);
                    const float v = conv_epilogue_c(acc[a][b][rr], tv[rr], ea, case_c);
                    if (FULL || o < p.Og) { p.y[yoff + (int64_t)o * OHW] = v; amax_take(y_am, v); }
                }
            }
        }
    });
    amax_commit(y_am, y_slot);
}

// Does the stride-2 kernel take this launch?  (conv2d, 3x3, stride 2, equal pads of 0 or 1, three-product arithmetic on fp32 tensors, one
// group, nothing riding along; output planes whose width is a power of two >= 16 and that divide into 128-pixel tiles of whole rows.)
static bool conv3x3s2_shape_ok(int OH, int OW) {
    if (OW < 16 || (OW & (OW - 1))) return false;
    const int seg = OW < 128 ? OW : 128, R = 128 / seg;
    return OH % R == 0;
}
static bool conv3x3s2_ok(const ConvFwdParams& p, int kh, int kw, int stride, int pad_h, int pad_w, int transposed) {
    static const bool enabled = !(getenv("PASTA_CONV_S2") && getenv("PASTA_CONV_S2")[0] == '0');         // A/B switch
    if (!enabled || transposed || p.bf16x6 != NP_F16X3 || p.io != IO_F32 || p.G != 1 || kh != 3 || kw != 3 || stride != 2 || pad_h != pad_w || pad_h > 1) return false;
    if (p.iscale || p.oscale || p.noise || p.ksplit != 1 || p.koff || p.x2 || p.Ig < 16 || p.Og <= 32) return false;
    if (p.x_pieces && ((p.Ig & 7) || pad_h != 0)) return false;         // the producer's units hold eight channels; the blur absorbs the padding
    return conv3x3s2_shape_ok(p.OH, p.OW);
}

}  // namespace pasta
