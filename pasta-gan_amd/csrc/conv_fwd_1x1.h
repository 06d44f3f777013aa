// Pointwise (1x1, stride 1) convolutions of the three-product fp16 arithmetic, optionally over the channel concatenation of TWO
// tensors that is never formed.  Instantiated by conv_tu_fwd_small.hip (conv_launch.h).
#pragma once
#include "conv_fwd_bf16x6.h"

namespace pasta {

//------------------------------------------------------------------------------------
// Y[n][o][pix] = sum_c W[o][c] X[n][c][pix]: a plain GEMM whose B operand is pixel-contiguous in HBM, and -- at the widths of this
// model (64 .. 576 input channels into 64 .. 512 outputs over 32^2 .. 256^2 planes) -- a BANDWIDTH-bound one: 128 -> 64 at 256^2 moves
// 805 MB for 17 GFLOP (134 us at 6 TB/s against 52 us of matrix work at this family's rate).  conv_fwd_bf16x6_kernel treats it as a
// one-tap convolution: every thread gathers its pixel's 8 channels with eight 4-byte loads per 16-channel chunk and 12 MFMAs, and
// reaches 0.45 of the HBM roofline on the 64-channel outputs (r3_by_shape.txt: 58 TFLOP/s, 0.295 ms).  Here:
//   * loads are 16 (BN = 256) or 8 (BN = 128) bytes per lane ALONG the pixels -- a wave reads 1 KB / 512 B contiguous per channel --
//     eight channels per thread and round, all in flight at once (a wave owns one channel octet of the round: scalar channel offsets);
//   * a round is 32 channels (two MFMA K steps): 24 MFMAs per wave between barriers;
//   * one LDS buffer, two barriers per round, the next round's loads in flight during the MFMAs; 40 / 44 KB of LDS and <= 170 VGPRs
//     keep three workgroups per CU resident: the latency is hidden by occupancy, not by a schedule (the matrix pipes are a fifth busy);
//   * TWO input tensors (p.x: channels [0, C1), p.x2: channels [C1, C_in)) serve `conv1x1(cat([x, x2], 1))` -- the merge layers of the
//     synthesis blocks, networks.py:5698-5700 -- without the concatenation pass, without its scan, and the layer's two input gradients
//     come back as two contiguous tensors (two launches of this kernel on weight slices) instead of channel slices of one.
// Weights: the standard packed layout of pack_weights_f16x3_kernel (one tap), rows scaled one by one (p.w_rowinv).
// Tiles: BM = 64 rows x BN = 256 pixels (waves 1 x 4) or 128 x 128 (waves 2 x 2); every wave a 64 x 64 sub-tile, as everywhere.
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv1x1_f16x3_kernel(ConvFwdParams p) {
    constexpr int NP = NP_F16X3;
    constexpr int WMT = 2, WNT = 2;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "four waves per workgroup");
    constexpr int PX = BN / 64;                         // consecutive pixels per lane: 4 (16-byte loads) or 2 (8-byte loads)
    constexpr int KS = 32, NOCT = KS / 8;               // channels per round; octets = the four waves' staging roles
    constexpr int AUNITS = 2 * 6 * BM, APT = AUNITS / 256;      // sixteen-byte units of a round's weights (2 chunks x 3 pieces x 2 halves x BM rows)
    static_assert(AUNITS % 256 == 0 && (APT == 3 || APT == 6), "three or six whole units per thread");
    __shared__ __attribute__((aligned(16))) __bf16 As[3 * NOCT * BM * 8];       // [piece][octet][row][8]
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2 * NOCT * BN * 8];       // [piece][octet][pixel][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int o_blk = blockIdx.y * BM;
    const int HW = p.H * p.W;
    const int64_t pix_blk = (int64_t)blockIdx.x * BN;   // the host guarantees HW % BN == 0: a tile lies inside one sample
    const int n_img = (int)(pix_blk / HW);
    const int poff = (int)(pix_blk - (int64_t)n_img * HW) + lane * PX;
    const int C1 = p.x2 ? p.C1 : p.Cin, C2 = p.Cin - C1;
    const int rounds = (p.Cin + KS - 1) / KS;
    const int NC = p.Ig_pad / 16;

    float sx, isx;
    {
        float am = amax_of_parts(p.x_amax);
        if (p.x2) am = fmaxf(am, amax_of_parts(p.x2_amax));      // one accumulator set: one scale for both operands
        scale_from_amax(am, sx, isx);
    }

    typedef float xvec __attribute__((ext_vector_type(PX)));
    xvec xv[8];                                         // this thread's 8 channels x PX pixels of the round in flight
    int xvalid = 0;                                     // how many of the 8 channels exist (the tail of C_in)
    auto load_x = [&](int r) {
        const int c0 = r * KS + wave * 8;               // scalar: the wave's octet
        xvalid = p.Cin - c0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int c = c0 + j < p.Cin ? c0 + j : p.Cin - 1;         // past the end: a valid address, masked below
            const bool second = c >= C1;                         // scalar
            const float* base = second ? p.x2 : p.x;
            const int ct = second ? C2 : C1;
            c = second ? c - C1 : c;
            xv[j] = *(const xvec*)(base + ((int64_t)n_img * ct + c) * HW + poff);
        }
    };
    float4 av0, av1, av2, av3, av4, av5;                // APT of them are used (scalars: an array here is not kept in registers)
    auto a_unit = [&](int k, int r) -> float4 {         // unit -> (chunk of the round, segment = piece * 2 + half, row)
        const int u = tid + 256 * k;
        const int ch = u / (6 * BM), rem = u - ch * 6 * BM;
        const int seg = rem / BM, row = rem - seg * BM;
        int cc = r * 2 + ch;
        cc = cc < NC ? cc : NC - 1;                     // an odd number of 16-channel chunks: the second half of the last round meets zero activations
        return *(const float4*)((const __bf16*)p.wp + ((int64_t)cc * 6 * p.Og_pad + (int64_t)seg * p.Og_pad + o_blk + row) * 8);
    };
    auto load_a = [&](int r) {
        av0 = a_unit(0, r); av1 = a_unit(1, r); av2 = a_unit(2, r);
        if constexpr (APT > 3) { av3 = a_unit(3, r); av4 = a_unit(4, r); av5 = a_unit(5, r); }
    };
    auto a_store = [&](int k, float4 v) {
        const int u = tid + 256 * k;
        const int ch = u / (6 * BM), rem = u - ch * 6 * BM;
        const int seg = rem / BM, row = rem - seg * BM;
        const int piece = seg >> 1, half = seg & 1;
        *(float4*)&As[((piece * NOCT + ch * 2 + half) * BM + row) * 8] = v;
    };
    auto store_a = [&]() {
        a_store(0, av0); a_store(1, av1); a_store(2, av2);
        if constexpr (APT > 3) { a_store(3, av3); a_store(4, av4); a_store(5, av5); }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int px = 0; px < PX; px++) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float v0 = xv[2 * k][px], v1 = xv[2 * k + 1][px];
                if (xvalid < 8) { v0 = 2 * k < xvalid ? v0 : 0.f; v1 = 2 * k + 1 < xvalid ? v1 : 0.f; }
                f16_split2(v0 * sx, v1 * sx, h[k], l[k]);
            }
            __bf16* const d = &Bs[(wave * BN + lane * PX + px) * 8];
            *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
            *(uint4*)(d + NOCT * BN * 8) = make_uint4(l[0], l[1], l[2], l[3]);
        }
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
    const AmaxSlot y_slot = amax_begin(p.y_amax);

    load_x(0);
    load_a(0);
    for (int r = 0; r < rounds; r++) {
        store_x();
        store_a();
        __syncthreads();
        const int rn = r + 1 < rounds ? r + 1 : r;      // the last round re-reads itself (static control flow around the loads)
        load_x(rn);
        load_a(rn);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fa[WMT][3], fb[WNT][2];
#pragma unroll
            for (int pc = 0; pc < 3; pc++)
#pragma unroll
                for (int a = 0; a < WMT; a++) fa[a][pc] = *(const bf16x8*)&As[((pc * NOCT + ks * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8];
#pragma unroll
            for (int pc = 0; pc < 2; pc++)
#pragma unroll
                for (int b = 0; b < WNT; b++) fb[b][pc] = *(const bf16x8*)&Bs[((pc * NOCT + ks * 2 + hl) * BN + (wn * WNT + b) * 32 + jl) * 8];
#define PASTA_MM1(PA, PB)                                                                                          \
            _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
                acc[a][b] = mfma16<IO_F32, NP>(fa[a][PA], fb[b][PB], acc[a][b]);
            PASTA_MM1(2, 1)     // h'' l', l h, h h: smallest terms first
            PASTA_MM1(1, 0)
            PASTA_MM1(0, 0)
#undef PASTA_MM1
        }
        __syncthreads();
    }

    // back to the operands' units, then the epilogue of the forward-type kernels (residual, bias, activation, gain, clamp)
    {
        const float* const wri = p.w_rowinv + o_blk;
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
                for (int r = 0; r < 16; r++) acc[a][b][r] = (acc[a][b][r] * isx) * ws[a][r];
    }
    const int64_t ybase = (int64_t)n_img * p.Cout * HW + (pix_blk - (int64_t)n_img * HW);
    // bias per output row and the residual of a 32 x 32 sub-tile are fetched in front of the stores (a load in front of every store serialises on
    // the memory counter: conv_fwd_rows2d_bf16x6.h)
    // the stores: instantiated per (activation, clamp, whole tile of rows) and chosen once per workgroup (conv_common.h)
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, true);
    conv_epilogue_dispatch<true>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
        const bool FULL = full_c;
#pragma unroll
        for (int b = 0; b < WNT; b++) {
            const int64_t yoff = ybase + (wn * WNT + b) * 32 + jl;
#pragma unroll
            for (int a = 0; a < WMT; a++) {
                float tv[16];                           // residual, then bias, through the same registers: sixteen loads in a row, then their use
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                        tv[r] = (FULL || o < p.Og) ? p.res[yoff + (int64_t)o * HW] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[a][b][r] += tv[r];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    tv[r] = (ea.on && p.bias) ? p.bias[(FULL || o < p.Og) ? o : p.Og - 1] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r, lane);
                    const float v = conv_epilogue_c(acc[a][b][r], tv[r], ea, case_c);
                    if (FULL || o < p.Og) { p.y[yoff + (int64_t)o * HW] = v; amax_take(y_am, v); }
                }
            }
        }
    });
    amax_commit(y_am, y_slot);
}

// Does the pointwise kernel take this launch?  Three-product arithmetic on fp32 tensors, one group, no scales / noise riding along,
// at least 16 input channels into more than 32 outputs, planes that divide into the pixel tiles.
static bool conv1x1_ok(const ConvFwdParams& p, int kh, int kw, int stride, int pad_h, int pad_w) {
    static const bool enabled = !(getenv("PASTA_CONV1X1") && getenv("PASTA_CONV1X1")[0] == '0');       // A/B switch
    if (!enabled || p.bf16x6 != NP_F16X3 || p.io != IO_F32 || p.G != 1 || kh != 1 || kw != 1 || stride != 1 || pad_h || pad_w) return false;
    if (p.iscale || p.oscale || p.noise || p.ksplit != 1 || p.koff || p.Ig < 16 || p.Og <= 32) return false;
    if (p.OH != p.H || p.OW != p.W) return false;
    const int bn = p.Og <= 64 ? 256 : 128;
    return ((int64_t)p.H * p.W) % bn == 0;
}

}  // namespace pasta
