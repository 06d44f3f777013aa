// Split-bf16 weight-gradient kernels for 3x3 convolutions: stride 1 (conv_wgrad3x3_bf16x6_kernel) and stride 2 (conv_wgrad3x3s2_bf16x6_kernel).  Instantiated by conv_tu_wgrad_{3x3,3x3s2,1x1}.hip (conv_launch.h).
#pragma once
#include "conv_common.h"

namespace pasta {

// Workgroup -> (K slice, b tile, a tile, group).  The workgroups that read the same pixels are the tiles (a, b) of one K slice.
// Order 0 (the original): K slice fastest -- those workgroups are ksplit apart in the grid (on one XCD when ksplit is a multiple
// of 8, but dispatched far from each other).  Order 1: the slice index modulo 8 fastest, then the tile, then the rest of the slice
// index: the tiles of a slice are 8 apart -- the same XCD (workgroups go round-robin to the eight XCDs), dispatched together.
__device__ __forceinline__ void wgrad_decode(const WgradParams& p, int& ks, int& bt, int& at, int& g) {
    int bid = blockIdx.x;
    if (p.xcd_order && (p.ksplit & 7) == 0) {
        const int lo = bid & 7; bid >>= 3;
        const int tiles = p.a_tiles * p.b_tiles;
        const int tile = bid % tiles; bid /= tiles;
        const int hi = bid % (p.ksplit >> 3); bid /= (p.ksplit >> 3);
        ks = hi * 8 + lo; bt = tile % p.b_tiles; at = tile / p.b_tiles; g = bid;
        return;
    }
    ks = bid % p.ksplit; bid /= p.ksplit;
    bt = bid % p.b_tiles; bid /= p.b_tiles;
    at = bid % p.a_tiles; bid /= p.a_tiles;
    g = bid;
}


//------------------------------------------------------------------------------------
// Weight gradient of 3x3 / stride-1 / pad-1 convolutions on the bf16 matrix cores with split-bf16 products
// (same arithmetic as conv_fwd_bf16x6_kernel: three bf16 pieces per fp32 operand, six exact products, fp32
// accumulate).  K = pixels: a chunk is 32 consecutive pixels of one image row (Q % 32 == 0), two K steps of 16.
//   A operand: S pieces in LDS as [piece][a][32 px] (row pitch 40 bf16 = 80 B: conflict-free 16-byte reads)
//   B operand: L halo pieces as [piece][b][3 rows][40 px], halo column 0 = image column q0 - 4, so every global and
//              LDS access is 16-byte aligned; the window of tap column ts starts at halo column 3 + ts: it is cut out
//              of two aligned 16-byte blocks with v_alignbit (ts = 0, 2) or by register renaming (ts = 1).
// One wave owns a 32 x 32 (a, b) tile for all 9 taps (144 accumulator registers): 54 MFMAs per K step.

template <int NP, int IO = IO_F32>       // bf16 pieces per operand: 3 (six products), 2 (three), 1 (one); IO: storage type of S and L (conv_common.h)
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_bf16x6_kernel(WgradParams p) {
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: one product");
    constexpr bool HX = Arith<NP>::f16x3;           // PASTA_MATH_F16X3: both operands as (h, l), three products (conv_common.h)
    constexpr int NPW = Arith<NP>::npw;             // pieces per operand in LDS
    constexpr int ES = io_size<IO>::value;
    constexpr int SP = 40, LP = 40;                 // row pitches in bf16 elements (80 B)
    constexpr int S_PIECE = 64 * SP;                // one piece of the S tile
    constexpr int L_PIECE = 64 * 3 * LP;            // one piece of the L halo tile
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    __bf16* Ss = smem16;                            // [3][64][SP]
    __bf16* Ls = smem16 + NPW * S_PIECE;            // [NPW][64][3][LP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;
    float s_scale = 1.f, l_scale = 1.f, out_scale = 1.f;       // PASTA_MATH_F16X3: operand scales from the tensors' partial maxima
    if constexpr (HX) {
        float is_, il_;
        scale_from_amax(amax_of_parts(p.s_amax), s_scale, is_);
        scale_from_amax(amax_of_parts(p.l_amax), l_scale, il_);
        out_scale = is_ * il_;
    }

    int ks, bt, at, g;
    wgrad_decode(p, ks, bt, at, g);
    const int a_blk = at * 64, b_blk = bt * 64;
    const int PQ = p.P * p.Q;
    const char* const Sg = (const char*)p.S + ((int64_t)g * p.Ag + a_blk) * PQ * ES;
    const char* const Lg = (const char*)p.L + ((int64_t)g * p.Bg + b_blk) * PQ * ES;       // LH == P, LW == Q for this kernel

    // Chunk order: down a 32-pixel column block of one image (chunk -> n, column block qb, row pp, pp fastest).  The L halo
    // of chunk (pp) is image rows pp-1 .. pp+1; the three rows live in a ring in LDS (row y in slot y mod 3), so going from pp
    // to pp+1 fetches, splits and stores ONE new row (pp+2's predecessor pp+1 is already there) instead of three: a third of
    // the L traffic and of the staging work of a row-major order.  Rows pp-1 and pp are primed where a column block or a
    // K slice begins.
    // staging roles (fixed): S unit = (channel a, group of 8 pixels); L row units = (channel b, group of 8 columns): 320 per row
    const int s_a = tid >> 2, s_grp = tid & 3;
    const bool s_ch_ok = a_blk + s_a < p.Ag;
    const int l_b0 = tid / 5, l_g0 = tid - l_b0 * 5;                    // unit tid
    const int l_b1 = (256 + tid) / 5, l_g1 = (256 + tid) - l_b1 * 5;    // unit 256 + tid (threads 0..63)
    const bool l_ok0 = b_blk + l_b0 < p.Bg, l_ok1 = tid < 64 && b_blk + l_b1 < p.Bg;

    float4 sreg[2], lreg[2][2];
    unsigned vmask = 0;                              // validity of the 6 sixteen-byte halves held in registers
    auto decode = [&](int ch, int& n, int& qb, int& pp) {
        const int per_img = p.P * p.qblocks;
        n = ch / per_img;
        const int rem = ch - n * per_img;
        qb = rem / p.P; pp = rem - qb * p.P;
    };
    // one L row (image row ly of sample n, columns q0 - 4 .. q0 + 36) into r[0..1]; returns the validity bits of its four halves
    // (no control flow around the loads: every load is issued, from the tensor's first element where its four-pack lies outside the plane or the
    // channel does not exist, and the validity bits zero it at the split -- the branches this was written with, one per four-pack, made the
    // compiler wait for ALL outstanding loads at each of them: round 5's census of the staging code)
    auto fetch_row = [&](int n, int q0, int ly, float4 (&r)[2][2]) -> unsigned {
        unsigned m = 0;
        const bool rowok = (unsigned)ly < (unsigned)p.P;
        {
            const int lx = q0 - 4 + 8 * l_g0;
            const char* lp = Lg + ((int64_t)n * p.LC * PQ + (int64_t)l_b0 * PQ + ly * p.Q + lx) * ES;
            const bool v0 = rowok && l_ok0 && lx >= 0 && lx + 4 <= p.Q, v1 = rowok && l_ok0 && lx + 4 >= 0 && lx + 8 <= p.Q;
            r[0][0] = io_ld4<IO>(v0 ? lp : Lg); r[0][1] = io_ld4<IO>(v1 ? lp + 4 * ES : Lg);
            m |= (v0 ? 1u : 0u) | (v1 ? 2u : 0u);
        }
        if (tid < 64) {                              // the second unit exists for the first wave only (wave-uniform)
            const int lx = q0 - 4 + 8 * l_g1;
            const char* lp = Lg + ((int64_t)n * p.LC * PQ + (int64_t)l_b1 * PQ + ly * p.Q + lx) * ES;
            const bool v0 = rowok && l_ok1 && lx >= 0 && lx + 4 <= p.Q, v1 = rowok && l_ok1 && lx + 4 >= 0 && lx + 8 <= p.Q;
            r[1][0] = io_ld4<IO>(v0 ? lp : Lg); r[1][1] = io_ld4<IO>(v1 ? lp + 4 * ES : Lg);
            m |= (v0 ? 4u : 0u) | (v1 ? 8u : 0u);
        }
        return m;
    };
    // prefetch of chunk ch: its S pixels and the ONE L row the ring does not hold yet (pp + 1)
    auto fetch = [&](int n, int qb, int pp) {
        const int q0 = qb * 32;
        vmask = 0;
        const bool s_in = s_ch_ok && q0 + 8 * s_grp + 8 <= p.Q;         // (rows of 16 pixels: the second half of the chunk does not exist -- conv_igemm.hip, wgrad_wide16)
        const char* sp = s_in ? Sg + ((int64_t)n * p.SC * PQ + (int64_t)s_a * PQ + pp * p.Q + q0 + 8 * s_grp) * ES : Sg;
        sreg[0] = io_ld4<IO>(sp); sreg[1] = io_ld4<IO>(sp + 4 * ES);
        vmask |= s_in ? 3u : 0u;
        vmask |= fetch_row(n, q0, pp + 1, lreg) << 2;
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    // split 8 floats (two float4 halves, each possibly invalid -> 0) into three packed bf16x8 pieces and store them
    auto split_store = [&](float4 h0, float4 h1, bool ok0, bool ok1, __bf16* dst, int piece_stride, float scale) {
        const float vals[8] = {ok0 ? h0.x : 0.f, ok0 ? h0.y : 0.f, ok0 ? h0.z : 0.f, ok0 ? h0.w : 0.f,
                               ok1 ? h1.x : 0.f, ok1 ? h1.y : 0.f, ok1 ? h1.z : 0.f, ok1 ? h1.w : 0.f};
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (HX) {
                f16_split2_direct(vals[2 * j] * scale, vals[2 * j + 1] * scale, q1[j], q2[j]);
                continue;
            }
            f32x2 v = {vals[2 * j], vals[2 * j + 1]};
            uint32_t w = io_pack2<IO>(vals[2 * j], vals[2 * j + 1]);
            q1[j] = w;
            if constexpr (NP >= 2) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                q2[j] = w;
            }
            if constexpr (NP >= 3) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            }
        }
        *(uint4*)(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        if constexpr (NPW >= 2) *(uint4*)(dst + piece_stride) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if constexpr (NPW >= 3) *(uint4*)(dst + 2 * piece_stride) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };
    // one L row from registers into ring slot `slot`
    auto stash_row = [&](const float4 (&r)[2][2], unsigned m, int slot) {
        split_store(r[0][0], r[0][1], m & 1u, m & 2u, Ls + (l_b0 * 3 + slot) * LP + 8 * l_g0, L_PIECE, l_scale);
        if (tid < 64) split_store(r[1][0], r[1][1], m & 4u, m & 8u, Ls + (l_b1 * 3 + slot) * LP + 8 * l_g1, L_PIECE, l_scale);
    };
    auto slot_of = [](int y) { return (y + 3) % 3; };       // y >= -1

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);
    // position of the current chunk and of the next one: decoded ONCE (two divisions by run-time values), then advanced -- the chunks walk down a
    // column block, then the column blocks of an image, then the images (a decode per chunk and per prefetch was ~200 scalar instructions per
    // chunk in front of this wave's matrix instructions)
    int n_c = 0, qb_c = 0, pp_c = 0, n_n = 0, qb_n = 0, pp_n = 0;
    auto advance = [&](int n, int qb, int pp, int& n2, int& qb2, int& pp2) {
        pp2 = pp + 1; qb2 = qb; n2 = n;
        if (pp2 == p.P) { pp2 = 0; qb2 = qb + 1; if (qb2 == p.qblocks) { qb2 = 0; n2 = n + 1; } }
    };
    if (c_begin < c_end) { decode(c_begin, n_c, qb_c, pp_c); fetch(n_c, qb_c, pp_c); advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n); }
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        if (pp_c == 0 || ch == c_begin) { // a column block or this K slice begins: rows pp-1 and pp are not in the ring yet
            float4 t[2][2];
            unsigned m = fetch_row(n_c, qb_c * 32, pp_c - 1, t);
            stash_row(t, m, slot_of(pp_c - 1));
            m = fetch_row(n_c, qb_c * 32, pp_c, t);
            stash_row(t, m, slot_of(pp_c));
        }
        split_store(sreg[0], sreg[1], vmask & 1u, vmask & 2u, Ss + s_a * SP + 8 * s_grp, S_PIECE, s_scale);
        stash_row(lreg, vmask >> 2, slot_of(pp_c + 1));
        __syncthreads();
        if (ch + 1 < c_end) fetch(n_n, qb_n, pp_n);
        const int slot0 = slot_of(pp_c - 1);          // ring slot of halo row 0; rows 1, 2 follow cyclically
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 af[3];
#pragma unroll
            for (int pc = 0; pc < NPW; pc++) af[pc] = *(const bf16x8*)&Ss[pc * S_PIECE + (wa * 32 + jl) * SP + 16 * s + 8 * hl];
#pragma unroll
            for (int pb = NPW - 1; pb >= 0; pb--) {  // B pieces from the smallest to the largest
#pragma unroll
                for (int row = 0; row < 3; row++) {
                    const int slot = slot0 + row >= 3 ? slot0 + row - 3 : slot0 + row;
                    const __bf16* lb = &Ls[pb * L_PIECE + ((wb * 32 + jl) * 3 + slot) * LP + 16 * s + 8 * hl];
                    // Only dwords 1..6 are used and the compiler narrows the two reads to ds_read2_b64 + ds_read2_b32, whose
                    // 32-bank rule makes the 60-dword lane stride conflict 2-way (SQ_LDS_BANK_CONFLICT = half of this kernel's
                    // LDS cycles).  Forcing whole ds_read_b128 (PASTA_KEEP_WHOLE) removes the conflicts; measured twice: round 1
                    // (250 of 256 registers) it spilled and ran 12 % slower, round 2 (with the row ring's smaller staging state it
                    // fits: 252 registers, no scratch) it runs 1 - 3 % slower than the narrowed reads
                    // (profiles/r2_wgrad_ring.txt) -- the conflicts are not what limits this kernel.
                    // (whole ds_read_b128 under the three-product arithmetic, where the conflicts could weigh more: measured, no change --
                    // gpurun_out/r3_ab_wgrad_whole.txt)
                    const uint4 b0 = *(const uint4*)lb, b1 = *(const uint4*)(lb + 8);
                    const uint32_t d[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    uint4 w[3];
                    w[0] = make_uint4(__builtin_amdgcn_alignbit(d[2], d[1], 16), __builtin_amdgcn_alignbit(d[3], d[2], 16),
                                      __builtin_amdgcn_alignbit(d[4], d[3], 16), __builtin_amdgcn_alignbit(d[5], d[4], 16));   // halo col 3
                    w[1] = make_uint4(d[2], d[3], d[4], d[5]);                                                                 // halo col 4
                    w[2] = make_uint4(__builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16),
                                      __builtin_amdgcn_alignbit(d[5], d[4], 16), __builtin_amdgcn_alignbit(d[6], d[5], 16));   // halo col 5
#pragma unroll
                    for (int ts = 0; ts < 3; ts++) {
                        const bf16x8 bw = __builtin_bit_cast(bf16x8, w[ts]);
                        const int tap = row * 3 + ts;
                        // the products of the arithmetic (split-bf16: pa + pb < NP), smallest A piece first
#pragma unroll
                        for (int pa = NPW - 1; pa >= 0; pa--)
                            if (mmw_on<NP>(pa, pb)) acc[tap] = mfma16<IO, NP>(af[pa], bw, acc[tap]);
                    }
                }
            }
        }
        n_c = n_n; qb_c = qb_n; pp_c = pp_n;
        advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n);
    }

    // partial slab: [ksplit][G][9][Ag_pad][Bg_pad], b contiguous (same layout as conv_wgrad_kernel)
    const int Ag_pad = p.a_tiles * 64, Bg_pad = p.b_tiles * 64;
    float* out = p.slab + ((int64_t)ks * p.G + g) * 9 * Ag_pad * Bg_pad;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* ot = out + (int64_t)t * Ag_pad * Bg_pad;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = b_blk + wb * 32 + jl;
            ot[(int64_t)a * Bg_pad + b] = HX ? acc[t][r] * out_scale : acc[t][r];
        }
    }
}

//------------------------------------------------------------------------------------
// Stride-2 sibling of conv_wgrad3x3_bf16x6_kernel (3x3, stride 2, pad 0 or 1 on both axes; conv2d and, with the
// operand roles swapped, conv_transpose2d): dW[tap r,s][a][b] = sum_pix S[a][p][q] * L[b][2p + r - pad][2q + s - pad].
// K chunk = 16 consecutive S pixels of one row = one K step.  The L halo is three rows of 40 columns starting at
// column 2*q0 - 4, split once and stored as [piece][b][3 rows][40] like the stride-1 kernel; the operand of tap column
// s is every second halo element from 4 - pad + s on, gathered from three aligned 16-byte LDS reads with v_perm_b32
// (the bf16 pairs of a dword are halo columns 2i, 2i + 1: a window of even or of odd columns is the low or the high
// halves of eight consecutive dwords).  L rows are not 16-byte aligned in general (257-pixel planes), so the halo is
// fetched with dword loads.
template <int PW, int NP, int IO = IO_F32>
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3s2_bf16x6_kernel(WgradParams p) {
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: one product");
    constexpr bool HX = Arith<NP>::f16x3;           // PASTA_MATH_F16X3: both operands as (h, l), three products (conv_common.h)
    constexpr int NPW = Arith<NP>::npw;             // pieces per operand in LDS
    constexpr int ES = io_size<IO>::value;
    constexpr int SP = 16, LP = 40;                 // row pitches in bf16 elements
    constexpr int S_PIECE = 64 * SP;
    constexpr int L_PIECE = 64 * 3 * LP;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    __bf16* Ss = smem16;                            // [3][64][SP]
    __bf16* Ls = smem16 + NPW * S_PIECE;            // [NPW][64][3][LP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;
    float s_scale = 1.f, l_scale = 1.f, out_scale = 1.f;       // PASTA_MATH_F16X3: operand scales from the tensors' partial maxima
    if constexpr (HX) {
        float is_, il_;
        scale_from_amax(amax_of_parts(p.s_amax), s_scale, is_);
        scale_from_amax(amax_of_parts(p.l_amax), l_scale, il_);
        out_scale = is_ * il_;
    }

    int ks, bt, at, g;
    wgrad_decode(p, ks, bt, at, g);
    const int a_blk = at * 64, b_blk = bt * 64;
    const int PQ = p.P * p.Q, LHW = p.LH * p.LW;
    const char* const Sg = (const char*)p.S + ((int64_t)g * p.Ag + a_blk) * PQ * ES;
    const char* const Lg = (const char*)p.L + ((int64_t)g * p.Bg + b_blk) * LHW * ES;

    // Chunk order as in the stride-1 kernel: down a 16-pixel column block of one image.  The halo of chunk pp is L rows
    // 2pp - pad + {0, 1, 2}; its last row is the first row of chunk pp + 1, so with the rows in a ring in LDS (row y in slot
    // y mod 3) a chunk fetches, splits and stores TWO new rows instead of three; the first row is primed where a column block
    // or a K slice begins.
    // staging roles (fixed): S unit = (channel a, group of 8 pixels), threads 0..127; L row units = (channel b, group of 8
    // columns), 320 per row: unit v = tid + 256 j < 640 belongs to new row 1 + v / 320
    const int s_a = tid >> 1, s_grp = tid & 1;
    const bool s_on = tid < 128 && a_blk + s_a < p.Ag;
    int l_b[3], l_grp[3], l_k[3];
    bool l_ch_ok[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int v = tid + 256 * j;
        l_k[j] = v >= 320 ? 2 : 1;
        const int u = v - (l_k[j] - 1) * 320;
        l_b[j] = u / 5; l_grp[j] = u - l_b[j] * 5;
        l_ch_ok[j] = v < 640 && b_blk + l_b[j] < p.Bg;
    }
    // the priming row (k = 0): unit tid, and unit 256 + tid for the first 64 threads
    const int p_b0 = tid / 5, p_g0 = tid - p_b0 * 5, p_b1 = (256 + tid) / 5, p_g1 = (256 + tid) - p_b1 * 5;

    float4 sreg[2];
    float lreg[3][8];
    unsigned lmask = 0;                              // validity bit of each of the 24 halo elements held in registers
    bool s_ok = false;
    auto decode = [&](int ch, int& n, int& qb, int& pp) {
        const int per_img = p.P * p.qblocks;
        n = ch / per_img;
        const int rem = ch - n * per_img;
        qb = rem / p.P; pp = rem - qb * p.P;
    };
    // eight consecutive elements of L row ly of channel b (sample n) from column lx on; returns their validity bits
    auto fetch_unit = [&](int n, int ly, int lx, int b, bool ch_ok, float (&r)[8]) -> unsigned {
        const bool rok = ch_ok && (unsigned)ly < (unsigned)p.LH;
        const char* lp = Lg + ((int64_t)n * p.LC * LHW + (int64_t)b * LHW + ly * p.LW + lx) * ES;
        int first = lx < 0 ? -lx : 0, last = p.LW - lx < 8 ? p.LW - lx : 8;
        if (!rok || last < 0) last = 0;
        if (first > last) first = last;
        const unsigned m = ((1u << last) - 1u) & ~((1u << first) - 1u);
        if (m == 0xffu) {
#pragma unroll
            for (int e = 0; e < 8; e++) r[e] = io_ld1<IO>(lp + e * ES);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if ((m >> e) & 1u) r[e] = io_ld1<IO>(lp + e * ES);
        }
        // (round 5, measured and dropped: every element fetched unconditionally from a selected address, as the stride-1 kernel now does with its
        // four-packs -- two address selects per ELEMENT here: 123 -> 98 TFLOP/s on 64 -> 128 at 257 x 257, profiles/r5_ab_wgrad_branchfree.txt)
        return m;
    };
    auto fetch = [&](int n, int qb, int pp) {
        const int q0 = qb * 16;
        s_ok = s_on;
        if (tid < 128) {                              // (wave-uniform: the S units belong to the first two waves)
            const char* sp = s_on ? Sg + ((int64_t)n * p.SC * PQ + (int64_t)s_a * PQ + pp * p.Q + q0 + 8 * s_grp) * ES : Sg;
            sreg[0] = io_ld4<IO>(sp); sreg[1] = io_ld4<IO>(sp + 4 * ES);
        }
        lmask = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            lmask |= fetch_unit(n, 2 * pp + l_k[j] - p.pad_h, 2 * q0 - 4 + 8 * l_grp[j], l_b[j], l_ch_ok[j], lreg[j]) << (8 * j);
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_store = [&](const float* vals, __bf16* dst, int piece_stride, float scale) {
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (HX) {
                f16_split2_direct(vals[2 * j] * scale, vals[2 * j + 1] * scale, q1[j], q2[j]);
                continue;
            }
            f32x2 v = {vals[2 * j], vals[2 * j + 1]};
            uint32_t w = io_pack2<IO>(vals[2 * j], vals[2 * j + 1]);
            q1[j] = w;
            if constexpr (NP >= 2) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                q2[j] = w;
            }
            if constexpr (NP >= 3) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            }
        }
        *(uint4*)(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        if constexpr (NPW >= 2) *(uint4*)(dst + piece_stride) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if constexpr (NPW >= 3) *(uint4*)(dst + 2 * piece_stride) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };
    auto stash = [&]() {
        if (tid < 128) {
            const float sv[8] = {s_ok ? sreg[0].x : 0.f, s_ok ? sreg[0].y : 0.f, s_ok ? sreg[0].z : 0.f, s_ok ? sreg[0].w : 0.f,
                                 s_ok ? sreg[1].x : 0.f, s_ok ? sreg[1].y : 0.f, s_ok ? sreg[1].z : 0.f, s_ok ? sreg[1].w : 0.f};
            split_store(sv, Ss + s_a * SP + 8 * s_grp, S_PIECE, s_scale);
        }
    };
    auto slot_of = [](int y) { return (y + 3) % 3; };       // y >= -1
    // one staged L unit (eight elements, validity bits m) into ring slot `slot`
    auto stash_unit = [&](const float (&r)[8], unsigned m, int b, int grp, int slot) {
        float lv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) lv[e] = ((m >> e) & 1u) ? r[e] : 0.f;
        split_store(lv, Ls + (b * 3 + slot) * LP + 8 * grp, L_PIECE, l_scale);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);
    // position of the current chunk and of the next one: decoded ONCE (two divisions by run-time values), then advanced -- the chunks walk down a
    // column block, then the column blocks of an image, then the images (a decode per chunk and per prefetch was ~200 scalar instructions per
    // chunk in front of this wave's matrix instructions)
    int n_c = 0, qb_c = 0, pp_c = 0, n_n = 0, qb_n = 0, pp_n = 0;
    auto advance = [&](int n, int qb, int pp, int& n2, int& qb2, int& pp2) {
        pp2 = pp + 1; qb2 = qb; n2 = n;
        if (pp2 == p.P) { pp2 = 0; qb2 = qb + 1; if (qb2 == p.qblocks) { qb2 = 0; n2 = n + 1; } }
    };
    if (c_begin < c_end) { decode(c_begin, n_c, qb_c, pp_c); fetch(n_c, qb_c, pp_c); advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n); }
    for (int ch = c_begin; ch < c_end; ch++) {
        const int y0 = 2 * pp_c - p.pad_h;
        __syncthreads();                  // the previous chunk's fragment reads are done
        if (pp_c == 0 || ch == c_begin) { // a column block or this K slice begins: the first halo row is not in the ring yet
            float t[8];
            unsigned m = fetch_unit(n_c, y0, 2 * qb_c * 16 - 4 + 8 * p_g0, p_b0, b_blk + p_b0 < p.Bg, t);
            stash_unit(t, m, p_b0, p_g0, slot_of(y0));
            if (tid < 64) {
                m = fetch_unit(n_c, y0, 2 * qb_c * 16 - 4 + 8 * p_g1, p_b1, b_blk + p_b1 < p.Bg, t);
                stash_unit(t, m, p_b1, p_g1, slot_of(y0));
            }
        }
        stash();
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (tid + 256 * j < 640) stash_unit(lreg[j], (lmask >> (8 * j)) & 0xffu, l_b[j], l_grp[j], slot_of(y0 + l_k[j]));
        __syncthreads();
        if (ch + 1 < c_end) fetch(n_n, qb_n, pp_n);
        const int slot0 = slot_of(y0);
        bf16x8 af[3];
#pragma unroll
        for (int pc = 0; pc < NPW; pc++) af[pc] = *(const bf16x8*)&Ss[pc * S_PIECE + (wa * 32 + jl) * SP + 8 * hl];
#pragma unroll
        for (int pb = NPW - 1; pb >= 0; pb--) {      // B pieces from the smallest to the largest
#pragma unroll
            for (int row = 0; row < 3; row++) {
                const int slot = slot0 + row >= 3 ? slot0 + row - 3 : slot0 + row;
                const __bf16* lb = &Ls[pb * L_PIECE + ((wb * 32 + jl) * 3 + slot) * LP + 16 * hl];
                uint4 b0 = *(const uint4*)lb, b1 = *(const uint4*)(lb + 8), b2 = *(const uint4*)(lb + 16);
                PASTA_KEEP_WHOLE(b0); PASTA_KEEP_WHOLE(b1); PASTA_KEEP_WHOLE(b2);      // whole ds_read_b128 (conflict-free) instead of narrowed read2 pairs: +2 % here
                const uint32_t d[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                for (int ts = 0; ts < 3; ts++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int c0 = 4 - PW + ts, d0 = c0 >> 1;
                    const uint32_t sel = (c0 & 1) ? 0x07060302u : 0x05040100u;
                    const uint4 w = make_uint4(__builtin_amdgcn_perm(d[d0 + 1], d[d0], sel), __builtin_amdgcn_perm(d[d0 + 3], d[d0 + 2], sel),
                                               __builtin_amdgcn_perm(d[d0 + 5], d[d0 + 4], sel), __builtin_amdgcn_perm(d[d0 + 7], d[d0 + 6], sel));
                    const bf16x8 bw = __builtin_bit_cast(bf16x8, w);
                    const int tap = row * 3 + ts;
#pragma unroll
                    for (int pa = NPW - 1; pa >= 0; pa--)
                        if (mmw_on<NP>(pa, pb)) acc[tap] = mfma16<IO, NP>(af[pa], bw, acc[tap]);
                }
            }
        }
        n_c = n_n; qb_c = qb_n; pp_c = pp_n;
        advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n);
    }

    const int Ag_pad = p.a_tiles * 64, Bg_pad = p.b_tiles * 64;
    float* out = p.slab + ((int64_t)ks * p.G + g) * 9 * Ag_pad * Bg_pad;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* ot = out + (int64_t)t * Ag_pad * Bg_pad;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = b_blk + wb * 32 + jl;
            ot[(int64_t)a * Bg_pad + b] = HX ? acc[t][r] * out_scale : acc[t][r];
        }
    }
}

//------------------------------------------------------------------------------------
// Round 5: the stride-2 weight gradient with L given as the producer-written operand (PASTA_LAYOUT_PIECES16, pieces.hip: [N][LC / 8][LH][2][LW] units of
// 16 bytes, fp16 h[8] / l'[8] of v S) -- the blurred tensor in front of every stride-2 convolution of the discriminator and the encoders
// (conv2d_resample.py:119-122), pad 0, PASTA_MATH_F16X3.  conv_wgrad3x3s2_bf16x6_kernel spends 41 % of its wave cycles issuing instructions
// (profiles/r4_pmc_summary.txt): 24 dword loads and three splits per thread and chunk for the halo, and in the product loop four v_perm per
// operand to gather every second halo column.  Here
//   * the halo rows are COPIED: a staging unit is (pixel, channel octet) = two sixteen-byte loads (h, l') and two sixteen-byte LDS stores into a
//     [piece][ring slot][column][64 channels] image -- the layout of the tensor itself, channels contiguous;
//   * the operand of tap (r, s) -- K = 16 S pixels x 32 channels b, K-contiguous per lane -- is gathered by the transposed LDS read of gfx950,
//     ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows (K: halo columns 2 k + s, i.e. every second column: the stride is just the
//     rows' addresses) x 16 columns (channels) arrives column-major.  The column pitch is 160 bytes (64 channels + 32 bytes of padding), so the
//     four rows of a block -- two columns = 320 bytes = 80 dwords apart -- fall on the four quarters of the 64 banks: conflict-free;
//   * l' is the pre-scaled low piece (2^11 r); the weight gradient's products (S h)(L h) + (S l)(L h) + (S h)(L l) become
//     (S h)(L h) + (S l)(L h) + (S h'')(L l') with h'' = 2^-11 h formed when S is split (exact down to 2^-3 in the operand's units, i.e. for
//     elements within 2^-17 of their tensor's largest: the range the arithmetic states for weight-gradient operands, conv_common.h).
// S (dy, fp32 NCHW) is staged and split as in the kernel above.  Same chunk order, ring of three halo rows, slab layout and reduction.
// LDS: S [3 pieces][64][16] + L [2][3][34][80] fp16 = 38.8 KB.
template <int PW>        // the convolution's pad: 0 (the blur in front of it absorbs the padding); a template so that only the unit that launches it holds the kernel
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3s2_pieces_kernel(WgradParams p) {
    static_assert(PW == 0, "the producer-written operand feeds pad-0 convolutions");
    constexpr int SP = 16, S_PIECE = 64 * SP;
    constexpr int LCOLS = 34, LPX = 80, L_ROW = LCOLS * LPX, L_PIECE = 3 * L_ROW;      // halo columns 2 q0 .. 2 q0 + 33 (taps reach 2 q0 + 32)
    __shared__ __attribute__((aligned(16))) _Float16 Ss[3 * S_PIECE];
    __shared__ __attribute__((aligned(16))) _Float16 Ls[2 * L_PIECE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;
    float s_scale, l_scale, is_, il_;
    scale_from_amax(amax_of_parts(p.s_amax), s_scale, is_);
    scale_from_amax(amax_of_parts(p.l_amax), l_scale, il_);           // the producer's bound row: the S the pieces were written with
    const float out_scale = is_ * il_;

    int ks, bt, at, g;
    wgrad_decode(p, ks, bt, at, g);                  // g = 0: one group
    const int a_blk = at * 64, b_blk = bt * 64;
    const int PQ = p.P * p.Q;
    const float* const Sg = p.S + (int64_t)a_blk * PQ;
    const int LC8 = p.LC >> 3;
    const char* const Lg = (const char*)p.L + (int64_t)(b_blk >> 3) * p.LH * p.LW * 32;

    auto decode = [&](int ch, int& n, int& qb, int& pp) {
        const int per_img = p.P * p.qblocks;
        n = ch / per_img;
        const int rem = ch - n * per_img;
        qb = rem / p.P; pp = rem - qb * p.P;
    };
    // staging roles: S unit = (channel a, eight pixels), threads 0..127; L units = (column, octet) of one halo row: 34 x 8 = 272 per row,
    // unit v = tid + 256 j < 544 belongs to new row 1 + v / 272 (the ring holds row 0); octet fastest: eight lanes fill one column's 128 bytes
    const int s_a = tid >> 1, s_grp = tid & 1;
    const bool s_on = tid < 128 && a_blk + s_a < p.Ag;
    int l_k[3], l_col[3], l_oct[3];
    bool l_on[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int v = tid + 256 * j;
        l_k[j] = v >= 272 ? 2 : 1;
        const int u = v - (l_k[j] - 1) * 272;
        l_col[j] = u >> 3; l_oct[j] = u & 7;
        l_on[j] = v < 544 && b_blk + 8 * l_oct[j] < p.Bg;
    }
    float4 sreg[2];
    u32x4 lreg[3][2];
    bool lok[3];
    // one unit of halo row ly: column lx, octet oct -> (h, l'); false = outside the plane (a zero unit)
    auto fetch_unit = [&](int n, int ly, int lx, int oct, bool on, u32x4 (&r)[2]) -> bool {
        const bool ok = on && (unsigned)ly < (unsigned)p.LH && (unsigned)lx < (unsigned)p.LW;
        const u32x4* src = (const u32x4*)(Lg + ((int64_t)n * LC8 + oct) * p.LH * p.LW * 32) + (ok ? 2 * ly * p.LW + lx : 0);      // [..][LH][2 pieces][LW] units
        if (!on) src = (const u32x4*)Lg;             // an octet beyond the tensor: any valid address
        r[0] = src[0]; r[1] = src[p.LW];
        return ok;
    };
    auto stash_unit = [&](const u32x4 (&r)[2], bool ok, int slot, int col, int oct) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        _Float16* const d = Ls + slot * L_ROW + col * LPX + oct * 8;
        *(u32x4*)d = ok ? r[0] : z;
        *(u32x4*)(d + L_PIECE) = ok ? r[1] : z;
    };
    auto fetch = [&](int n, int qb, int pp) {
        const int q0 = qb * 16;
        if (tid < 128) {                              // (wave-uniform; a channel beyond the tensor re-reads its first element and is zeroed at the split)
            const float* sp = s_on ? Sg + (int64_t)n * p.SC * PQ + (int64_t)s_a * PQ + pp * p.Q + q0 + 8 * s_grp : Sg;
            sreg[0] = *(const float4*)sp; sreg[1] = *(const float4*)(sp + 4);
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
            lok[j] = fetch_unit(n, 2 * pp + l_k[j], 2 * q0 + l_col[j], l_oct[j], l_on[j], lreg[j]);
    };
    auto slot_of = [](int y) { return y % 3; };          // y >= 0 (pad 0)
    typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
    auto stash_s = [&]() {
        if (tid >= 128) return;
        const float sv[8] = {s_on ? sreg[0].x : 0.f, s_on ? sreg[0].y : 0.f, s_on ? sreg[0].z : 0.f, s_on ? sreg[0].w : 0.f,
                             s_on ? sreg[1].x : 0.f, s_on ? sreg[1].y : 0.f, s_on ? sreg[1].z : 0.f, s_on ? sreg[1].w : 0.f};
        uint32_t qh[4], ql[4], qs[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f16_split2_direct(sv[2 * j] * s_scale, sv[2 * j + 1] * s_scale, qh[j], ql[j]);
            const f16x2v h2 = __builtin_bit_cast(f16x2v, qh[j]);
            const f16x2v k2 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};       // 2^-11: an exponent shift
            qs[j] = __builtin_bit_cast(uint32_t, h2 * k2);
        }
        _Float16* const d = Ss + s_a * SP + 8 * s_grp;
        *(uint4*)d = make_uint4(qh[0], qh[1], qh[2], qh[3]);
        *(uint4*)(d + S_PIECE) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
        *(uint4*)(d + 2 * S_PIECE) = make_uint4(qs[0], qs[1], qs[2], qs[3]);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // transposed reads: lane 4 q + c of a 16-lane group supplies the address of block row q (K), columns 4 c .. 4 c + 3 (channels); the group's
    // lane i receives column i, row q in element q.  Group (lane >> 4): channels 16 (grp & 1) .. + 15 of this wave's 32, K block 8 (grp >> 1) (= 8 hl)
    const int tq = (lane >> 2) & 3, tc = lane & 3, tgrp = lane >> 4;
    const _Float16* const lbase = Ls + (2 * (8 * (tgrp >> 1) + tq)) * LPX + wb * 32 + 16 * (tgrp & 1) + 4 * tc;
    typedef __fp16 v4h __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4h* lds_v4h;

    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);
    // position of the current chunk and of the next one: decoded ONCE (two divisions by run-time values), then advanced -- the chunks walk down a
    // column block, then the column blocks of an image, then the images (a decode per chunk and per prefetch was ~200 scalar instructions per
    // chunk in front of this wave's matrix instructions)
    int n_c = 0, qb_c = 0, pp_c = 0, n_n = 0, qb_n = 0, pp_n = 0;
    auto advance = [&](int n, int qb, int pp, int& n2, int& qb2, int& pp2) {
        pp2 = pp + 1; qb2 = qb; n2 = n;
        if (pp2 == p.P) { pp2 = 0; qb2 = qb + 1; if (qb2 == p.qblocks) { qb2 = 0; n2 = n + 1; } }
    };
    if (c_begin < c_end) { decode(c_begin, n_c, qb_c, pp_c); fetch(n_c, qb_c, pp_c); advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n); }
    for (int ch = c_begin; ch < c_end; ch++) {
        const int y0 = 2 * pp_c;
        __syncthreads();                  // the previous chunk's fragment reads are done
        if (pp_c == 0 || ch == c_begin) { // a column block or this K slice begins: the first halo row is not in the ring yet (272 units: tid, and 256 + tid for 16 threads)
            u32x4 t[2];
            bool ok = fetch_unit(n_c, y0, 2 * qb_c * 16 + (tid >> 3), tid & 7, b_blk + 8 * (tid & 7) < p.Bg, t);
            stash_unit(t, ok, slot_of(y0), tid >> 3, tid & 7);
            if (tid < 16) {
                ok = fetch_unit(n_c, y0, 2 * qb_c * 16 + 32 + (tid >> 3), tid & 7, b_blk + 8 * (tid & 7) < p.Bg, t);
                stash_unit(t, ok, slot_of(y0), 32 + (tid >> 3), tid & 7);
            }
        }
        stash_s();
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (tid + 256 * j < 544) stash_unit(lreg[j], lok[j], slot_of(y0 + l_k[j]), l_col[j], l_oct[j]);
        __syncthreads();
        if (ch + 1 < c_end) fetch(n_n, qb_n, pp_n);
        const int slot0 = slot_of(y0);
        bf16x8 af[3];
#pragma unroll
        for (int pc = 0; pc < 3; pc++) af[pc] = *(const bf16x8*)&Ss[pc * S_PIECE + (wa * 32 + jl) * SP + 8 * hl];
#pragma unroll
        for (int row = 0; row < 3; row++) {
            const int slot = slot0 + row >= 3 ? slot0 + row - 3 : slot0 + row;
            const _Float16* const lrow = lbase + slot * L_ROW;
#pragma unroll
            for (int ts = 0; ts < 3; ts++) {
                // K rows 8 hl + q and 8 hl + 4 + q of tap column ts: halo columns 2 k + ts
                const v4h h0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h)(lrow + ts * LPX));
                const v4h h1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h)(lrow + (8 + ts) * LPX));
                const v4h l0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h)(lrow + L_PIECE + ts * LPX));
                const v4h l1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h)(lrow + L_PIECE + (8 + ts) * LPX));
                typedef __fp16 v8h __attribute__((ext_vector_type(8)));
                const bf16x8 bh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                const int tap = row * 3 + ts;
                acc[tap] = mfma16<IO_F32, NP_F16X3>(af[2], bl, acc[tap]);       // (S h'')(L l'): the smallest term first
                acc[tap] = mfma16<IO_F32, NP_F16X3>(af[1], bh, acc[tap]);       // (S l)(L h)
                acc[tap] = mfma16<IO_F32, NP_F16X3>(af[0], bh, acc[tap]);       // (S h)(L h)
            }
        }
        n_c = n_n; qb_c = qb_n; pp_c = pp_n;
        advance(n_c, qb_c, pp_c, n_n, qb_n, pp_n);
    }

    const int Ag_pad = p.a_tiles * 64, Bg_pad = p.b_tiles * 64;
    float* out = p.slab + (int64_t)ks * 9 * Ag_pad * Bg_pad;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* ot = out + (int64_t)t * Ag_pad * Bg_pad;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int a = a_blk + wa * 32 + acc_row(r, lane), b = b_blk + wb * 32 + jl;
            ot[(int64_t)a * Bg_pad + b] = acc[t][r] * out_scale;
        }
    }
}

//------------------------------------------------------------------------------------
// 1x1, stride 1, no padding (the residual skips, ToRGB-side merges and the discriminator's fromrgb/skip convolutions):
// a plain GEMM dW[a][b] = sum_pix S[a][pix] * L[b][pix] over K = all pixels, both operands pixel-contiguous.  Same
// split-bf16 arithmetic; workgroup tile (64 WA) x (64 WB) channels, every wave WA x WB tiles of 32 x 32; K chunk = 32
// consecutive pixels of one image (P*Q % 32 == 0).  Both operands are split and stored as [piece][channel][32 px]
// (row pitch 40 bf16).  The shape is bandwidth-bound: 2 x 64 x (WA + WB) x 32 floats per 32 x (64 WA)(64 WB) MACs.
template <int WA, int WB, int NP, int IO = IO_F32>
__global__ __launch_bounds__(256, 2) void conv_wgrad1x1_bf16x6_kernel(WgradParams p) {
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: one product");
    constexpr bool HX = Arith<NP>::f16x3;           // PASTA_MATH_F16X3: both operands as (h, l), three products (conv_common.h)
    constexpr int NPW = Arith<NP>::npw;             // pieces per operand in LDS
    constexpr int ES = io_size<IO>::value;
    constexpr int SP = 40;
    constexpr int TA = 64 * WA, TB = 64 * WB;
    constexpr int A_PIECE = TA * SP, B_PIECE = TB * SP;
    constexpr int UA = TA * 4 / 256, UB = TB * 4 / 256;             // (channel, 8-pixel group) units per thread
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    __bf16* Ss = smem16;                            // [3][TA][SP]
    __bf16* Ls = smem16 + NPW * A_PIECE;            // [NPW][TB][SP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int hl = lane >> 5, jl = lane & 31;
    float s_scale = 1.f, l_scale = 1.f, out_scale = 1.f;       // PASTA_MATH_F16X3: operand scales from the tensors' partial maxima
    if constexpr (HX) {
        float is_, il_;
        scale_from_amax(amax_of_parts(p.s_amax), s_scale, is_);
        scale_from_amax(amax_of_parts(p.l_amax), l_scale, il_);
        out_scale = is_ * il_;
    }

    int ks, bt, at, g;
    wgrad_decode(p, ks, bt, at, g);
    const int a_blk = at * TA, b_blk = bt * TB;
    const int PQ = p.P * p.Q;
    const char* const Sg = (const char*)p.S + ((int64_t)g * p.Ag + a_blk) * PQ * ES;
    const char* const Lg = (const char*)p.L + ((int64_t)g * p.Bg + b_blk) * PQ * ES;
    const int chunks_per_image = PQ / 32;

    float4 sreg[UA][2], lreg[UB][2];
    auto fetch = [&](int ch) {
        const int n = ch / chunks_per_image, px0 = (ch - n * chunks_per_image) * 32;
#pragma unroll
        for (int j = 0; j < UA; j++) {
            const int u = tid + 256 * j, c = u >> 2, grp = u & 3;
            if (a_blk + c < p.Ag) {
                const char* sp = Sg + ((int64_t)n * p.SC * PQ + (int64_t)c * PQ + px0 + 8 * grp) * ES;
                sreg[j][0] = io_ld4<IO>(sp); sreg[j][1] = io_ld4<IO>(sp + 4 * ES);
            }
        }
#pragma unroll
        for (int j = 0; j < UB; j++) {
            const int u = tid + 256 * j, c = u >> 2, grp = u & 3;
            if (b_blk + c < p.Bg) {
                const char* lp = Lg + ((int64_t)n * p.LC * PQ + (int64_t)c * PQ + px0 + 8 * grp) * ES;
                lreg[j][0] = io_ld4<IO>(lp); lreg[j][1] = io_ld4<IO>(lp + 4 * ES);
            }
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto split_store = [&](float4 h0, float4 h1, bool ok, __bf16* dst, int piece_stride, float scale) {
        const float vals[8] = {ok ? h0.x : 0.f, ok ? h0.y : 0.f, ok ? h0.z : 0.f, ok ? h0.w : 0.f,
                               ok ? h1.x : 0.f, ok ? h1.y : 0.f, ok ? h1.z : 0.f, ok ? h1.w : 0.f};
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (HX) {
                f16_split2_direct(vals[2 * j] * scale, vals[2 * j + 1] * scale, q1[j], q2[j]);
                continue;
            }
            f32x2 v = {vals[2 * j], vals[2 * j + 1]};
            uint32_t w = io_pack2<IO>(vals[2 * j], vals[2 * j + 1]);
            q1[j] = w;
            if constexpr (NP >= 2) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                q2[j] = w;
            }
            if constexpr (NP >= 3) {
                v[0] -= __builtin_bit_cast(float, w << 16);
                v[1] -= __builtin_bit_cast(float, w & 0xffff0000u);
                q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            }
        }
        *(uint4*)(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        if constexpr (NPW >= 2) *(uint4*)(dst + piece_stride) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if constexpr (NPW >= 3) *(uint4*)(dst + 2 * piece_stride) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < UA; j++) {
            const int u = tid + 256 * j, c = u >> 2, grp = u & 3;
            split_store(sreg[j][0], sreg[j][1], a_blk + c < p.Ag, Ss + c * SP + 8 * grp, A_PIECE, s_scale);
        }
#pragma unroll
        for (int j = 0; j < UB; j++) {
            const int u = tid + 256 * j, c = u >> 2, grp = u & 3;
            split_store(lreg[j][0], lreg[j][1], b_blk + c < p.Bg, Ls + c * SP + 8 * grp, B_PIECE, l_scale);
        }
    };

    f32x16 acc[WA][WB];
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < WB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int c_begin = (int)(((unsigned)p.chunks_total * (unsigned)ks) / (unsigned)p.ksplit);
    const int c_end = (int)(((unsigned)p.chunks_total * (unsigned)(ks + 1)) / (unsigned)p.ksplit);
    if (c_begin < c_end) fetch(c_begin);
    for (int ch = c_begin; ch < c_end; ch++) {
        __syncthreads();                  // the previous chunk's fragment reads are done
        stash();
        __syncthreads();
        if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 af[WA][3], bf[WB][3];
#pragma unroll
            for (int pc = 0; pc < NPW; pc++) {
#pragma unroll
                for (int a = 0; a < WA; a++) af[a][pc] = *(const bf16x8*)&Ss[pc * A_PIECE + ((wa * WA + a) * 32 + jl) * SP + 16 * s + 8 * hl];
#pragma unroll
                for (int b = 0; b < WB; b++) bf[b][pc] = *(const bf16x8*)&Ls[pc * B_PIECE + ((wb * WB + b) * 32 + jl) * SP + 16 * s + 8 * hl];
            }
#pragma unroll
            for (int pb = NPW - 1; pb >= 0; pb--)        // smallest terms first
#pragma unroll
                for (int pa = NPW - 1; pa >= 0; pa--)
                    if (mmw_on<NP>(pa, pb)) {
#pragma unroll
                    for (int a = 0; a < WA; a++)
#pragma unroll
                        for (int b = 0; b < WB; b++)
                            acc[a][b] = mfma16<IO, NP>(af[a][pa], bf[b][pb], acc[a][b]);
                    }
        }
    }

    const int Ag_pad = p.a_tiles * TA, Bg_pad = p.b_tiles * TB;
    float* out = p.slab + ((int64_t)ks * p.G + g) * Ag_pad * Bg_pad;
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < WB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ai = a_blk + (wa * WA + a) * 32 + acc_row(r, lane), bi = b_blk + (wb * WB + b) * 32 + jl;
                out[(int64_t)ai * Bg_pad + bi] = HX ? acc[a][b][r] * out_scale : acc[a][b][r];
            }
}

}  // namespace pasta
