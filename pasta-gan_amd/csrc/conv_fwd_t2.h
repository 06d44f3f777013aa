// 3x3 stride-2 conv_transpose2d of the three-product fp16 arithmetic as ONE pass over the input lattice (round 5): every upsampling layer of
// the generator (conv2d_resample.py:104-115, the transposed convolution in front of the blur) and the input gradient of every stride-2
// convolution.  Instantiated by conv_tu_fwd_t2.hip (conv_launch.h).
#pragma once
#include "conv_fwd_bf16x6.h"

namespace pasta {

//------------------------------------------------------------------------------------
// With pad 0 (OH = 2 H + 1: every live shape) input pixel (i, j) and kernel tap (r, c) meet in output (2 i + r, 2 j + c); seen from the
// output, y[2 i + a, 2 j + b] sums the taps of parity (r & 1, c & 1) = (a, b) over the input pixels (i - (r >> 1), j - (c >> 1)): four
// parity classes with 4, 2, 2 and 1 taps, all of them reading the 2 x 2 window x[i - 1 .. i, j - 1 .. j].  The parity-pair mode of the row
// kernel (conv_fwd_bf16x6.h: one vertical parity per workgroup, one input ROW per stage, 143 - 220 TFLOP/s on the live shapes and nothing at
// all for planes below 128 x 128 with a remainder row, which ran as four per-class launches) stages every input row twice and a tap's
// weights once per 128 pixels.  Here a workgroup owns a tile of R x SEG = 8 x 32 lattice pixels and 64 output channels, ALL FOUR classes:
//   * B image of a 16-channel chunk: the (R + 1) x (SEG + 1) input pixels of the tile's windows, fetched and split once (1.16 staging units
//     per thread); the four window positions are four fragment offsets into it, read once per chunk and wave (16 ds_read_b128);
//   * A image of a chunk: the nine taps' weights, 9 x 6 x 64 sixteen-byte units = 54 KB, by LDS-DMA (global_load_lds_dwordx4: the packed
//     weight tensor has the image's layout; seven instructions per thread and chunk, no registers, no ds_write) one chunk ahead;
//   * a wave = 32 output channels x 64 pixels x 4 classes: eight accumulator tiles, 9 taps x 2 pixel blocks x 3 products = 54 MFMAs per
//     chunk on 27 + 16 fragment reads, ONE barrier per chunk (the dominant kernel: 12 MFMAs on 10 reads per barrier);
//   * eight waves (2 x 4), one workgroup per CU: 112 KB of weights + 37 KB of activations double-buffered in LDS;
//   * an output row leaves as 8-byte pairs (2 j, 2 j + 1), 256 contiguous bytes per 32 lanes, the two rows 2 i and 2 i + 1 by the same lane.
// Output row 2 H and column 2 W (OH = 2 H + 1: lattice row i = H, lattice column j = W) are EDGE TILES of the same launch -- as a launch of
// their own (conv_t2_edge_kernel: a few dozen workgroups with K loops as long as anyone's) they took 52 - 149 us in front of a main kernel
// of 123 - 190 us.  An edge tile keeps the tile's shape and changes what a tile row / column means:
//   * row edge:    tile row r = image n0 + r, tile columns = lattice columns j0 .. j0 + 31 of lattice row H: only the windows one row up
//                  exist (taps 6, 7, 8: classes (0, 0) and (0, 1)), B image slot (ir, ic) = x[n0 + ir, H - 1, j0 - 1 + ic];
//   * column edge: tile row r = image n0 + r, tile columns = lattice ROWS i0 .. i0 + 31 (up to H: the corner) of lattice column W: only the
//                  windows one column to the left exist (taps 2, 5, 8: classes (0, 0) and (1, 0)); the row above is the tile column to the left:
//                  slot (ir, ic) = x[n0 + ir - 1, i0 + ic - 1, W - 1], taps 2 and 5 at window position (0, 0), tap 8 at (0, -1).  The column
//                  itself comes from p.x2 = x[:, :, :, W - 1] gathered as [N][C][H] by t2_column_gather_kernel in front of the launch: read in
//                  place, a lane's four bytes are a cache line of their own whichever way the tile is laid (4752 line requests per chunk and
//                  workgroup: a column tile then took 36 - 75 us, as long as a regular tile with three times its MFMAs).
// The K loop, the weights and the fragment offsets are the regular tile's; the kind is uniform per workgroup.
// ISC: p.iscale[n, channel] (the styles of a modulated layer: pasta_conv2d_modulated's x * s) multiplied onto the activations between
// fetch and split, as in the other forward-type kernels.
// The DMA is a vector-memory operation whose LDS write the compiler does not see: the waits are written by hand -- `s_waitcnt vmcnt(0)`
// in front of the barrier that ends a chunk (the activation loads of the chunk have been consumed by then: nothing else is in flight).
// KIND is a template parameter of the body: with the kind as a (uniform) run-time value the branches around the window groups cut the K loop into
// basic blocks and the regular tiles ran 30 - 60 % slower (0.152 -> 0.199 ms on 256 -> 128 at 64 x 64, 0.136 -> 0.223 ms on 512 -> 256 at 32 x 32).
// SEG: tile columns, 32 (tiles of 8 x 32 lattice pixels) or 16 (16 x 16: the 16 x 16 input planes; a 32-pixel MFMA block is then two tile rows).
template <bool ISC, int KIND, int SEG>
__device__ __forceinline__ void conv_t2_body(const ConvFwdParams& p, const int n0, const int i0, const int j0) {
    constexpr int kind = KIND;
    constexpr int NP = NP_F16X3, BM = 64, R = 256 / SEG, NT = 512;
    static_assert(SEG == 32 || SEG == 16, "tiles of 8 x 32 or 16 x 16 lattice pixels");
    constexpr int IW = SEG + 1, SLOTS = (R + 1) * IW;           // B image: rows i0 - 1 .. i0 + R - 1, columns j0 - 1 .. j0 + SEG - 1
    constexpr int AUNITS = 9 * 6 * BM, APT = (AUNITS + NT - 1) / NT;
    constexpr int ABUF = APT * NT * 8;                          // 16-bit elements of an A buffer (padded to whole DMA instructions)
    constexpr int BSEG = SLOTS * 8, BBUF = 2 * 2 * BSEG;        // [piece][k-half][slot][8]
    constexpr int BUNITS = 2 * SLOTS;                           // (slot, k-half) staging units: every thread one, the first BUNITS - NT a second one
    static_assert(BUNITS > NT && BUNITS <= 2 * NT, "one or two staging units per thread");
    extern __shared__ __attribute__((aligned(16))) __bf16 t2_smem[];
    __bf16* const As = t2_smem;                         // [2][ABUF]
    __bf16* const Bs = t2_smem + 2 * ABUF;              // [2][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hl = lane >> 5, jl = lane & 31;
    const int g = blockIdx.z;
    const int o_blk = blockIdx.y * BM;
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / 16;

    // ---- weights: sixteen-byte unit u = tid + NT j of a chunk's image -> (tap, piece * 2 + k-half, output row)
    const int64_t a_chunk = (int64_t)6 * p.Og_pad * 8;             // elements of one packed 16-channel chunk of one tap
    const __bf16* const wb = (const __bf16*)p.wp + (int64_t)g * 9 * NC * a_chunk + (int64_t)o_blk * 8;
    unsigned a_off[APT];                                // byte offsets from the chunk's (uniform) base: scalar base + 32-bit lane offset addressing
#pragma unroll
    for (int j = 0; j < APT; j++) {
        int u = tid + NT * j;
        u = u < AUNITS ? u : AUNITS - 1;                // past the image: a valid address, the unit lands in the buffer's padding
        const int tap = u / (6 * BM), rem = u - tap * (6 * BM);
        const int seg = rem / BM, within = rem - seg * BM;
        a_off[j] = ((unsigned)tap * (unsigned)NC * (unsigned)a_chunk + (unsigned)(seg * p.Og_pad + within) * 8u) * 2u;
    }
    auto glds_a = [&](int cc, int buf) {
        const char* const wc = (const char*)(wb + (int64_t)cc * a_chunk);
#pragma unroll
        for (int j = 0; j < APT; j++) {
            __bf16* dst = As + buf * ABUF + (wave * 64 + NT * j) * 8;        // the wave's base: the hardware adds lane * 16 bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wc + a_off[j]), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    if constexpr (kind == 0) glds_a(0, 0);
    // Edge tiles: the units of the THREE taps they multiply (row edge 6, 7, 8; column edge 2, 5, 8), 2.25 per thread, through registers and two
    // chunks ahead (below): a third of a regular tile's bytes and MFMAs, and as a copy of the regular tile's schedule -- everything of the next
    // chunk fetched at the head of a chunk and waited for at its end -- still three quarters of its time (the memory latency once per chunk).
    constexpr int EUNITS = 3 * 6 * BM, EPT = (EUNITS + NT - 1) / NT;
    unsigned e_off[EPT];
    int e_lds[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int u = tid + NT * j;
        const int uu = u < EUNITS ? u : 0;
        const int t3 = uu / (6 * BM), rem = uu - t3 * (6 * BM);
        const int tap = kind == 1 ? 6 + t3 : 3 * t3 + 2;
        const int seg = rem / BM, within = rem - seg * BM;
        e_off[j] = ((unsigned)tap * (unsigned)NC * (unsigned)a_chunk + (unsigned)(seg * p.Og_pad + within) * 8u) * 2u;
        e_lds[j] = u < EUNITS ? (tap * (6 * BM) + rem) * 8 : -1;
    }
    constexpr int SETS = kind == 0 ? 1 : 2;             // register sets of staged operands: chunks in flight
    u32x4 ea[SETS][EPT];
    auto load_a = [&](int cc, int set) {
        const char* const wc = (const char*)(wb + (int64_t)cc * a_chunk);
#pragma unroll
        for (int j = 0; j < EPT; j++) ea[set][j] = *(const u32x4*)(wc + e_off[j]);
    };
    auto store_a = [&](int set, int buf) {
#pragma unroll
        for (int j = 0; j < EPT; j++)
            if (e_lds[j] >= 0) *(u32x4*)(As + buf * ABUF + e_lds[j]) = ea[set][j];
    };

    // ---- activations: staging unit k of this thread -> (slot, k-half); pixel offset and validity
    const char* const xb = kind == 2 ? (const char*)p.x2 : (const char*)p.x;      // (uniform; the planner keeps the tensor below 2^30 elements: 32-bit byte offsets)
    const unsigned cstride = kind == 2 ? (unsigned)p.H * 4u : (unsigned)HW * 4u;      // bytes from a channel to the next
    unsigned u_pix[2];
    int u_half[2], u_lds[2], u_isc[2];
    bool u_ok[2];
    const bool second = tid < BUNITS - NT;              // wave-uniform for all but one wave (82 threads)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int u = (k == 0 || second) ? tid + NT * k : 0;
        const int half = u >= SLOTS ? 1 : 0, slot = u - half * SLOTS;
        const int ir = slot / IW, ic = slot - ir * IW;
        int img = n0, y = i0 - 1 + ir, x = j0 - 1 + ic;
        bool in_tile = true;
        if (kind == 1) { img = n0 + ir; y = p.H - 1; in_tile = ir < R; }
        if (kind == 2) { img = n0 + ir - 1; y = i0 + ic - 1; x = p.W - 1; in_tile = ir >= 1; }
        u_ok[k] = (k == 0 || second) && in_tile && img < p.N && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        u_isc[k] = u_ok[k] ? img * p.Cin + g * p.Ig : 0;
        u_pix[k] = !u_ok[k] ? 0u : kind == 2 ? ((unsigned)u_isc[k] * (unsigned)p.H + (unsigned)y) * 4u : ((unsigned)u_isc[k] * (unsigned)HW + (unsigned)(y * p.W + x)) * 4u;
        u_half[k] = half;
        u_lds[k] = (half * SLOTS + slot) * 8;
    }
    float sb[SETS][2][8], sc[SETS][2][ISC ? 8 : 1];
    int nv[SETS][2];
    auto load_units = [&](int cc, int set = 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !second) continue;
            const int c0 = cc * 16 + u_half[k] * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = c0 + j < p.Ig ? c0 + j : p.Ig - 1;
                sb[set][k][j] = io_ld<IO_F32>(xb, u_pix[k] + (unsigned)c * cstride);
                if constexpr (ISC) sc[set][k][j] = p.iscale[u_isc[k] + c];
            }
            nv[set][k] = u_ok[k] ? p.Ig - c0 : 0;
        }
    };
    load_units(0);

    float x_scale, out_scale;
    scale_from_amax(amax_of_parts(p.x_amax), x_scale, out_scale);
    auto store_units = [&](int buf, int set = 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !second) continue;
            uint32_t q1[4], q2[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v0 = sb[set][k][2 * j], v1 = sb[set][k][2 * j + 1];
                if constexpr (ISC) { v0 *= sc[set][k][2 * j]; v1 *= sc[set][k][2 * j + 1]; }
                v0 = 2 * j < nv[set][k] ? v0 : 0.f;
                v1 = 2 * j + 1 < nv[set][k] ? v1 : 0.f;
                f16_split2(v0 * x_scale, v1 * x_scale, q1[j], q2[j]);
            }
            __bf16* const bd = Bs + buf * BBUF + u_lds[k];
            *(uint4*)bd = make_uint4(q1[0], q1[1], q1[2], q1[3]);
            *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        }
    };

    f32x16 acc[4][2];                                   // [class 2 a + b][pixel block]
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][b][r] = 0.f;

    // Fragment addresses: everything that depends on the lane in ONE base per operand (A: the lane's output row and k-half; B: the slot of
    // the lane's pixel of block b at window position (0, 0) -- tile row wn * 2 + b, column jl -- and its k-half), everything else -- buffer,
    // tap, piece, window position -- literal byte offsets (ds_read's immediate field; loop-invariant address registers are what spills here)
    const char* const a_lane = (const char*)(As + ((hl * BM) + wm * 32 + jl) * 8);
    const char* b_lane[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int t = (wn * 2 + b) * 32 + jl;           // this lane's pixel of block b: tile row t / SEG, column t % SEG
        b_lane[b] = (const char*)(Bs + (hl * SLOTS + (t / SEG + 1) * IW + t % SEG + 1) * 8);
    }

    auto compute = [&](int buf, bool stage) {
        bf16x8 bh[2], bl[2];
        auto window = [&](int off) {
#pragma unroll
            for (int b = 0; b < 2; b++) {
                bl[b] = *(const bf16x8*)(b_lane[b] + (buf * BBUF + 2 * BSEG + off * 8) * 2);
                bh[b] = *(const bf16x8*)(b_lane[b] + (buf * BBUF + off * 8) * 2);
            }
        };
        auto tap = [&](int t, int cls) {                // both literals: three products, smallest terms first (h'' l', l h, h h)
            const char* const at = a_lane + (buf * ABUF + t * (6 * BM * 8)) * 2;
            const bf16x8 a2 = *(const bf16x8*)(at + (4 * BM * 8) * 2);
            const bf16x8 a1 = *(const bf16x8*)(at + (2 * BM * 8) * 2);
            const bf16x8 a0 = *(const bf16x8*)(at);
#pragma unroll
            for (int b = 0; b < 2; b++) acc[cls][b] = mfma16<IO_F32, NP>(a2, bl[b], acc[cls][b]);
#pragma unroll
            for (int b = 0; b < 2; b++) acc[cls][b] = mfma16<IO_F32, NP>(a1, bh[b], acc[cls][b]);
#pragma unroll
            for (int b = 0; b < 2; b++) acc[cls][b] = mfma16<IO_F32, NP>(a0, bh[b], acc[cls][b]);
        };
        // tap (r, c) = 3 r + c: class (r & 1, c & 1), window position (-(r >> 1), -(c >> 1)); an edge tile has one row / column of windows
        if (kind == 2) { window(-1); tap(8, 0); window(0); tap(2, 0); tap(5, 2); }      // (the column edge's own map: see the head of the file)
        if (kind != 2) { window(-IW - 1); tap(8, 0); window(-IW); tap(6, 0); tap(7, 1); }
        if (kind == 0) { window(-1);  tap(2, 0); tap(5, 2); }
        // the next chunk's activations: split and stored HERE, between the MFMA groups (their loads were issued at the head of the chunk, five
        // taps ago), so that this wave's vector instructions run beside the other wave's MFMAs instead of behind everybody's
        if (stage) store_units(buf ^ 1);
        if (kind == 0) { window(0);   tap(0, 0); tap(1, 1); tap(3, 2); tap(4, 3); }
    };

    if constexpr (kind != 0) {
        // edge tile: chunk cc + 2 is fetched into register set cc & 1 while chunk cc is multiplied and chunk cc + 1 -- fetched a chunk ago -- is
        // split and stored: every load has a whole chunk to land (ordinary loads throughout: the compiler counts the waits)
        load_a(0, 0);
        if (NC > 1) { load_units(1, 1); load_a(1, 1); }
        store_units(0, 0); store_a(0, 0);
        __syncthreads();
        for (int c = 0; c < NC; c += 2) {
#pragma unroll
            for (int par = 0; par < 2; par++) {
                const int cc = c + par;
                if (cc < NC) {                           // (uniform)
                    if (cc + 2 < NC) { load_units(cc + 2, par); load_a(cc + 2, par); }
                    compute(par, false);
                    if (cc + 1 < NC) { store_units(par ^ 1, par ^ 1); store_a(par ^ 1, par ^ 1); }
                    __syncthreads();
                }
            }
        }
    } else {
    // prologue: the first chunk's activations; the barrier's fence waits for the DMA of its weights
    store_units(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < NC; c += 2) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            const int cc = c + par;
            if (cc < NC) {                               // (uniform)
                const bool more = cc + 1 < NC;
                if (more) {
                    // (the ordinary loads FIRST: behind a DMA hipcc puts `s_waitcnt vmcnt(0)` in front of the first write to a register that an
                    // earlier load returned into -- the DMA just issued would be waited for at the head of every chunk: 44 us of 196 on 256 -> 128 at 64 x 64)
                    load_units(cc + 1);
                    asm volatile("" ::: "memory");
                    glds_a(cc + 1, par ^ 1);
                }
                compute(par, more);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }
    }

    // back to the operands' units: 1 / S_x for the tile, 1 / S_w per weight row (p.w_rowinv, written by the packing kernel)
    const float* const wri = p.w_rowinv + (int64_t)g * p.Og_pad + o_blk + wm * 32;
    float ws[16];
#pragma unroll
    for (int r = 0; r < 16; r++) ws[r] = wri[acc_row(r, lane)] * out_scale;
    struct __attribute__((packed, aligned(4))) Pair { float even, odd; };
    const int64_t OHW = (int64_t)p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int t = (wn * 2 + b) * 32 + jl;
        const int tr = t / SEG, tc = t % SEG;           // tile row / column of this lane's pixel
        const int img = kind == 0 ? n0 : n0 + tr;
        const int i = kind == 1 ? p.H : kind == 2 ? i0 + tc : i0 + tr, j = kind == 2 ? p.W : j0 + tc;
        if (img >= p.N || 2 * i >= p.OH) continue;      // (edge tiles: images / rows past the end)
        float* const yb = (float*)p.y + ((int64_t)img * p.Cout + (int64_t)g * p.Og) * OHW + (int64_t)(2 * i) * p.OW + 2 * j;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = o_blk + wm * 32 + acc_row(r, lane);
            if (o < p.Og) {
                float* const yo = yb + (int64_t)o * OHW;
                if (kind == 2) {                        // column 2 W: classes (0, 0) and (1, 0)
                    yo[0] = acc[0][b][r] * ws[r];
                    if (i < p.H) yo[p.OW] = acc[2][b][r] * ws[r];
                } else {
                    Pair v0;
                    v0.even = acc[0][b][r] * ws[r]; v0.odd = acc[1][b][r] * ws[r];
                    *(Pair*)yo = v0;
                    if (kind == 0) {
                        Pair v1;
                        v1.even = acc[2][b][r] * ws[r]; v1.odd = acc[3][b][r] * ws[r];
                        *(Pair*)(yo + p.OW) = v1;
                    }
                }
            }
        }
    }
}

template <bool ISC, int SEG>
__global__ __launch_bounds__(512, 1) void conv_t2_f16x3_kernel(ConvFwdParams p) {
    constexpr int R = 256 / SEG;
    const int cblocks = p.W / SEG, tpi = (p.H / R) * cblocks;
    // tile kind (uniform): 0 regular, 1 row edge (lattice row H), 2 column edge (lattice column W).  The edge tiles come FIRST: they are a third
    // of a regular tile's work, and the CUs that start with one pick up regular tiles behind it -- at the end of the grid they would be a wave
    // of workgroups of their own behind a grid that fills the chip exactly.
    int bx = blockIdx.x;
    const int rb = p.H / SEG + 1;                       // blocks of 32 lattice rows of the column edge: rows 0 .. H
    const int row_tiles = p.OH > 2 * p.H ? ((p.N + R - 1) / R) * cblocks : 0;
    const int col_tiles = p.OW > 2 * p.W ? ((p.N + R - 1) / R) * rb : 0;
    if (bx < row_tiles) {
        conv_t2_body<ISC, 1, SEG>(p, (bx / cblocks) * R, p.H, (bx % cblocks) * SEG);
    } else if (bx < row_tiles + col_tiles) {
        bx -= row_tiles;
        conv_t2_body<ISC, 2, SEG>(p, (bx / rb) * R, (bx % rb) * SEG, p.W);
    } else {
        bx -= row_tiles + col_tiles;
        const int n0 = bx / tpi, t_in = bx - n0 * tpi;
        conv_t2_body<ISC, 0, SEG>(p, n0, (t_in / cblocks) * R, (t_in % cblocks) * SEG);
    }
}

// x[:, :, :, W - 1] as [planes][H]: the column-edge tiles' operand (one thread per element; the reads are a cache line each, spread over the chip)
__global__ __launch_bounds__(256) void t2_column_gather_kernel(const float* __restrict__ x, float* __restrict__ col, int64_t planes, int H, int W) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= planes * H) return;
    const int64_t plane = e / H;
    const int i = (int)(e - plane * H);
    col[e] = x[(plane * H + i) * W + W - 1];
}

// Does the kernel take the MAIN lattice of this launch (pad 0, OH in {2 H, 2 H + 1})?  Planes of 8 x 32 tiles, or of 16 x 16 tiles.
static bool conv_t2_shape_ok(int H, int W) { return (H % 8 == 0 && W % 32 == 0) || (H % 16 == 0 && W % 16 == 0); }

template <bool ISC, int SEG>
static void launch_conv_t2_seg(const ConvFwdParams& p, hipStream_t s) {
    constexpr int R = 256 / SEG, NT = 512, APT = (9 * 6 * 64 + NT - 1) / NT, SLOTS = (R + 1) * (SEG + 1);
    constexpr size_t lds = (size_t)(2 * APT * NT * 8 + 2 * 2 * 2 * SLOTS * 8) * sizeof(__bf16);
    // the edge tiles of lattice row H (R images x SEG columns each) and of lattice column W (R images x SEG rows, rows 0 .. H), then the regular tiles
    const int64_t tiles = (int64_t)p.N * (p.H / R) * (p.W / SEG) + (p.OH > 2 * p.H ? (int64_t)((p.N + R - 1) / R) * (p.W / SEG) : 0) +
                          (p.OW > 2 * p.W ? (int64_t)((p.N + R - 1) / R) * (p.H / SEG + 1) : 0);
    const dim3 grid((unsigned)tiles, (unsigned)((p.Og + 63) / 64), (unsigned)p.G);
    PASTA_SET_LDS((conv_t2_f16x3_kernel<ISC, SEG>), lds);
    hipLaunchKernelGGL((conv_t2_f16x3_kernel<ISC, SEG>), grid, dim3(NT), lds, s, p);
}

// (p.x2: N * C_in * H floats of workspace for the gathered column, used when OW = 2 W + 1)
static void launch_conv_t2(const ConvFwdParams& p, hipStream_t s) {
    if (p.OW > 2 * p.W) {
        const int64_t planes = (int64_t)p.N * p.Cin;
        hipLaunchKernelGGL(t2_column_gather_kernel, dim3((unsigned)((planes * p.H + 255) / 256)), dim3(256), 0, s, (const float*)p.x, (float*)p.x2, planes, p.H, p.W);
    }
    const bool wide = p.H % 8 == 0 && p.W % 32 == 0;
    if (p.iscale) { if (wide) launch_conv_t2_seg<true, 32>(p, s); else launch_conv_t2_seg<true, 16>(p, s); }
    else          { if (wide) launch_conv_t2_seg<false, 32>(p, s); else launch_conv_t2_seg<false, 16>(p, s); }
}

}  // namespace pasta
