// Two-dimensional pixel tiles for the row-reuse forward kernel.  Instantiated by conv_tu_rows2d_*.hip (conv_launch.h).
#pragma once
#include "conv_fwd_bf16x6.h"
#include <type_traits>

namespace pasta {

//------------------------------------------------------------------------------------
// conv_fwd_rows_bf16x6_kernel fetches, splits and stores one input row per (kernel row dy, 16-channel chunk): three rows of
// staging work per output row, and the output rows above and below (other workgroups) stage the same input rows again.
// Here a pixel tile is R output rows x SEG = BN / R columns, and the B image of a chunk is the R + 2 input rows the tile
// touches (one halo column on either side): (R + 2) / R rows of staging per output row instead of 3 -- 1.5 for R = 4 --
// with the same 9 x 24 MFMAs per chunk and wave.  (The weight gradient gained 8 - 17 % from the same reuse, and the
// timing-only ablation of the row kernel puts the activation fetch + split + LDS stores at a quarter of its time.)
//   B image: [piece][k-half][slot][8 bf16], slot = i * (SEG + 2) + j for image row i (input row p0 + ymin + i) and image
//            column j (input column q0 + d0 + j).  The fragment of tap (dy, dx) for the 32 pixels of tile row r starts at
//            slot (r + dy - ymin) * (SEG + 2) + (dx - d0) + 32 * (column block of the fragment).
//   Staging: the 2 * SLOTS (slot, k-half) units of the next chunk are dealt round-robin to the threads (2 units each for
//            R = 4, SEG = 32); unit k is fetched at step 3k of the current chunk and split + stored at step 3k + 3 (four units of the
//            64 x 256 tile: steps 2k and 2k + 2), so at most two eight-register sets are in flight and every fetch has two
//            or three steps (48 - 72 MFMAs) to land.
//   Weights: per step (tap) as in the row kernel, fetched two steps ahead into one of two register sets.
// One loop trip = two chunks = 18 steps, so that every buffer index is a literal and the control flow around memory
// operations is static (the s_waitcnt counters then let a fetch stay in flight across steps).
// NP / IO as in conv_fwd_bf16x6_kernel: bf16 pieces per operand and the storage type of x / y / res.
// ISC: p.iscale[n, channel] (the styles of a modulated convolution) multiplied onto the activations between fetch and split.
// NT = threads per workgroup: 256 (four waves; two workgroups per CU) or 512 (eight waves on a 128 x 256 tile, one workgroup per
// CU: the weights of a step are fetched and stored once for 256 pixels instead of once for 128).
// Measured and dropped (round 3, three-product arithmetic, profiles/r3_ab_tap_pairs.txt): TWO taps per barrier on the eight-wave
// tile (four weight buffers; the weights of the next pair fetched when a pair begins, stored when it ends; 246 VGPRs): 2 % SLOWER
// on every live shape (248 / 304 / 318 / 324 against 255 / 310 / 325 / 328 TFLOP/s) although 34 % of the wave cycles are parked
// at s_waitcnt / s_barrier -- the barrier count is not what parks them.
// Measured and dropped (round 4, profiles/r4_ab_fragment_prefetch.txt; "ab" = without): the ACTIVATION fragments of tap S + 1 read at the end of
// step S, in front of the barrier (the B image of a chunk is complete when the chunk begins), so that only the six weight-fragment reads stand
// between the barrier and the first MFMA group: 327.0 -> 324.4 TFLOP/s in the training step, 316.5 -> 313.7 on 256 -> 128 at 128 x 128: 0.8 % SLOWER.
// Measured and dropped (round 4, profiles/r4_ab_rows2d_64x256_eight_waves.txt): EIGHT waves of 32 rows x 64 pixels on the 64 x 256 tile (1.33 staging units
// per thread and chunk instead of 2.66; 168 VGPRs, one workgroup per CU): 64 -> 64 at 256 x 256 0.400 -> 0.460 ms forward, 0.360 -> 0.431 ms input gradient,
// step 153.2 -> 155.2 ms -- a B fragment then serves one MFMA instead of two, and one workgroup per CU leaves nothing to overlap with.
// Also measured and dropped (round 3, profiles/r3_ab_wave128.txt, r3_ab_rows2d_pipe*.txt, r3_pipe_ablation.txt; the kernel is kept, out of
// the build; deleted in round 4, last in commit 7b449f3 as tools/experiments/conv_fwd_rows2d_pipe.h): FOUR waves of 64 x 128 outputs on the same tile (14 fragment reads per
// 24 MFMAs instead of 10 per 12), one wave per SIMD -- equal to the eight waves within 2 % on every shape, with or without the fragments of
// step g + 1 read behind the MFMAs of step g (two fragment sets, 253 + 128 registers, the order within a step pinned by
// sched_group_barrier).  Its timing-only instances on the 512 -> 512 layer at 32 x 32: MFMAs + barriers alone 509 TFLOP/s, + fragment
// reads 421, + staging (everything) 339: the chip is at its power cap (1.9 GHz), and every LDS byte and fetch next to the MFMAs costs
// clock, however well it is hidden in the schedule.
// XP (round 5): x is PASTA_LAYOUT_PIECES16 (pieces.hip) -- the producer wrote the operand's fp16 pieces h | l' as 16-byte units of eight channels,
// [N][C/8][H][piece][W] -- so a staging unit is TWO 16-byte loads (lanes along a row: 1 KB runs) straight into the B image: eight dword loads,
// the scale and the split (some forty vector instructions per unit) are the producer's, done once per tensor instead of once per consuming tile.
// GA (round 5): the WEIGHTS of a step go from L2 straight into their LDS buffer (global_load_lds_dwordx4: the packed tensor already has the
// image's layout, one 16-byte unit per lane, a wave's units contiguous) one step ahead, instead of two steps ahead into one of two register
// sets and from there to LDS: no ds_write_b128 pass in front of the barrier, 24 registers fewer.  The waits are counted by hand (the DMA is a
// vector-memory operation the compiler does not see as a write to LDS): at the end of a step `s_waitcnt vmcnt(K)`, K = the activation loads
// issued BEHIND the DMA in that step (a compiler barrier pins that order), so that those stay in flight across the barrier and the DMA has
// landed; in the steps that split and store a unit the DMA is issued behind the split (hipcc waits vmcnt(0) at the first use of an ordinary
// load while a DMA is in flight: nothing else is outstanding there).
template <int BM, int BN, int R, int NP = 3, int IO = IO_F32, bool ISC = false, int NT = 256, bool XP = false, bool GA = false>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void conv_fwd_rows2d_bf16x6_kernel(ConvFwdParams p) {
    static_assert(!GA || (!ISC && IO == IO_F32), "weights by LDS-DMA: the counted waits are written for the plain fp32-storage launches");
    static_assert(IO == IO_F32 || NP == 1, "16-bit storage: the element is the operand, one product");
    static_assert(!XP || (NP == NP_F16X3 && IO == IO_F32 && !ISC), "operand pieces: the three-product arithmetic's, plain launches");
    static_assert(!ISC || ((NP == 3 || NP == NP_F16X3) && IO == IO_F32), "the input scale rides in the fp32-equivalent staging");
    constexpr bool HX = Arith<NP>::f16x3;               // PASTA_MATH_F16X3: fp16 pieces, three products (conv_common.h)
    constexpr int NPA = Arith<NP>::npa, NPB = Arith<NP>::npb;
    constexpr unsigned ES = io_size<IO>::value;
    constexpr int WMT = 2, WNT = 2, KC = 16;
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == NT / 64, "one wave per 64 x 64 sub-tile");
    constexpr int SEG = BN / R, SW = SEG + 2, SLOTS = (R + 2) * SW;
    static_assert(SEG % 32 == 0, "a fragment's 32 pixels lie in one tile row");
    constexpr int UNITS = 2 * SLOTS, UPT = (UNITS + NT - 1) / NT;          // (slot, k-half) staging units; per thread
    static_assert(UPT <= 4, "at most four staging units per thread and chunk");
    // unit k of the next chunk is fetched at step LSTEP(k) and split + stored at step TSTEP(k)
    constexpr int USTRIDE = UPT <= 3 ? 3 : 2;
    constexpr int AUNITS = 2 * NPA * BM, APT = (AUNITS + NT - 1) / NT;
    constexpr int ABUF = APT * NT * 8, BSEG = SLOTS * 8, BBUF = 2 * NPB * BSEG;      // 16-bit elements
    extern __shared__ __attribute__((aligned(16))) __bf16 rows2d_smem[];
    __bf16* const As = rows2d_smem;                     // [2][ABUF]
    __bf16* const Bs = rows2d_smem + 2 * ABUF;          // [2][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    const int P = p.cls[0].P, Q = p.cls[0].Q;
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    // (32-bit: NC <= a few hundred chunks, ksplit <= 32; the 64-bit quotients this was written with are ~150 scalar instructions each, in front of
    // the first load of a workgroup that is alone on its CU)
    const int c_first = (int)(((unsigned)NC * (unsigned)ks) / (unsigned)p.ksplit);
    const int nchunks = (int)(((unsigned)NC * (unsigned)(ks + 1)) / (unsigned)p.ksplit) - c_first;
    // tile -> (image, row block, column block)
    const int cblocks = Q / SEG, tpi = (P / R) * cblocks;
    // Workgroups go round-robin to the eight XCDs (one L2 each).  On the eight-wave tile an XCD receives CONSECUTIVE pixel tiles -- the
    // neighbours that share halo rows -- instead of every eighth: +2 - 3.5 % on every 128-channel shape, dominant kernel 326 -> 336
    // TFLOP/s, step 155.3 -> 154.4 ms (same box, twice; profiles/r3_ab_xcd_order.txt).  The same order on the four-wave 64 x 256 tile:
    // -2.4 %; on the base kernel: nothing (r3_ab_xcd_order2.txt) -- not applied there.  Also measured: the output-channel tiles of a pixel
    // tile back to back on one XCD (layers with more than 128 output channels): 315 -> 303 TFLOP/s on 128 -> 256, nothing on 256 -> 256 and
    // 512 -> 512 (r3_ab_xcd_otiles.txt).  PASTA_XCD_ORDER=0 switches it off.
    unsigned bx = blockIdx.x;
    if (NT == 512 && p.xcd_order && (gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const int n_img = bx / tpi;
    const int t_in = bx - n_img * tpi;
    const int p0 = (t_in / cblocks) * R, q0 = (t_in % cblocks) * SEG;
    const int ymin = p.rows_y0, d0 = p.rows_d0;

    const char* const xbytes = (const char*)p.x;
    const unsigned xb_off = (unsigned)(((int64_t)n_img * p.Cin + (int64_t)g * p.Ig) * HW) * ES;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int64_t a_chunk = (int64_t)6 * p.Og_pad * 8;             // bf16 elements of one packed 16-channel chunk

    // ---- staging units of this thread: unit u = tid + NT k -> (slot, k-half); pixel byte offset (channel 0) and validity
    unsigned u_pix[UPT];
    bool u_ok[UPT];
    int u_half[UPT], u_lds[UPT];
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        const int u = tid + NT * k;
        const int half = u >= SLOTS ? 1 : 0;
        int slot = u - half * SLOTS;
        const bool real = u < UNITS;
        slot = real ? slot : 0;
        const int i = slot / SW, j = slot - i * SW;
        const int y = p0 + ymin + i, x = q0 + d0 + j;
        u_ok[k] = real && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        if constexpr (XP) u_pix[k] = (unsigned)((n_img * (p.Cin >> 3)) * HW) * 32u + (u_ok[k] ? (unsigned)(2 * y * p.W + x) * 16u : 0u);      // octet 0, piece h
        else u_pix[k] = u_ok[k] ? xb_off + (unsigned)(y * p.W + x) * ES : xb_off;
        u_half[k] = half;
        u_lds[k] = real ? (half * SLOTS + slot) * 8 : -1;           // element offset inside a piece of the B image; -1: no unit
    }

    float sb0[8], sb1[8];                               // the two staging register sets
    float sc0[ISC ? 8 : 1], sc1[ISC ? 8 : 1];           // ISC: the input scales of their channels
    const float* const isb = ISC ? p.iscale + (int64_t)n_img * p.Cin + (int64_t)g * p.Ig : nullptr;
    int nv0 = 0, nv1 = 0;                               // valid channels of the set (0: pixel outside / chunk past the end)
    auto load_unit = [&](int k, int cc, bool real_chunk, float (&sb)[8], float (&sc)[ISC ? 8 : 1], int& nv) {
        const int c0 = cc * KC + u_half[k] * 8;
        const int last = p.Ig - 1;
        if constexpr (XP) {
            const int c8 = c0 < p.Ig ? c0 >> 3 : 0;             // past the last octet: a valid address, unused data
            const char* src = xbytes + u_pix[k] + (unsigned)c8 * (unsigned)HW * 32u;
            const u32x4 hq = *(const u32x4*)src, lq = *(const u32x4*)(src + (unsigned)p.W * 16u);
#pragma unroll
            for (int j = 0; j < 4; j++) { const uint32_t hj = hq[j], lj = lq[j]; sb[j] = __builtin_bit_cast(float, hj); sb[4 + j] = __builtin_bit_cast(float, lj); }      // (a bit_cast of the vector ELEMENT itself reads element 0: hipcc 7.2)
            nv = (u_ok[k] && real_chunk && c0 < p.Ig) ? 8 : 0;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = c0 + j < last ? c0 + j : last;
            sb[j] = io_ld<IO>(xbytes, u_pix[k] + (unsigned)c * (unsigned)HW * ES);
            if constexpr (ISC) sc[j] = isb[c];
        }
        nv = (u_ok[k] && real_chunk) ? p.Ig - c0 : 0;
    };
    // ---- weights: tap slabs and the fetch position (two steps ahead of the multiplication)
    const __bf16* wtap[9];
#pragma unroll
    for (int t = 0; t < 9; t++) wtap[t] = wb + (int64_t)p.tap_slab[t] * NC * a_chunk;
    float4 areg0, areg1, areg2, breg0, breg1, breg2;
    auto load_a = [&](int tap, int cc, int set) {              // tap and set are literals at every call
        const int ccl = cc < NC ? cc : NC - 1;          // past the end of the K range (or of an empty slice): a valid address, unused data
        const __bf16* wt = wtap[tap] + (int64_t)ccl * a_chunk;
        auto unit = [&](int j) {
            int e = tid + NT * j;
            if (NT * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;
            const int seg = e / BM, within = e - seg * BM;
            return *(const float4*)(wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8);
        };
        if (set == 0) { areg0 = unit(0); if (APT > 1) areg1 = unit(1); if (APT > 2) areg2 = unit(2); }
        else          { breg0 = unit(0); if (APT > 1) breg1 = unit(1); if (APT > 2) breg2 = unit(2); }
    };
    auto glds_a = [&](int tap, int cc, int buf) {              // GA: the weights of (tap, chunk cc) into A buffer `buf`
        const int ccl = cc < NC ? cc : NC - 1;
        const __bf16* wt = wtap[tap] + (int64_t)ccl * a_chunk;
#pragma unroll
        for (int j = 0; j < APT; j++) {
            int e = tid + NT * j;
            if (NT * (j + 1) > AUNITS) e = e < AUNITS ? e : AUNITS - 1;       // past the image: a valid address into the buffer's padding
            const int seg = e / BM, within = e - seg * BM;
            const __bf16* src = wt + ((int64_t)seg * p.Og_pad + o_blk + within) * 8;
            __bf16* dst = As + buf * ABUF + (wave * 64 + NT * j) * 8;        // the wave's base: the hardware adds lane * 16 bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // The first trip to memory is issued HERE, in front of the rest of the set-up (operand scale, accumulators, fragment slots, tap offsets:
    // several hundred mostly scalar instructions that the compiler otherwise places in front of the first load -- 680 instructions on the
    // eight-wave tile, whose workgroup is alone on its CU: nothing hides them): the weights of the first tap and every unit of the first
    // chunk (ONE trip to memory in front of the K loop instead of UPT; the accumulators are not live yet: the register sets are free).
    if constexpr (GA) glds_a(0, c_first, 0); else load_a(0, c_first, 0);
    float fb[UPT][8], fc[UPT][ISC ? 8 : 1];
    int fnv[UPT];
#pragma unroll
    for (int k = 0; k < UPT; k++) load_unit(k, c_first, nchunks > 0, fb[k], fc[k], fnv[k]);

    // PASTA_MATH_F16X3: power-of-two scales of the two operands (every wave reduces the partial maxima itself)
    float x_scale = 1.f, out_scale = 1.f;
    if constexpr (HX) {
        float sx, isx;
        scale_from_amax(amax_of_parts(p.x_amax), sx, isx);
        x_scale = sx; out_scale = isx;                 // the weight rows carry their own scales: p.w_rowinv, applied per output row in the epilogue
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    uint32_t q1[4], q2[4], q3[4];
    auto split_pair = [&](const float (&sb)[8], const float (&sc)[ISC ? 8 : 1], int nv, int j) {
        if constexpr (XP) {
            q1[j] = nv ? __builtin_bit_cast(uint32_t, sb[j]) : 0u;
            q2[j] = nv ? __builtin_bit_cast(uint32_t, sb[4 + j]) : 0u;
            return;
        }
        float v0 = sb[2 * j], v1 = sb[2 * j + 1];
        if constexpr (ISC) { v0 *= sc[2 * j]; v1 *= sc[2 * j + 1]; }
        if (nv < 8) {
            v0 = 2 * j < nv ? v0 : 0.f;
            v1 = 2 * j + 1 < nv ? v1 : 0.f;
        }
        if constexpr (HX) {
            f16_split2(v0 * x_scale, v1 * x_scale, q1[j], q2[j]);
            return;
        }
        f32x2 v = {v0, v1};
        uint32_t w = io_pack2<IO>(v0, v1);
        q1[j] = w;
        if constexpr (NP >= 2) {
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            q2[j] = w;
        }
        if constexpr (NP >= 3) {
            v0 -= __builtin_bit_cast(float, w << 16);
            v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
            PASTA_KEEP_SCALAR(v0);
            v = f32x2{v0, v1};
            q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        }
    };
    auto store_unit = [&](int k, int bbuf) {
        if (u_lds[k] >= 0) {
            __bf16* bd = Bs + bbuf * BBUF + u_lds[k];
            *(uint4*)(bd) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
            if constexpr (NPB >= 2) *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
            if constexpr (NPB >= 3) *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
        }
    };

    auto store_a = [&](int buf, int set) {
        __bf16* d = As + buf * ABUF;
        *(float4*)&d[tid * 8] = set ? breg0 : areg0;
        if (APT > 1) *(float4*)&d[(tid + NT) * 8] = set ? breg1 : areg1;
        if (APT > 2) *(float4*)&d[(tid + 2 * NT) * 8] = set ? breg2 : areg2;
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
    int fslot[WNT];                                  // slot of this lane's pixel of B fragment b for the tap (ymin, d0)
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int t = (wn * WNT + b) * 32 + jl;
        const int r = t / SEG, c = t - r * SEG;
        fslot[b] = r * SW + c;
    }
    // slot offset of tap t = 3 * dyi + dxi: (dy - ymin) rows, (dx - d0) columns (scalars)
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; t++) toff[t] = (p.tap_dy[t] - ymin) * SW + (p.tap_dx[t] - d0);

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

    // One step = one tap of one chunk: 24 MFMAs in six groups.  S (0..8: tap within the chunk) and PAR (parity of the chunk
    // within the trip) are literals; cc = the chunk being multiplied.
    //   weights:  step g = 9 * chunk + S multiplies A buffer g & 1, stores the weights of step g + 1 (fetched by step g - 1
    //             into register set (g + 1) & 1) and fetches those of step g + 2 into set g & 1.
    //   staging:  unit k of chunk cc + 1 is fetched at step 3k into set k & 1 and split + stored at step 3k + 3 (the last
    //             one at step 8 when it would fall off the chunk).
    auto step = [&](const int S, const int PAR, int cc) {
        const int gpar = (PAR * 9 + S) & 1;              // parity of the global step index within the trip (18 steps: even)
        const bool next_real = cc + 1 < c_first + nchunks;
        // which unit is split and stored in this step (literal): k with 3k + 3 == S, or the last unit at S == 8
        constexpr int NOUNIT = -1;
        int ku = NOUNIT;
#pragma unroll
        for (int k = 0; k < UPT; k++)
            if (S == (USTRIDE * (k + 1) < 8 ? USTRIDE * (k + 1) : 8)) ku = k;
        int behind = 0;                                  // GA: vector-memory loads issued behind the DMA in this step
        auto fetches = [&]() {
            // fetches first: they are the oldest outstanding loads when the split of a later step waits for them
            if constexpr (GA) {
                if (S + 1 < 9) glds_a(S + 1, cc, gpar ^ 1); else glds_a(0, cc + 1, gpar ^ 1);
                asm volatile("" ::: "memory");           // the activation loads below stay behind the DMA (they are counted)
            } else {
                if (S + 2 < 9) load_a(S + 2, cc, gpar); else load_a(S + 2 - 9, cc + 1, gpar);
            }
#pragma unroll
            for (int k = 0; k < UPT; k++)
                if (S == USTRIDE * k) {
                    if ((k & 1) == 0) load_unit(k, cc + 1 < c_first + nchunks ? cc + 1 : cc, next_real, sb0, sc0, nv0);
                    else              load_unit(k, cc + 1 < c_first + nchunks ? cc + 1 : cc, next_real, sb1, sc1, nv1);
                    behind += XP ? 2 : 8;
                }
        };
        if (!GA || ku == NOUNIT) fetches();
        Frag f;
        read_frag(f, gpar, PAR, toff[S]);
#define PASTA_MM(PA, PB)                                                                                       \
        if constexpr (mm_on<NP>(PA, PB)) {                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            acc[a][b] = mfma16<IO, NP>(f.a[a][PA], f.b[b][PB], acc[a][b]); }
#define PASTA_SPLIT(J) if (ku != NOUNIT) { if ((ku & 1) == 0) split_pair(sb0, sc0, nv0, J); else split_pair(sb1, sc1, nv1, J); }
        if constexpr (HX && GA) {           // the whole split behind the first group, then the DMA: it has two groups to land
            PASTA_MM(2, 1)
            PASTA_SPLIT(0)
            PASTA_SPLIT(1)
            PASTA_SPLIT(2)
            PASTA_SPLIT(3)
            if (ku != NOUNIT) { store_unit(ku, PAR ^ 1); fetches(); }
            PASTA_MM(1, 0)
        } else if constexpr (HX) {          // three product groups: h'' l', l h, h h -- smallest terms first
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
        if constexpr (HX && GA) {
        } else if constexpr (GA) {
            if (ku != NOUNIT) { store_unit(ku, PAR ^ 1); fetches(); }                 // behind the split (see the note at the head of the kernel)
        } else {
            if (ku != NOUNIT) store_unit(ku, PAR ^ 1);
            store_a(gpar ^ 1, gpar ^ 1);
        }
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        if constexpr (GA) {
            if (behind == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (behind == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (behind == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else __syncthreads();
    };

    // prologue, second half: split and store the first chunk (fetched above), the weights of its first two taps
#pragma unroll
    for (int k = 0; k < UPT; k++) {
#pragma unroll
        for (int j = 0; j < 4; j++) split_pair(fb[k], fc[k], fnv[k], j);
        store_unit(k, 0);
    }
    if constexpr (!GA) {
        store_a(0, 0);
        load_a(1, c_first, 1);                       // stored by step 0
    }
    __syncthreads();                                 // (GA: its fence waits for the DMA of the first tap's weights)
    for (int c = 0; c < nchunks; c += 2) {           // an odd count runs one all-zero chunk (its fetches re-read valid addresses)
        const int cc = c_first + c;
        step(0, 0, cc); step(1, 0, cc); step(2, 0, cc); step(3, 0, cc); step(4, 0, cc); step(5, 0, cc); step(6, 0, cc); step(7, 0, cc); step(8, 0, cc);
        step(0, 1, cc + 1); step(1, 1, cc + 1); step(2, 1, cc + 1); step(3, 1, cc + 1); step(4, 1, cc + 1); step(5, 1, cc + 1); step(6, 1, cc + 1); step(7, 1, cc + 1); step(8, 1, cc + 1);
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
    // What an output ROW brings to the epilogue -- bias, output scale -- depends on (a, r16) only: fetched once, in front of the stores, like
    // the row scales above; the residual of a 32 x 32 sub-tile is fetched as sixteen loads in a row and then stored over.  (One load in
    // front of every store serialises on the memory counter: the input-gradient launches that carry another consumer's gradient as residual
    // ran 4 % SLOWER than launch + torch addition that way, profiles/r4_ab_grad_join.txt.)
    const bool fused = p.ksplit == 1;
    const bool has_res = p.res && fused;
    const float* osb = (p.oscale && fused) ? p.oscale + (int64_t)n_img * p.Cout + (int64_t)g * p.Og : nullptr;
    const float* bsb = (p.act && fused && p.bias) ? p.bias + g * p.Og : nullptr;
    if (!osb && !bsb && !has_res) {
        // nothing to fetch: the plain launches (every input gradient without a residual, every convolution without an epilogue) keep the
        // store loop they always had -- the general path below, taken by them too, cost the dominant kernel 3.0 % (286.3 -> 294.8 us on
        // the micro-benchmark's shapes, same box, profiles/r4_ab_epilogues.txt)
        if (fused && !p.act && o_blk + BM <= p.Og) {
            // ... and the commonest of them -- no activation, whole tile of output rows, no K slices -- a loop without a branch per element
            // (the compiler does not unswitch the wave-uniform tests of the general loops: 454 branches in the epilogue's code)
            auto lean = [&](auto with_amax) {
                constexpr bool AM = decltype(with_amax)::value;
#pragma unroll
                for (int b = 0; b < WNT; b++) {
                    const int t = (wn * WNT + b) * 32 + jl;
                    const int r = t / SEG, c = t - r * SEG;
                    const int plane_off = (p0 + r) * p.OW + q0 + c;
                    const int64_t yoff = ((int64_t)n_img * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
                    const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n_img * OHW : 0) + plane_off] * nstr : 0.f;
#pragma unroll
                    for (int a = 0; a < WMT; a++)
#pragma unroll
                        for (int r16 = 0; r16 < 16; r16++) {
                            const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane);
                            const float v = acc[a][b][r16] + nz;
                            io_st<IO>(p.y, yoff + (int64_t)o * OHW, v);
                            if constexpr (AM) amax_take(y_am, v);
                        }
                }
            };
            if (p.y_amax) lean(std::true_type{}); else lean(std::false_type{});
            amax_commit(y_am, y_slot);
            return;
        }
#pragma unroll
        for (int b = 0; b < WNT; b++) {
            const int t = (wn * WNT + b) * 32 + jl;
            const int r = t / SEG, c = t - r * SEG;
            const int plane_off = (p0 + r) * p.OW + q0 + c;
            const int64_t yoff = ((int64_t)n_img * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
            const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n_img * OHW : 0) + plane_off] * nstr : 0.f;
            float* pb = p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW + yoff : nullptr;
#pragma unroll
            for (int a = 0; a < WMT; a++)
#pragma unroll
                for (int r16 = 0; r16 < 16; r16++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane);
                    if (o < p.Og) {
                        float v = acc[a][b][r16] + nz;
                        if (p.act && fused) v = conv_epilogue(v, 0.f, p.act, p.alpha, p.gain, p.clamp);
                        if (pb) pb[(int64_t)o * OHW] = v;
                        else { io_st<IO>(p.y, yoff + (int64_t)o * OHW, v); if (p.y_amax) amax_take(y_am, v); }
                    }
                }
        }
        if (p.ksplit == 1) amax_commit(y_am, y_slot);
        return;
    }
    // (here fused holds: an output scale, a bias or a residual exist only without K slices)
    const EpiAct ea = conv_epi_act(p.act, p.alpha, p.gain, p.clamp, true);
    conv_epilogue_dispatch<(NP == NP_F16X3 || IO != IO_F32)>(o_blk + BM <= p.Og, ea, [&](auto full_c, auto case_c) {
        const bool FULL = full_c;
#pragma unroll
        for (int b = 0; b < WNT; b++) {
            const int t = (wn * WNT + b) * 32 + jl;
            const int r = t / SEG, c = t - r * SEG;
            const int plane_off = (p0 + r) * p.OW + q0 + c;             // stride-1 lattice: the output plane itself
            const int64_t yoff = ((int64_t)n_img * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
            const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n_img * OHW : 0) + plane_off] * nstr : 0.f;
#pragma unroll
            for (int a = 0; a < WMT; a++) {
                // one operand kind at a time through the same sixteen registers (output scale, residual, bias side by side in arrays of
                // their own spill next to the sixty-four accumulators and the store addresses): sixteen loads in a row, then their use
                float tv[16];
                if (osb) {
#pragma unroll
                    for (int r16 = 0; r16 < 16; r16++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane); tv[r16] = osb[(FULL || o < p.Og) ? o : p.Og - 1]; }
#pragma unroll
                    for (int r16 = 0; r16 < 16; r16++) acc[a][b][r16] = fmaf(acc[a][b][r16], tv[r16], nz);
                } else if (has_noise) {
#pragma unroll
                    for (int r16 = 0; r16 < 16; r16++) acc[a][b][r16] += nz;
                }
                if (has_res) {
#pragma unroll
                    for (int r16 = 0; r16 < 16; r16++) {
                        const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane);
                        tv[r16] = (FULL || o < p.Og) ? io_ld1<IO>((const char*)p.res + (yoff + (int64_t)o * OHW) * ES) : 0.f;
                    }
#pragma unroll
                    for (int r16 = 0; r16 < 16; r16++) acc[a][b][r16] += tv[r16];
                }
#pragma unroll
                for (int r16 = 0; r16 < 16; r16++) { const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane); tv[r16] = bsb ? bsb[(FULL || o < p.Og) ? o : p.Og - 1] : 0.f; }
#pragma unroll
                for (int r16 = 0; r16 < 16; r16++) {
                    const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane);
                    const float v = conv_epilogue_c(acc[a][b][r16], tv[r16], ea, case_c);
                    if (FULL || o < p.Og) { io_st<IO>(p.y, yoff + (int64_t)o * OHW, v); amax_take(y_am, v); }
                }
            }
        }
    });
    if (p.ksplit == 1) amax_commit(y_am, y_slot);
}

// Is the 2-D tile applicable: 3x3 stride-1 lattice (9 taps in 3 rows of 3, any order), planes divisible into R x (BN / R) tiles.
template <int BN, int R>
static bool rows2d_tile_ok(int P, int Q) {
    constexpr int SEG = BN / R;
    return P % R == 0 && Q % SEG == 0;
}

template <int BM, int BN, int R, int NP, int IO, bool ISC = false, int NT = 256, bool XP = false, bool GA = false>
static void launch_fwd_rows2d_np(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    constexpr int SEG = BN / R, SLOTS = (R + 2) * (SEG + 2);
    constexpr int APT = (2 * Arith<NP>::npa * BM + NT - 1) / NT;
    constexpr size_t lds = (size_t)(2 * APT * NT * 8 + 2 * 2 * Arith<NP>::npb * SLOTS * 8) * sizeof(__bf16);
    PASTA_SET_LDS((conv_fwd_rows2d_bf16x6_kernel<BM, BN, R, NP, IO, ISC, NT, XP, GA>), lds);
    hipLaunchKernelGGL((conv_fwd_rows2d_bf16x6_kernel<BM, BN, R, NP, IO, ISC, NT, XP, GA>), grid, dim3(NT), lds, s, q);
}

template <int BM, int BN, int R>
static void launch_fwd_rows2d(const ConvFwdParams& p, hipStream_t s) {
    ConvFwdParams q = p;
    q.o_tiles = (p.Og + BM - 1) / BM;
    constexpr int SEG = BN / R;
    const int64_t tiles = (int64_t)p.N * (p.cls[0].P / R) * (p.cls[0].Q / SEG);
    dim3 grid((unsigned)tiles, q.o_tiles * q.ksplit, p.G);
    if (p.iscale && p.bf16x6 == NP_F16X3) launch_fwd_rows2d_np<BM, BN, R, NP_F16X3, IO_F32, true>(q, grid, s);
    else if (p.iscale)       launch_fwd_rows2d_np<BM, BN, R, 3, IO_F32, true>(q, grid, s);   // fp32 storage, six products (the caller checked)
    else if (p.io == IO_BF16) launch_fwd_rows2d_np<BM, BN, R, 1, IO_BF16>(q, grid, s);     // 16-bit storage: one product
    else if (p.io == IO_F16) launch_fwd_rows2d_np<BM, BN, R, 1, IO_F16>(q, grid, s);
    else if (p.bf16x6 == 1)  launch_fwd_rows2d_np<BM, BN, R, 1, IO_F32>(q, grid, s);
    else if (p.bf16x6 == 2)  launch_fwd_rows2d_np<BM, BN, R, 2, IO_F32>(q, grid, s);
    else if (p.bf16x6 == NP_F16X3) launch_fwd_rows2d_np<BM, BN, R, NP_F16X3, IO_F32>(q, grid, s);
    else                     launch_fwd_rows2d_np<BM, BN, R, 3, IO_F32>(q, grid, s);
}

// Rows per 2-D tile for a P x Q lattice on the 128 x 128 tile: 4 (32-column segments), else 2 (64 columns), else 0 = the row
// kernel.  PASTA_ROWS2D=0 keeps the row kernel, =2 prefers two-row tiles, =1 keeps the 64 x 256 tile on the row kernel (A/B measurements).
static int rows2d_rows(int P, int Q) {
    static const int mode = getenv("PASTA_ROWS2D") ? atoi(getenv("PASTA_ROWS2D")) : 8;
    if (mode == 0) return 0;
    if (mode != 2 && rows2d_tile_ok<128, 4>(P, Q)) return 4;
    return rows2d_tile_ok<128, 2>(P, Q) ? 2 : 0;
}

// Eight waves on a 128 x 256 tile (8 rows x 32 columns; plain six-product fp32 launches): the weights of a step are fetched from L2
// and stored to LDS once for 256 pixels instead of once for 128 -- +3.7 .. 6 % over the four-wave 128 x 128 tile on every live
// shape (profiles/r2_rows2d.txt).  A 64 x 512 tile on eight waves (the 64-channel layers) spills and is 10 % slower: not kept.
// PASTA_ROWS2D=4 keeps the four-wave tiles.
static bool rows2d_wide(int P, int Q) {
    static const int mode = getenv("PASTA_ROWS2D") ? atoi(getenv("PASTA_ROWS2D")) : 8;
    return mode == 8 && rows2d_tile_ok<256, 8>(P, Q);
}

static bool rows2d_rows256(int P, int Q) {
    static const int mode = getenv("PASTA_ROWS2D") ? atoi(getenv("PASTA_ROWS2D")) : 8;
    return mode != 0 && mode != 1 && rows2d_tile_ok<256, 8>(P, Q);          // PASTA_ROWS2D=1: 2-D tiles for the 128 x 128 tile only
}

}  // namespace pasta
