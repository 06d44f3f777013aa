// EXPERIMENT, not part of the build (round 3).  The 2-D tile forward kernel, software-pipelined for ONE wave per SIMD.
// To revive: copy next to conv_fwd_rows2d_bf16x6.h, include it from conv_igemm.hip behind that header and call
// launch_fwd_rows2d_pipe<NP>() from try_fwd_rows2d() in place of the eight-wave launch.  It passed tests/test_conv_rows2d_gpu.py and
// tests/test_conv_precision_gpu.py and ran within 2 % of the eight-wave kernel on every shape (profiles/r3_ab_rows2d_pipe2.txt):
//   spade 256->128 @128: 303.8 against 307.9 TFLOP/s; 256->256 @64: 315.9 / 330.2; 512->512 @32: 339.1 / 344.5.
// What it taught (the reasons it is kept):
//   * hipcc moves MFMAs (pure functions of registers) across s_barrier and sched_barrier: the MFMAs of step g + 1 were hoisted to just
//     behind the fragment reads of step g, where they wait for them; the empty asm statements at the top of step() stop that;
//   * a conditional LDS store cuts the loop body into basic blocks, and the fragment reads of the next step then sink across the block
//     boundary: the spare slots keep the trip one basic block;
//   * the timing-only instances (ABL): MFMAs + barriers 509 TFLOP/s, + fragment reads 421, + staging 339 -- a power-capped chip pays
//     for every LDS byte and fetch next to its MFMAs in clock (profiles/r3_pipe_ablation.txt, r3_power_probe.txt: 1.9 GHz at 1390 W).
#pragma once
#include "conv_fwd_rows2d_bf16x6.h"

namespace pasta {

//------------------------------------------------------------------------------------
// conv_fwd_rows2d_bf16x6_kernel keeps two waves per SIMD (64 x 64 outputs each) and lets them cover each other's waits.  On
// the eight-wave tile both waves of a SIMD belong to the same workgroup and meet the same barrier at the same time; measured
// (profiles/r3_pmc_summary.txt) the matrix pipe is busy 47 % of the cycles and a third of the wave cycles are parked.  Here:
//   * four waves on the same 128 x 256 tile (8 rows x 32 columns), each 64 output channels x 128 pixels: 24 MFMAs per step and
//     wave from 14 fragment reads (0.58 KB of LDS traffic per MFMA instead of 0.83), 128 accumulator registers;
//   * one wave per SIMD, so the register file (512 per lane) holds TWO fragment sets: the fragments of step g + 1 are read
//     from LDS while step g multiplies, and no MFMA ever waits for an LDS read it has just issued;
//   * for that the weights of a step reach LDS one step earlier: fetched (L2 -> registers) at step g - 3, stored at g - 2,
//     read as fragments during g - 1, multiplied at g.  Two LDS buffers still suffice: the readers of a buffer have their
//     fragments in registers before the barrier that precedes its next store (the barrier waits for LDS reads);
//   * the B image of the next chunk is complete one step earlier too (units fetched at steps 0, 2, 4 of a chunk and split +
//     stored at 3, 5, 7), because step 8 already reads the next chunk's first fragments.
// Three-product fp16 arithmetic (PASTA_MATH_F16X3) and the six-product split-bf16 one, fp32 storage, no input scale.
// ABL (timing-only instances, results are garbage): 1 = no activation fetches, 2 = no split / B stores, 4 = no weight fetches, 8 = no MFMAs,
// 16 = no weight stores, 32 = no fragment reads in the loop, 64 = no barriers in the loop
template <int NP, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_fwd_rows2d_pipe_kernel(ConvFwdParams p) {
    constexpr bool SGB = !(ABL & 128);                  // 128: leave the order within a step to the compiler
    constexpr int BM = 128, BN = 256, R = 8, NT = 256, IO = IO_F32;
    constexpr bool HX = Arith<NP>::f16x3;
    constexpr int NPA = Arith<NP>::npa, NPB = Arith<NP>::npb;
    constexpr unsigned ES = 4;
    constexpr int WMT = 2, WNT = 4, KC = 16;
    constexpr int WAVES_N = BN / (32 * WNT);                        // 2 x 2 waves
    constexpr int SEG = BN / R, SW = SEG + 2, SLOTS = (R + 2) * SW;
    constexpr int UNITS = 2 * SLOTS, UPT = (UNITS + NT - 1) / NT;    // 680 (slot, k-half) units: three per thread
    static_assert(UPT == 3, "fetch steps 0, 2, 4; store steps 3, 5, 7");
    constexpr int AUNITS = 2 * NPA * BM, APT = (AUNITS + NT - 1) / NT;
    // Threads without a third unit store to spare slots behind the image instead of branching around the store: a branch would cut
    // the step into basic blocks, and hipcc sinks the fragment reads of the next step across the block boundary to their first use.
    constexpr int SPARE = (UPT * NT - UNITS + 1) / 2, SLOTS_P = SLOTS + SPARE;
    constexpr int ABUF = APT * NT * 8, BSEG = SLOTS_P * 8, BBUF = 2 * NPB * BSEG;    // 16-bit elements
    extern __shared__ __attribute__((aligned(16))) __bf16 rows2dp_smem[];
    __bf16* const As = rows2dp_smem;                    // [2][ABUF]
    __bf16* const Bs = rows2dp_smem + 2 * ABUF;         // [2][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int ks = blockIdx.y / p.o_tiles;
    const int o_blk = (blockIdx.y - ks * p.o_tiles) * BM;
    const int P = p.cls[0].P, Q = p.cls[0].Q;
    const int HW = p.H * p.W;
    const int NC = p.Ig_pad / KC;
    const int c_first = (int)((int64_t)NC * ks / p.ksplit);
    const int nchunks = (int)((int64_t)NC * (ks + 1) / p.ksplit) - c_first;
    const int cblocks = Q / SEG, tpi = (P / R) * cblocks;
    const int n_img = blockIdx.x / tpi;
    const int t_in = blockIdx.x - n_img * tpi;
    const int p0 = (t_in / cblocks) * R, q0 = (t_in % cblocks) * SEG;
    const int ymin = p.rows_y0, d0 = p.rows_d0;

    const char* const xbytes = (const char*)p.x;
    const unsigned xb_off = (unsigned)(((int64_t)n_img * p.Cin + (int64_t)g * p.Ig) * HW) * ES;
    const __bf16* wb = (const __bf16*)p.wp + (int64_t)g * p.KK * NC * 6 * p.Og_pad * 8;
    const int64_t a_chunk = (int64_t)6 * p.Og_pad * 8;

    // ---- staging units of this thread (as in conv_fwd_rows2d_bf16x6_kernel)
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
        u_pix[k] = u_ok[k] ? xb_off + (unsigned)(y * p.W + x) * ES : xb_off;
        u_half[k] = half;
        const int spare = u - UNITS;                    // 0 .. 2 SPARE - 1 for the threads beyond the last unit
        u_lds[k] = real ? (half * SLOTS_P + slot) * 8 : ((spare >= SPARE ? SLOTS_P : 0) + SLOTS + (spare >= SPARE ? spare - SPARE : spare)) * 8;
    }

    float x_scale = 1.f, out_scale = 1.f;
    if constexpr (HX) {
        float sx, isx, sw, isw;
        scale_from_amax(amax_of_parts(p.x_amax), sx, isx);
        scale_from_amax(amax_of_parts(p.w_amax) * p.w_gain, sw, isw);
        x_scale = sx; out_scale = isx * isw;
    }
    float sb0[8], sb1[8];
    int nv0 = 0, nv1 = 0;
    auto load_unit = [&](int k, int cc, bool real_chunk, float (&sb)[8], int& nv) {
        const int c0 = cc * KC + u_half[k] * 8;
        const int last = p.Ig - 1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = c0 + j < last ? c0 + j : last;
            sb[j] = io_ld<IO>(xbytes, u_pix[k] + (unsigned)c * (unsigned)HW * ES);
        }
        nv = (u_ok[k] && real_chunk) ? p.Ig - c0 : 0;
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    uint32_t q1[4], q2[4], q3[4];
    auto split_pair = [&](const float (&sb)[8], int nv, int j) {
        float v0 = sb[2 * j], v1 = sb[2 * j + 1];
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
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        q2[j] = w;
        v0 -= __builtin_bit_cast(float, w << 16);
        v1 -= __builtin_bit_cast(float, w & 0xffff0000u);
        PASTA_KEEP_SCALAR(v0);
        v = f32x2{v0, v1};
        q3[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    };
    auto store_unit = [&](int k, int bbuf) {
        __bf16* bd = Bs + bbuf * BBUF + u_lds[k];
        *(uint4*)(bd) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        if constexpr (NPB >= 2) *(uint4*)(bd + 2 * BSEG) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if constexpr (NPB >= 3) *(uint4*)(bd + 4 * BSEG) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    };

    // ---- weights: A(j), the weights of global step j, live in register set j & 1 and LDS buffer j & 1
    const __bf16* wtap[9];
#pragma unroll
    for (int t = 0; t < 9; t++) wtap[t] = wb + (int64_t)p.tap_slab[t] * NC * a_chunk;
    float4 areg0, areg1, areg2, breg0, breg1, breg2;
    auto load_a = [&](int tap, int cc, int set) {              // tap and set are literals at every call
        const int ccl = cc < NC ? cc : NC - 1;          // past the end of the K range: a valid address, unused data
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
    const AmaxSlot y_slot = amax_begin(p.y_amax);
    int fslot[WNT];
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int t = (wn * WNT + b) * 32 + jl;
        const int r = t / SEG, c = t - r * SEG;
        fslot[b] = r * SW + c;
    }
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; t++) toff[t] = (p.tap_dy[t] - ymin) * SW + (p.tap_dx[t] - d0);

    struct Frag { bf16x8 a[WMT][3], b[WNT][3]; };
    auto read_frag = [&](Frag& f, int abuf, int bbuf, int off) {
        const __bf16* A_ = As + abuf * ABUF;
        const __bf16* B_ = Bs + bbuf * BBUF;
#define PASTA_LDA(PC) if constexpr ((PC) < NPA) { _Pragma("unroll") for (int a = 0; a < WMT; a++) f.a[a][PC] = *(const bf16x8*)&A_[(((PC) * 2 + hl) * BM + (wm * WMT + a) * 32 + jl) * 8]; }
#define PASTA_LDB(PC) if constexpr ((PC) < NPB) { _Pragma("unroll") for (int b = 0; b < WNT; b++) f.b[b][PC] = *(const bf16x8*)&B_[(((PC) * 2 + hl) * SLOTS_P + fslot[b] + off) * 8]; }
        if constexpr (HX) { PASTA_LDA(2) PASTA_LDB(1) PASTA_LDA(1) PASTA_LDB(0) PASTA_LDA(0) }
        else { PASTA_LDA(2) PASTA_LDB(0) PASTA_LDA(0) PASTA_LDB(2) PASTA_LDA(1) PASTA_LDB(1) }
#undef PASTA_LDA
#undef PASTA_LDB
    };

    // One step = one tap of one chunk.  S (0..8), PAR (parity of the chunk within the trip) and the fragment sets are
    // literals; global step index g = 9 * chunk + S, parity gpar.
    //   this step multiplies `cur` (read during the previous step) and reads `nxt` = the fragments of step g + 1:
    //     A buffer (g + 1) & 1, and the B image of the same chunk at the next tap -- at S == 8 the NEXT chunk's image at tap 0;
    //   stores A(g + 2) from register set g & 1 into buffer g & 1 (its readers, step g - 1, are done: barrier) and fetches
    //     A(g + 3) into set (g + 1) & 1;
    //   staging unit k of chunk cc + 1: fetched at step 2k into set k & 1, split + stored at step 2k + 3.
    auto step = [&](const int S, const int PAR, int cc, Frag& cur, Frag& nxt) {
        __builtin_amdgcn_sched_barrier(0);              // nothing of the previous step sinks into this one (hipcc moves register-only MFMAs across s_barrier)
        // ... and nothing of this step rises into the previous one: an MFMA is a pure function of registers, and LLVM hoists the MFMAs of step
        // g + 1 to just behind the fragment reads of step g (where they wait for those reads).  The empty statements re-define this step's
        // fragments HERE.
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int pc = 0; pc < NPA; pc++) asm volatile("" : "+v"(cur.a[a][pc]));
#pragma unroll
        for (int b = 0; b < WNT; b++)
#pragma unroll
            for (int pc = 0; pc < NPB; pc++) asm volatile("" : "+v"(cur.b[b][pc]));
        const int gpar = (PAR * 9 + S) & 1;
        const bool next_real = cc + 1 < c_first + nchunks;
        {   // A(g + 3): tap S + 3 of this chunk, or of the next one
            const int t3 = S + 3 < 9 ? S + 3 : S + 3 - 9;
            if (!(ABL & 4)) load_a(t3, S + 3 < 9 ? cc : cc + 1, gpar ^ 1);
        }
#pragma unroll
        for (int k = 0; k < UPT; k++)
            if (S == 2 * k && !(ABL & 1)) {
                if ((k & 1) == 0) load_unit(k, next_real ? cc + 1 : cc, next_real, sb0, nv0);
                else              load_unit(k, next_real ? cc + 1 : cc, next_real, sb1, nv1);
            }
        constexpr int NOUNIT = -1;
        int ku = NOUNIT;
#pragma unroll
        for (int k = 0; k < UPT; k++)
            if (S == 2 * k + 3 && !(ABL & 2)) ku = k;
        // fragments of the next step: they land behind this step's MFMAs
        if constexpr (ABL & 32) { asm volatile("" : "+v"(nxt.a[0][0]), "+v"(nxt.b[0][0])); }
        else if (S < 8) read_frag(nxt, gpar ^ 1, PAR, toff[S + 1]);
        else            read_frag(nxt, gpar ^ 1, PAR ^ 1, toff[0]);
#define PASTA_MM(PA, PB)                                                                                       \
        if constexpr (mm_on<NP>(PA, PB)) {                                                                       \
        _Pragma("unroll") for (int a = 0; a < WMT; a++) _Pragma("unroll") for (int b = 0; b < WNT; b++)          \
            { if constexpr (ABL & 8) asm volatile("" :: "v"(cur.a[a][PA]), "v"(cur.b[b][PB])); else acc[a][b] = mfma16<IO, NP>(cur.a[a][PA], cur.b[b][PB], acc[a][b]); } }
#define PASTA_SPLIT(J) if (ku != NOUNIT) { if ((ku & 1) == 0) split_pair(sb0, nv0, J); else split_pair(sb1, nv1, J); }
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
        if (ku != NOUNIT) store_unit(ku, PAR ^ 1);
        if (!(ABL & 16)) store_a(gpar, gpar);
        PASTA_MM(0, 0)
#undef PASTA_MM
#undef PASTA_SPLIT
        // The order within the step, pinned: the fetches first (the oldest loads when a later split waits), then one fragment read
        // of the next step behind each of the first fourteen MFMAs (a burst of fourteen ds_read_b128 holds the wave's issue until
        // the LDS queue has drained), the split's VALU work behind the next five, the LDS stores behind the last five.
        if constexpr (SGB) {
            __builtin_amdgcn_sched_group_barrier(0x020, 11, 0);
#define PASTA_SGB2(M1, N1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(M1, N1, 0);
            PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1)
            PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1) PASTA_SGB2(0x100, 1)
            PASTA_SGB2(0x002, 12) PASTA_SGB2(0x002, 12) PASTA_SGB2(0x002, 12) PASTA_SGB2(0x002, 12) PASTA_SGB2(0x002, 12)
            PASTA_SGB2(0x200, 1) PASTA_SGB2(0x200, 1) PASTA_SGB2(0x200, 1) PASTA_SGB2(0x200, 1) PASTA_SGB2(0x200, 1)
#undef PASTA_SGB2
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 64)) __syncthreads();
    };

    // prologue: the B image of the first chunk, A(0) and A(1) in LDS, A(2) in flight (register set 0), the fragments of step 0
    load_a(0, c_first, 0);
    load_a(1, c_first, 1);
#pragma unroll
    for (int k = 0; k < UPT; k++) {
        load_unit(k, c_first, nchunks > 0, sb0, nv0);
#pragma unroll
        for (int j = 0; j < 4; j++) split_pair(sb0, nv0, j);
        store_unit(k, 0);
    }
    store_a(0, 0);
    store_a(1, 1);
    load_a(2, c_first, 0);                           // stored by step 0
    __syncthreads();
    Frag f0, f1;
    read_frag(f0, 0, 0, toff[0]);
    __syncthreads();                                 // every wave holds its fragments of step 0 before step 0 overwrites A buffer 0
    for (int c = 0; c < nchunks; c += 2) {           // an odd count runs one all-zero chunk (its fetches re-read valid addresses)
        const int cc = c_first + c;
        step(0, 0, cc, f0, f1); step(1, 0, cc, f1, f0); step(2, 0, cc, f0, f1); step(3, 0, cc, f1, f0); step(4, 0, cc, f0, f1);
        step(5, 0, cc, f1, f0); step(6, 0, cc, f0, f1); step(7, 0, cc, f1, f0); step(8, 0, cc, f0, f1);
        step(0, 1, cc + 1, f1, f0); step(1, 1, cc + 1, f0, f1); step(2, 1, cc + 1, f1, f0); step(3, 1, cc + 1, f0, f1); step(4, 1, cc + 1, f1, f0);
        step(5, 1, cc + 1, f0, f1); step(6, 1, cc + 1, f1, f0); step(7, 1, cc + 1, f0, f1); step(8, 1, cc + 1, f1, f0);
    }

    const int OHW = p.OH * p.OW;
    const bool has_noise = p.noise && p.ksplit == 1;
    const float nstr = has_noise ? p.noise_strength[0] : 0.f;
#pragma unroll
    for (int b = 0; b < WNT; b++) {
        const int t = (wn * WNT + b) * 32 + jl;
        const int r = t / SEG, c = t - r * SEG;
        const int plane_off = (p0 + r) * p.OW + q0 + c;
        const int64_t yoff = ((int64_t)n_img * p.Cout + (int64_t)g * p.Og) * OHW + plane_off;
        const float nz = has_noise ? p.noise[(p.noise_ps ? (int64_t)n_img * OHW : 0) + plane_off] * nstr : 0.f;
        float* pb = p.ksplit > 1 ? p.partial + (int64_t)ks * p.N * p.Cout * OHW + yoff : nullptr;
        const bool has_res = p.res && p.ksplit == 1;
        const float* osb = (p.oscale && p.ksplit == 1) ? p.oscale + (int64_t)n_img * p.Cout + (int64_t)g * p.Og : nullptr;
#pragma unroll
        for (int a = 0; a < WMT; a++)
#pragma unroll
            for (int r16 = 0; r16 < 16; r16++) {
                const int o = o_blk + (wm * WMT + a) * 32 + acc_row(r16, lane);
                if (o < p.Og) {
                    float v = acc[a][b][r16];
                    if constexpr (HX) v *= out_scale;
                    v = conv_scale_noise(v, osb, o, nz);
                    if (has_res) v += io_ld1<IO>((const char*)p.res + (yoff + (int64_t)o * OHW) * ES);
                    if (p.act && p.ksplit == 1) v = conv_epilogue(v, p.bias ? p.bias[g * p.Og + o] : 0.f, p.act, p.alpha, p.gain, p.clamp);
                    if (pb) pb[(int64_t)o * OHW] = v;
                    else { io_st<IO>(p.y, yoff + (int64_t)o * OHW, v); if (p.y_amax) amax_take(y_am, v); }
                }
            }
    }
    if (p.ksplit == 1) amax_commit(y_am, y_slot);
}

template <int NP, int ABL = 0>
static void launch_fwd_rows2d_pipe_abl(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    constexpr int SLOTS = (8 + 2) * (32 + 2) + (3 * 256 - 2 * (8 + 2) * (32 + 2) + 1) / 2;          // the image and the spare slots
    constexpr int APT = (2 * Arith<NP>::npa * 128 + 255) / 256;
    constexpr size_t lds = (size_t)(2 * APT * 256 * 8 + 2 * 2 * Arith<NP>::npb * SLOTS * 8) * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_rows2d_pipe_kernel<NP, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_fwd_rows2d_pipe_kernel<NP, ABL>), grid, dim3(256), lds, s, q);
}

template <int NP>
static void launch_fwd_rows2d_pipe(const ConvFwdParams& q, dim3 grid, hipStream_t s) {
    if constexpr (NP == NP_F16X3) {
        static const int abl = getenv("PASTA_PIPE_ABLATE") ? atoi(getenv("PASTA_PIPE_ABLATE")) : 0;
        switch (abl) {
#define PASTA_ABL(A) case A: launch_fwd_rows2d_pipe_abl<NP, A>(q, grid, s); return;
            PASTA_ABL(128) PASTA_ABL(1) PASTA_ABL(3) PASTA_ABL(4) PASTA_ABL(20) PASTA_ABL(23) PASTA_ABL(55) PASTA_ABL(8) PASTA_ABL(64) PASTA_ABL(87) PASTA_ABL(119)
#undef PASTA_ABL
            default: break;
        }
    }
    launch_fwd_rows2d_pipe_abl<NP, 0>(q, grid, s);
}

}  // namespace pasta
