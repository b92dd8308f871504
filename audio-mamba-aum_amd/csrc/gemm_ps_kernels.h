// gemm_ps_kernels.h -- aum_gemm_tn with PACED STORES (round 6): the 8-wave 256 x 256 x 64 kernel of gemm_kernels.h (same LDS image, swizzles,
// MFMA roles, store map) as ONE continuous stream of K-steps across a workgroup's tiles, with a hand-placed schedule.
//
// What round 6's ablations of the round-3 persistent kernel said (profiles/r06_gemm_ablations.txt; in_proj forward, 64 x 513 tokens):
//   whole kernel 161 us | no tile-end stores 140 | + no DMA pieces (reads + MFMAs + barriers only) 113 | DMA pieces + barriers only 90
// -- a K-step is 1.56 us of compute (the compiler reads two fragments, waits for them, issues eight MFMAs, eight times over: the matrix
// pipe waits out an LDS round trip per row pair in both waves of a SIMD at once), 0.37 us on top when the step's eight DMA pieces per wave
// are issued in one burst at its head, and a tile's 16 stores per wave cost 3.5 us with every matrix pipe idle.  The L2 -> LDS path
// delivers a K-step's 64 KB per CU in ~1.15 us whether the pieces come in a burst or as a deep queue (and whether they go through
// registers or not): that, not the MFMA rate (0.85-1.0 us per step), is the floor -- so the tile stays 256 x 256 (a 256 x 192 tile
// moves 17 % more bytes per flop) and everything else has to hide under it.
//
//   * accumulators: 128 AGPRs, updated in place by inline-assembly MFMAs with tied operands; volatile assembly keeps its order and
//     memory operations do not move across it, so every fragment read, DMA piece and store below is PLACED between MFMAs by source order;
//   * fragments: the four weight fragments of a half K-step are double-buffered (bfA / bfB), the activation fragments go through a ring of
//     four (row g of the step's 16 fragment rows is read while row g - 3 multiplies): 48 registers instead of two full sets (96);
//   * one workgroup barrier per K-step, between fragment rows 12 and 13: every read of the step's buffer has been issued (and waited
//     for) by then, the next step's pieces have landed; rows 13 .. 15 cover the first reads of the next step and the first four pieces
//     of the step after it (the other four follow under rows 0 and 1);
//   * a finished tile is rounded to 16 bits into 64 registers while its last half step multiplies (row i - 1 behind the MFMAs of row
//     i), and its stores leave paced: fragment rows 0, 1 inside that last step, rows 2 .. 7 two stores per K-step under the NEXT tile's first
//     six steps (48 parked registers; with all 64 parked the kernel spilled).  Buffer stores: rows beyond M are dropped by the range
//     check, so every tile issues the same 16 stores per wave and the counted waits -- vmcnt(2), vmcnt(4) in a last step -- hold for all;
//   * a tile's first half step multiplies with a zero addend (no accumulator fill);
//   * a ragged last row block (64 x 513 tokens = 128 row blocks + 64 rows) is an ordinary item whose rows beyond M read as zero (no memory
//     traffic: such a tile runs at the MFMA stream's pace, 0.65 of a full one) and are dropped by the stores' range check.
// Built on top of this and measured slower or level, all bit-equal (profiles/r06_gemm_ablations.txt (5), removed again): four 32-deep stages with
// the pieces requested four steps ahead (half cache lines per request: every line crosses the L2 -> L1 path twice -- also why round 5's ring
// kernel lost); L2 prefetch touches three steps ahead (64 lines per 4-byte load are 64 tag look-ups in the CU's L1); skipping the MFMAs of
// fragment rows beyond M (a scalar test per MFMA costs every tile 20 %; a second copy of the steps for ragged tiles makes the register
// allocator move the accumulators between the copies: 112 spills); two copies of the loop for the two waves of a SIMD with their pieces
// and stores at different MFMAs (level).  Second session of round 6 (profiles/r06_gemm_ring3.txt, commit b425e1a): the activation or the weight tiles in
// a ring of three LDS slots (the weights' piece stream 20 % faster, the kernel level), non-temporal loads (-15..30 %), an XCD's items from one
// half of the column tiles (level) -- and the ablation that explains them: with every operand line an L2 hit the kernel is no faster; fragment
// reads + arriving pieces are 256 KB of LDS traffic per K-step and CU = the 2 048 clocks the step's MFMAs take.
#pragma once
#include "gemm_kernels.h"

namespace aumg {

#ifndef AUM_PS_DEAD_WAVES
#define AUM_PS_DEAD_WAVES 1     // 0: A/B build -- waves without live rows multiply zeros (as until the split last round)
#endif
#ifndef AUM_PS_ABL
#define AUM_PS_ABL 0        // timing experiments only (wrong results): 1 no DMA pieces in the steps, 2 no stores, 4 no MFMAs, 8 every store dropped by the range check,
                            // 16 every tile reads the activation rows of row block 0 (L2 hits), 32 the weight rows of column tile 0 (first kernel only)
#endif

// acc += a . b with the accumulator named as an AGPR tile updated in place (through the builtin hipcc picks an early-clobber destination and
// rotates the accumulators of the loop through copies).  Inline assembly is outside the compiler's hazard recogniser: the only hazard
// here is a vector-ALU read of an accumulator behind its last MFMA, covered where the tile is rounded.
template <bool BF16> __device__ __forceinline__ void ps_mfma(f4v& c, const s8v& a, const s8v& b) {
    if constexpr (AUM_PS_ABL & 4) return;
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool BF16> __device__ __forceinline__ void ps_mfma0(f4v& c, const s8v& a, const s8v& b) {
    if constexpr (AUM_PS_ABL & 4) { c = f4v{0.f, 0.f, 0.f, 0.f}; return; }
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
template <int V> struct PsC { static constexpr int value = V; };

// work item id -> tile (GemmLaunch in gemm_kernels.h): a whole item is a 256-row block (the last one may be ragged: its rows beyond M read as
// zero and are not stored) x one 256-column tile; a half item is 128 (the last one: 128 + fold) rows of a split tile, run as a ragged tile.
// In a complete round of `grid` items XCD x (= workgroup % 8) takes grid / 8 consecutive items (both halves of a split tile on one XCD)
// (the tile is handed back as plain integers: a struct captured by reference by the lambdas below is an alloca, the compiler moves it to
// LDS, and every copy of it then waits for ALL outstanding DMA pieces -- vmcnt(0) at the head of the tile loop)
__device__ __forceinline__ void ps_item(const GemmLaunch& L, int id, int ntn, int grid, int& m0, int& n0, int& rows) {
    int item = id;
    const int r0 = id / grid * grid;
    if ((grid & 7) == 0 && r0 + grid <= L.nitems) {
        const int q = id - r0;
        item = r0 + (q & 7) * (grid >> 3) + (q >> 3);
    }
    const int h = item - L.nwhole;                      // >= 0: half h & 1 of tile nwhole + (h >> 1)
    const int tile = h < 0 ? item : L.nwhole + (h >> 1);
    const int tm = tile / ntn, tn = tile - tm * ntn;
    m0 = tm * BM;
    n0 = tn * BN;
    int span = BM;
    if (h >= 0) {
        m0 += (h & 1) * (BM / 2);
        span = BM / 2 + (((h & 1) && (tm + 1) * BM + L.fold == L.g.m) ? L.fold : 0);       // the fold rows: behind the last block's second half
    }
    rows = L.g.m - m0 < span ? L.g.m - m0 : span;
}

constexpr int PS_NST = 6;           // K-steps of a tile that carry the previous tile's stores (fragment rows 2 .. 7, two stores per wave each); rows 0, 1 leave
                                    // inside the tile's own last step: 48 parked registers instead of 64 (64 spilled); k >= 7 * 64

template <bool BF16>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn_ps(GemmLaunch L) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    const int ntn = g.n / BN, grid = (int)gridDim.x, nk = g.k / BK;

    // staging (gemm_kernels.h): piece c = j * 8 + w is rows 8 c .. 8 c + 7 of the operand tile, lane l fills slot l & 7 of row 8 c + (l >> 3)
    const int srow = w * 8 + (lane >> 3);
    const int f_a = ((w & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = ((w & 3) << 1) | ((lane >> 4) & 1);
    const int voff_a = srow * g.lda * 2 + (((lane & 7) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 7) ^ f_b) << 4);
    const int rowstep_a = 64 * g.lda * 2, rowstep_b = 64 * g.ldb * 2;
    // fragment reads: lane = (operand row rho, k-group kg)
    const int rho = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + rho) * 128 + ((kg ^ ((lane >> 1) & 7)) << 4);                                  // + i * 2048, ^ 64 for the second half of K
    const int b_row = wc * 64 + (rho >> 2) * 8 + (rho & 3);
    const int b_rd = TILE_BYTES + b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);     // + b_joff(j), ^ 64
    // stores: lane holds, for fragment row i, columns wc * 64 + 32 (j >> 1) + 8 kg + 4 (j & 1) + r of row wr * 128 + 16 i + rho
    const int ldc2 = g.ldc * 2;
    const int c_voff = (wr * 128 + rho) * ldc2 + (wc * 64 + kg * 8) * 2;

    auto rsrc_a = [&](int m0, int rows) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.a) + (int64_t)((AUM_PS_ABL & 16) ? 0 : m0) * g.lda * 2), 0, rows * g.lda * 2, 0x00020000);
    };
    auto rsrc_b = [&](int n0) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.b) + (int64_t)((AUM_PS_ABL & 32) ? 0 : n0) * g.ldb * 2), 0, BN * g.ldb * 2, 0x00020000);
    };
    auto rsrc_c = [&](int m0, int n0, int rows) {          // the tile's rows of C from column n0 on: rows beyond `rows` are out of range (stores dropped)
        return __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(g.c) + ((int64_t)m0 * g.ldc + n0) * 2, 0, (rows - 1) * ldc2 + BN * 2, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), 0, 0, 0x00020000);

    int id = (int)blockIdx.x;
    if (id >= L.nitems) return;
    int m0, n0, rows;
    ps_item(L, id, ntn, grid, m0, n0, rows);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0, rows), rb = rsrc_b(n0);

    // piece n of a K-step (n < 4: activation rows, else weight rows) into stage `dst`
    auto piece = [&](__amdgpu_buffer_rsrc_t ra_s, __amdgpu_buffer_rsrc_t rb_s, int kbyte, char* dst, int n) {
        if (AUM_PS_ABL & 1) return;
        if (AUM_PS_ABL & 64) ra_s = rb_s = r_null;          // 64: every piece out of range (issued, zero-filled, no memory traffic)
        if (n < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_s, (lds_ptr_t)(dst + (n * 8 + w) * 1024), 16, voff_a, kbyte + n * rowstep_a, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_s, (lds_ptr_t)(dst + TILE_BYTES + ((n - 4) * 8 + w) * 1024), 16, voff_b,
                                                      kbyte + (n - 4) * rowstep_b, 0, 0);
    };

    s8v bfA[4], bfB[4], af[4];
    u4v pend[6][2];                     // fragment rows 2 .. 7 of the finished tile, rounded
#pragma unroll
    for (int i = 0; i < 6; ++i) pend[i][0] = pend[i][1] = u4v{0u, 0u, 0u, 0u};
    f4v acc[8][4];

    // ---- prologue: step 0 of the first tile lands, the first four pieces of step 1 leave, the first fragments of step 0 are read
    if (!(AUM_PS_ABL & 1)) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds + (n * 8 + w) * 1024), 16, voff_a, n * rowstep_a, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds + TILE_BYTES + ((n - 4) * 8 + w) * 1024), 16, voff_b, (n - 4) * rowstep_b, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const bool has_next = id + grid < L.nitems;
        int m1 = m0, n1 = n0, rows1 = rows;
        if (has_next) ps_item(L, id + grid, ntn, grid, m1, n1, rows1);
        const bool same = 1 < nk;
        const __amdgpu_buffer_rsrc_t ra_s = same ? ra : (has_next ? rsrc_a(m1, rows1) : r_null), rb_s = same ? rb : (has_next ? rsrc_b(n1) : r_null);
#pragma unroll
        for (int n = 0; n < 4; ++n) piece(ra_s, rb_s, same ? BK * 2 : 0, lds + STAGE_BYTES, n);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bfA[j] = lds_frag(lds, b_rd + b_joff(j));
#pragma unroll
    for (int r = 0; r < 3; ++r) af[r] = lds_frag(lds, a_rd + r * 2048);
    int par = 0;                                    // the stage that holds the step about to run
    __amdgpu_buffer_rsrc_t rc_prev = r_null;        // C rows of the tile whose rounded values wait in `pend`

    while (true) {
        const int nid = id + grid;
        const bool has_next = nid < L.nitems;
        int m1 = m0, n1 = n0, rows1 = rows;
        if (has_next) ps_item(L, nid, ntn, grid, m1, n1, rows1);
        const __amdgpu_buffer_rsrc_t ra_n = has_next ? rsrc_a(m1, rows1) : r_null, rb_n = has_next ? rsrc_b(n1) : r_null;
        const __amdgpu_buffer_rsrc_t rc = (AUM_PS_ABL & 8) ? r_null : rsrc_c(m0, n0, rows);

        // One K-step.  FIRST: a tile's first step (zero addend in its first half); ST >= 0: the step carries stores 2 ST, 2 ST + 1 of the
        // previous tile; LAST: the tile's last step (rows rounded into `pend` behind the second half's MFMAs).
        auto kstep = [&](auto first_c, auto st_c, auto last_c, int t) {
            constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
            constexpr int ST = decltype(st_c)::value;
            char* cur = lds + par * STAGE_BYTES;
            char* oth = lds + (par ^ 1) * STAGE_BYTES;
            // sources of the pieces this step issues: the rest of step t + 1 (rows 0, 1), the head of step t + 2 (rows 13 .. 15)
            const bool same1 = t + 1 < nk, same2 = t + 2 < nk;
            const __amdgpu_buffer_rsrc_t ra_1 = same1 ? ra : ra_n, rb_1 = same1 ? rb : rb_n;
            const __amdgpu_buffer_rsrc_t ra_2 = same2 ? ra : ra_n, rb_2 = same2 ? rb : rb_n;
            const int kb1 = (same1 ? t + 1 : 0) * (BK * 2), kb2 = (same2 ? t + 2 : t + 2 - nk) * (BK * 2);
            // ---- first half: fragment rows g = 0 .. 7 on bfA
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FIRST) ps_mfma0<BF16>(acc[i][j], bfA[j], af[i & 3]);
                    else ps_mfma<BF16>(acc[i][j], bfA[j], af[i & 3]);
                    if (j == 1) {                                   // the activation fragment of row g + 3 into the slot row g - 1 has left
                        if (i + 3 < 8) af[(i + 3) & 3] = lds_frag(cur, a_rd + (i + 3) * 2048);
                        else af[(i + 3) & 3] = lds_frag(cur, (a_rd ^ 64) + (i + 3 - 8) * 2048);
                    }
                    if (j == 3 && i >= 4) bfB[i - 4] = lds_frag(cur, (b_rd ^ 64) + b_joff(i - 4));      // the second half's weight fragments
                    if (i < 2 && (j == 0 || j == 2)) piece(ra_1, rb_1, kb1, oth, 4 + i * 2 + (j >> 1));
                    if constexpr (ST >= 0 && !(AUM_PS_ABL & 2)) {
                        if ((i == 3 || i == 4) && j == 0)
                            __builtin_amdgcn_raw_buffer_store_b128(pend[ST][i - 3], rc_prev, c_voff + (ST + 2) * 16 * ldc2 + (i - 3) * 64, 0, 0);
                    }
                }
            }
            // ---- second half: fragment rows g = 8 .. 15 on bfB
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == 5) {
                    // every read of this step's stage has been issued (the last one under row 12); this wave's have returned, its pieces of
                    // step t + 1 have landed (behind them: this step's two stores at most) -- and, past the barrier, everybody's
                    // (behind the pieces: the step's stores -- two, or the four of rows 0, 1 in a tile's last step)
                    constexpr int NSTO = (AUM_PS_ABL & 2) ? 0 : ST >= 0 ? 2 : LAST ? 4 : 0;
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ps_mfma<BF16>(acc[i][j], bfB[j], af[i & 3]);
                    if (j == 1 && i + 3 < 8) af[(i + 3) & 3] = lds_frag(cur, (a_rd ^ 64) + (i + 3) * 2048);
                    if (i >= 5) {
                        // under rows 13 .. 15: the next step's first fragments (its stage has landed) and the first four pieces of the step
                        // after it (into the stage this step has finished with).  Ring slots: rows 13, 14, 15 multiply on slots 1, 2, 3 --
                        // slot 0 is free, slot 1 once row 13's MFMAs have been issued, slot 2 after row 14's
                        const int q = (i - 5) * 4 + j;              // 0 .. 11
                        if (q == 0 || q == 3 || q == 6 || q == 9) piece(ra_2, rb_2, kb2, cur, q / 3);
                        if (q == 1) bfA[0] = lds_frag(oth, b_rd + b_joff(0));
                        if (q == 2) af[0] = lds_frag(oth, a_rd);
                        if (q == 4) bfA[1] = lds_frag(oth, b_rd + b_joff(1));
                        if (q == 5) af[1] = lds_frag(oth, a_rd + 2048);
                        if (q == 7) bfA[2] = lds_frag(oth, b_rd + b_joff(2));
                        if (q == 8) bfA[3] = lds_frag(oth, b_rd + b_joff(3));
                        if (q == 10) af[2] = lds_frag(oth, a_rd + 2 * 2048);
                    }
                }
                if constexpr (LAST) {
                    if (i > 0) {
                        // fragment row i - 1 is final (its last MFMAs were issued four MFMAs ago): round it.  Its vector-ALU reads must stay
                        // BEHIND row i's MFMAs -- the compiler knows nothing about the latency of the assembly that produced the values
                        // and would hoist them right behind it: an empty volatile statement that "rewrites" the row pins them here
#pragma unroll
                        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i - 1][j]));
                        const u4v lo = u4v{pack2<BF16>(acc[i - 1][0][0], acc[i - 1][0][1]), pack2<BF16>(acc[i - 1][0][2], acc[i - 1][0][3]),
                                           pack2<BF16>(acc[i - 1][1][0], acc[i - 1][1][1]), pack2<BF16>(acc[i - 1][1][2], acc[i - 1][1][3])};
                        const u4v hi = u4v{pack2<BF16>(acc[i - 1][2][0], acc[i - 1][2][1]), pack2<BF16>(acc[i - 1][2][2], acc[i - 1][2][3]),
                                           pack2<BF16>(acc[i - 1][3][0], acc[i - 1][3][1]), pack2<BF16>(acc[i - 1][3][2], acc[i - 1][3][3])};
                        if (i - 1 < 2) {            // rows 0, 1 leave at once (under rows 9 .. 11: in front of the barrier's wait, which leaves them in flight)
                            if (!(AUM_PS_ABL & 2)) {
                                __builtin_amdgcn_raw_buffer_store_b128(lo, rc, c_voff + (i - 1) * 16 * ldc2, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(hi, rc, c_voff + (i - 1) * 16 * ldc2 + 64, 0, 0);
                            }
                        } else {
                            pend[i - 3][0] = lo;
                            pend[i - 3][1] = hi;
                        }
                    }
                }
            }
            if constexpr (LAST) {
                // last MFMA -> vector-ALU reads of its accumulators: 18 wait states, and the reads pinned behind them
                asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]));
                pend[5][0] = u4v{pack2<BF16>(acc[7][0][0], acc[7][0][1]), pack2<BF16>(acc[7][0][2], acc[7][0][3]),
                                 pack2<BF16>(acc[7][1][0], acc[7][1][1]), pack2<BF16>(acc[7][1][2], acc[7][1][3])};
                pend[5][1] = u4v{pack2<BF16>(acc[7][2][0], acc[7][2][1]), pack2<BF16>(acc[7][2][2], acc[7][2][3]),
                                 pack2<BF16>(acc[7][3][0], acc[7][3][1]), pack2<BF16>(acc[7][3][2], acc[7][3][3])};
            }
            par ^= 1;
        };
        // A wave whose 128 rows all lie beyond the item's rows (the lower wave row of a half item or of a ragged row block of at most 128
        // rows -- always the workgroup's LAST item: both kinds sit in the last round) has nothing to multiply: it keeps its share of the
        // pieces, the previous tile's paced stores and the barriers, and leaves the SIMD's matrix pipe to the wave that has rows.
        if (AUM_PS_DEAD_WAVES && wr == 1 && rows <= BM / 2 && !has_next) {
            auto kdead = [&](auto st_c, int t) {
                constexpr int ST = decltype(st_c)::value;
                char* cur = lds + par * STAGE_BYTES;
                char* oth = lds + (par ^ 1) * STAGE_BYTES;
                const bool same1 = t + 1 < nk, same2 = t + 2 < nk;
#pragma unroll
                for (int n = 4; n < 8; ++n) piece(same1 ? ra : r_null, same1 ? rb : r_null, (same1 ? t + 1 : 0) * (BK * 2), oth, n);
                if constexpr (ST >= 0 && !(AUM_PS_ABL & 2)) {
                    __builtin_amdgcn_raw_buffer_store_b128(pend[ST][0], rc_prev, c_voff + (ST + 2) * 16 * ldc2, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pend[ST][1], rc_prev, c_voff + (ST + 2) * 16 * ldc2 + 64, 0, 0);
                }
                constexpr int NSTO = (AUM_PS_ABL & 2) ? 0 : ST >= 0 ? 2 : 0;
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO) : "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int n = 0; n < 4; ++n) piece(same2 ? ra : r_null, same2 ? rb : r_null, (same2 ? t + 2 : 0) * (BK * 2), cur, n);
                par ^= 1;
            };
            kdead(PsC<0>{}, 0);
            kdead(PsC<1>{}, 1);
            kdead(PsC<2>{}, 2);
            kdead(PsC<3>{}, 3);
            kdead(PsC<4>{}, 4);
            kdead(PsC<5>{}, 5);
            for (int t = PS_NST; t < nk; ++t) kdead(PsC<-1>{}, t);
            return;                                     // nothing of this item to store: its rows of C are all out of range
        }
        kstep(PsC<1>{}, PsC<0>{}, PsC<0>{}, 0);
        kstep(PsC<0>{}, PsC<1>{}, PsC<0>{}, 1);
        kstep(PsC<0>{}, PsC<2>{}, PsC<0>{}, 2);
        kstep(PsC<0>{}, PsC<3>{}, PsC<0>{}, 3);
        kstep(PsC<0>{}, PsC<4>{}, PsC<0>{}, 4);
        kstep(PsC<0>{}, PsC<5>{}, PsC<0>{}, 5);
        for (int t = PS_NST; t + 1 < nk; ++t) kstep(PsC<0>{}, PsC<-1>{}, PsC<0>{}, t);
        kstep(PsC<0>{}, PsC<-1>{}, PsC<1>{}, nk - 1);
        rc_prev = rc;
        if (!has_next) break;
        id = nid;
        m0 = m1, n0 = n1, rows = rows1;
        ra = ra_n;
        rb = rb_n;
    }
    // the last tile's stores (rows 2 .. 7; rows 0, 1 left in its last step)
    if (!(AUM_PS_ABL & 2)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_raw_buffer_store_b128(pend[i][0], rc_prev, c_voff + (i + 2) * 16 * ldc2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(pend[i][1], rc_prev, c_voff + (i + 2) * 16 * ldc2 + 64, 0, 0);
        }
    }
}

}  // namespace aumg
