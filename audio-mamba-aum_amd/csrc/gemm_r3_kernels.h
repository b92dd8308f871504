// gemm_r3_kernels.h -- aum_gemm_tn, the paced-store kernel of gemm_ps_kernels.h with the ACTIVATION tiles in a ring of three LDS slots
// (5 x 32 KB = the CU's whole 160 KB: three activation slots, two weight slots).  Why: profiles/r06_gemm_ablations.txt (4) -- with both
// operands forced L2-resident the piece stream of a K-step takes 0.78 us instead of 1.19: what the two-stage kernel waits for in every
// step is the FIRST TOUCH of the step's activation lines (all column tiles of a row block run the same K-step at the same time: nobody
// has fetched the line earlier), and two 64 KB stages hold at most one and a half steps of requests.  Vector-memory operations of a wave
// complete in order, so a wave that waits for its youngest just-in-time piece has also waited for every older far-ahead one: the two
// kinds of requests therefore come from DIFFERENT waves -- waves 0 .. 3 issue all activation pieces (step t + 3's head behind step t's
// barrier, its rest under step t + 1's first rows: two steps in flight behind the one being waited for, vmcnt(8 + stores)), waves
// 4 .. 7 all weight pieces on the two-slot timeline of the paced kernel (weights are L2 hits: every row block re-reads them).
// Fragment reads, MFMA order, rounding and the paced stores are the paced kernel's, bit for bit.
#pragma once
#include "gemm_ps_kernels.h"

namespace aumg {

constexpr int R3_LDS_BYTES = 5 * TILE_BYTES;        // 160 KB
constexpr int R3_B_BASE = 3 * TILE_BYTES;
#ifndef AUM_R3_DEEP_B
#define AUM_R3_DEEP_B 0      // 1: the WEIGHT tiles get the ring of three (experiment)
#endif

template <bool BF16>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn_r3(GemmLaunch L) {
    __shared__ __attribute__((aligned(1024))) char lds[R3_LDS_BYTES];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    const int ntn = g.n / BN, grid = (int)gridDim.x, nk = g.k / BK;

    // staging (gemm_kernels.h): piece c = j * 8 + w is rows 8 c .. 8 c + 7 of the operand tile, lane l fills slot l & 7 of row 8 c + (l >> 3)
    // ROLES: waves 0 .. 3 fetch the activation tiles (three slots: their pieces run up to three steps ahead), waves 4 .. 7 the weight
    // tiles (two slots).  A wave's eight pieces of a step: piece n is rows 64 (n >> 1) + 8 (w4 + 4 (n & 1)) .. + 7 of its operand's tile.
    const bool isA = w < 4;
    const int w4 = w & 3;
    const int srow = w4 * 8 + (lane >> 3);
    const int f_a = ((w4 & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = (w4 << 1) | ((lane >> 4) & 1);
    const int ldo = isA ? g.lda : g.ldb;
    const int voff0 = srow * ldo * 2 + (((lane & 7) ^ (isA ? f_a : f_b)) << 4);
    const int voff1 = voff0 + 32 * ldo * 2;
    const int rowstep = 64 * ldo * 2;
    // fragment reads: lane = (operand row rho, k-group kg)
    const int rho = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + rho) * 128 + ((kg ^ ((lane >> 1) & 7)) << 4);                                  // + i * 2048, ^ 64 for the second half of K
    const int b_row = wc * 64 + (rho >> 2) * 8 + (rho & 3);
    const int b_rd = b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);     // + b_joff(j), ^ 64
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
    __amdgpu_buffer_rsrc_t ro = isA ? rsrc_a(m0, rows) : rsrc_b(n0);          // this wave's operand of the current item

    // piece n (0 .. 7) of this wave's operand, K offset kbyte, into slot `dst`
    auto piece = [&](__amdgpu_buffer_rsrc_t r_s, int kbyte, char* dst, int n) {
        if (AUM_PS_ABL & 1) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_s, (lds_ptr_t)(dst + ((n >> 1) * 8 + w4 + (n & 1) * 4) * 1024), 16, (n & 1) ? voff1 : voff0,
                                                 kbyte + (n >> 1) * rowstep, 0, 0);
    };
    // slot offsets: activations of steps t, t + 1, t + 2 (ring of three from 0), weights of steps t, t + 1 (ring of two from R3_B_BASE)
    // (od*: the deep operand's ring of three, os*: the other operand's two slots)
    int od0 = 0, od1 = TILE_BYTES, od2 = 2 * TILE_BYTES, os0 = R3_B_BASE, os1 = R3_B_BASE + TILE_BYTES;
    const bool deep = AUM_R3_DEEP_B ? !isA : isA;
    const int la = deep ? 1 : 0;         // the deep operand's pieces run one step further ahead

    s8v bfA[4], bfB[4], af[4];
    u4v pend[6][2];                     // fragment rows 2 .. 7 of the finished tile, rounded
#pragma unroll
    for (int i = 0; i < 6; ++i) pend[i][0] = pend[i][1] = u4v{0u, 0u, 0u, 0u};
    f4v acc[8][4];

    // ---- prologue: step 0 of the first tile lands, the first four pieces of step 1 leave, the first fragments of step 0 are read
    // (k >= 7 K-steps: steps 1 and 2 are steps of the first item)
#pragma unroll
    for (int n = 0; n < 8; ++n) piece(ro, 0, lds + (deep ? od0 : os0), n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (deep) {
#pragma unroll
        for (int n = 0; n < 8; ++n) piece(ro, BK * 2, lds + od1, n);
#pragma unroll
        for (int n = 0; n < 4; ++n) piece(ro, 2 * BK * 2, lds + od2, n);
    } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) piece(ro, BK * 2, lds + os1, n);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bfA[j] = lds_frag(lds + (AUM_R3_DEEP_B ? od0 : os0), b_rd + b_joff(j));
#pragma unroll
    for (int r = 0; r < 3; ++r) af[r] = lds_frag(lds + (AUM_R3_DEEP_B ? os0 : od0), a_rd + r * 2048);
    __amdgpu_buffer_rsrc_t rc_prev = r_null;        // C rows of the tile whose rounded values wait in `pend`

    while (true) {
        const int nid = id + grid;
        const bool has_next = nid < L.nitems;
        int m1 = m0, n1 = n0, rows1 = rows;
        if (has_next) ps_item(L, nid, ntn, grid, m1, n1, rows1);
        const __amdgpu_buffer_rsrc_t ro_n = has_next ? (isA ? rsrc_a(m1, rows1) : rsrc_b(n1)) : r_null;
        const __amdgpu_buffer_rsrc_t rc = (AUM_PS_ABL & 8) ? r_null : rsrc_c(m0, n0, rows);

        // One K-step.  FIRST: a tile's first step (zero addend in its first half); ST >= 0: the step carries stores 2 ST, 2 ST + 1 of the
        // previous tile; LAST: the tile's last step (rows rounded into `pend` behind the second half's MFMAs).
        auto kstep = [&](auto first_c, auto st_c, auto last_c, int t) {
            constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
            constexpr int ST = decltype(st_c)::value;
            const char* curA = lds + (AUM_R3_DEEP_B ? os0 : od0);
            const char* curB = lds + (AUM_R3_DEEP_B ? od0 : os0);
            const char* nxtA = lds + (AUM_R3_DEEP_B ? os1 : od1);
            const char* nxtB = lds + (AUM_R3_DEEP_B ? od1 : os1);
            // the pieces this wave issues in this step: the rest of step s1 = t + 1 (+ 1 for the activation waves) under rows 0, 1 into the
            // slot that step t - 1 left (activations: slot 2, weights: the other slot), the head of step s1 + 1 under rows 13 .. 15 into the
            // slot this step has finished with
            const int s1 = t + 1 + la, s2 = s1 + 1;
            const bool same1 = s1 < nk, same2 = s2 < nk;
            const __amdgpu_buffer_rsrc_t r_1 = same1 ? ro : ro_n, r_2 = same2 ? ro : ro_n;
            const int kb1 = (same1 ? s1 : s1 - nk) * (BK * 2), kb2 = (same2 ? s2 : s2 - nk) * (BK * 2);
            char* dst1 = lds + (deep ? od2 : os1);
            char* dst2 = lds + (deep ? od0 : os0);
            // ---- first half: fragment rows g = 0 .. 7 on bfA
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FIRST) ps_mfma0<BF16>(acc[i][j], bfA[j], af[i & 3]);
                    else ps_mfma<BF16>(acc[i][j], bfA[j], af[i & 3]);
                    if (j == 1) {                                   // the activation fragment of row g + 3 into the slot row g - 1 has left
                        if (i + 3 < 8) af[(i + 3) & 3] = lds_frag(curA, a_rd + (i + 3) * 2048);
                        else af[(i + 3) & 3] = lds_frag(curA, (a_rd ^ 64) + (i + 3 - 8) * 2048);
                    }
                    if (j == 3 && i >= 4) bfB[i - 4] = lds_frag(curB, (b_rd ^ 64) + b_joff(i - 4));      // the second half's weight fragments
                    if (i < 2 && (j == 0 || j == 2)) piece(r_1, kb1, dst1, 4 + i * 2 + (j >> 1));
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
                    // (an activation wave's eight pieces of step t + 2 stay in flight as well: only step t + 1's must have landed)
                    constexpr int NSTO = (AUM_PS_ABL & 2) ? 0 : ST >= 0 ? 2 : LAST ? 4 : 0;
                    if (deep) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO + ((AUM_PS_ABL & 1) ? 0 : 8)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ps_mfma<BF16>(acc[i][j], bfB[j], af[i & 3]);
                    if (j == 1 && i + 3 < 8) af[(i + 3) & 3] = lds_frag(curA, (a_rd ^ 64) + (i + 3) * 2048);
                    if (i >= 5) {
                        // under rows 13 .. 15: the next step's first fragments (its stage has landed) and the first four pieces of the step
                        // after it (into the stage this step has finished with).  Ring slots: rows 13, 14, 15 multiply on slots 1, 2, 3 --
                        // slot 0 is free, slot 1 once row 13's MFMAs have been issued, slot 2 after row 14's
                        const int q = (i - 5) * 4 + j;              // 0 .. 11
                        if (q == 0 || q == 3 || q == 6 || q == 9) piece(r_2, kb2, dst2, q / 3);
                        if (q == 1) bfA[0] = lds_frag(nxtB, b_rd + b_joff(0));
                        if (q == 2) af[0] = lds_frag(nxtA, a_rd);
                        if (q == 4) bfA[1] = lds_frag(nxtB, b_rd + b_joff(1));
                        if (q == 5) af[1] = lds_frag(nxtA, a_rd + 2048);
                        if (q == 7) bfA[2] = lds_frag(nxtB, b_rd + b_joff(2));
                        if (q == 8) bfA[3] = lds_frag(nxtB, b_rd + b_joff(3));
                        if (q == 10) af[2] = lds_frag(nxtA, a_rd + 2 * 2048);
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
            { const int o = od0; od0 = od1; od1 = od2; od2 = o; }
            { const int o = os0; os0 = os1; os1 = o; }
        };
        // A wave whose 128 rows all lie beyond the item's rows (the lower wave row of a half item or of a ragged row block of at most 128
        // rows -- always the workgroup's LAST item: both kinds sit in the last round) has nothing to multiply: it keeps its share of the
        // pieces, the previous tile's paced stores and the barriers, and leaves the SIMD's matrix pipe to the wave that has rows.
        if (AUM_PS_DEAD_WAVES && wr == 1 && rows <= BM / 2 && !has_next) {
            auto kdead = [&](auto st_c, int t) {
                constexpr int ST = decltype(st_c)::value;
                const int s1 = t + 1 + la, s2 = s1 + 1;
                const bool same1 = s1 < nk, same2 = s2 < nk;
                char* dst1 = lds + (deep ? od2 : os1);
                char* dst2 = lds + (deep ? od0 : os0);
#pragma unroll
                for (int n = 4; n < 8; ++n) piece(same1 ? ro : r_null, (same1 ? s1 : 0) * (BK * 2), dst1, n);
                if constexpr (ST >= 0 && !(AUM_PS_ABL & 2)) {
                    __builtin_amdgcn_raw_buffer_store_b128(pend[ST][0], rc_prev, c_voff + (ST + 2) * 16 * ldc2, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pend[ST][1], rc_prev, c_voff + (ST + 2) * 16 * ldc2 + 64, 0, 0);
                }
                constexpr int NSTO = (AUM_PS_ABL & 2) ? 0 : ST >= 0 ? 2 : 0;
                if (deep) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO + ((AUM_PS_ABL & 1) ? 0 : 8)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTO) : "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int n = 0; n < 4; ++n) piece(same2 ? ro : r_null, (same2 ? s2 : 0) * (BK * 2), dst2, n);
                { const int o = od0; od0 = od1; od1 = od2; od2 = o; }
                { const int o = os0; os0 = os1; os1 = o; }
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
        ro = ro_n;
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
