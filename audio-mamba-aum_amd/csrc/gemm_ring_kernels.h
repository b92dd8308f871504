// gemm_ring_kernels.h -- aum_gemm_tn as ONE continuous stream of 32-deep K-steps through a five-stage LDS ring (round 5; AUM_GEMM_RING).
//
// What bounded the earlier forms (gemm_kernels.h: 8 waves, 256 x 256 x 64, two 64 KB buffers; gemm_w4_kernels.h: 4 waves, 256 x 192 x 64,
// two 56 KB buffers): a K-step took 1.3-1.5 us on every shape and for either division of the work -- the time of ONE global -> LDS round
// trip under load, because the pieces of step t + 1 are requested during step t and nothing can start before they land; the matrix pipe
// needs 0.64-0.85 us for the step.  Here the step is 32 deep, a stage is 28 KB, and FIVE stages fit: the pieces of step t + 5 are
// requested during step t, so a round trip may take four steps before anybody waits.
//
//   * workgroup = 4 waves (one per SIMD), tile 256 x 192; wave (wr, wc) of the 2 x 2 grid owns 128 rows x 96 columns = 8 x 6 accumulator
//     fragments of v_mfma_f32_16x16x32_bf16 in 192 AGPRs, updated in place by inline assembly with tied operands (gemm_w4_kernels.h);
//   * a stage = 256 + 192 rows of 64 bytes; a DMA piece (buffer_load_dwordx4 ... lds, 1 KB) is 16 rows: lane l fills physical 16-byte
//     slot l & 3 of row 16 c + (l >> 2) with k-slot (l & 3) ^ f(row).  ds_read_b128 serves 16 lanes per cycle over 256 bytes = FOUR rows,
//     so f has to separate rows 4 apart: f_A(row) = G[(row >> 2) & 3], f_B(row) = G[(row >> 3) & 3] with G = {0, 3, 2, 1} (the weight
//     fragment's 16 rows are {8 q + e}: q takes the place of row >> 2) -- every service group of every fragment read touches 16 distinct
//     slots (tests/test_gemm_layout.py::test_ring_tile_product_and_bank_slots restates it lane by lane);
//   * a step = [counted wait: my pieces of step t + 1 landed, my fragment reads of step t returned] [workgroup barrier] [48 MFMAs on
//     fragment set t & 1, and between them: the 14 fragment reads of step t + 1 into the other set, the 7 pieces of step t + 5 into the
//     stage step t was read from].  Memory operations retire in issue order, so the wait names the operations YOUNGER than the pieces
//     it needs: three steps' pieces (21), plus the previous tile's 24 stores during a tile's first four steps (45);
//   * persistent: one workgroup per CU walks its tiles as one stream -- the ring does not drain at a tile boundary (the last five steps
//     of a tile request the first five of the next).  A tile's first step multiplies with a zero addend (no accumulator fill); its last
//     step rounds and stores fragment row i - 1 between the MFMAs of row i (the accumulators of a row are final once its MFMAs of the
//     last step have been issued; buffer stores with the tile's row range: rows beyond M are dropped by the range check, so every tile
//     issues the same 24 stores per wave).
#pragma once
#include "gemm_w4_kernels.h"

namespace aumg {

#ifndef AUM_RING_ABL
#define AUM_RING_ABL 0      // timing experiments only (wrong results): 1 no DMA pieces inside the steps, 2 no fragment reads inside the steps, 4 no barrier, 8 no MFMAs
#endif
constexpr int RING_BK = 32, RING_NJ = 6, RING_BN = 32 * RING_NJ;
constexpr int RING_TA = BM * RING_BK * 2, RING_TB = RING_BN * RING_BK * 2;        // 16 KB + 12 KB
constexpr int RING_STG = RING_TA + RING_TB, RING_S = 5;
constexpr int RING_NDMA = 7, RING_NST = 4 * RING_NJ;                              // per wave: pieces per step, stores per tile

template <bool BF16> __device__ __forceinline__ void mfma_first(f4v& c, const s8v& a, const s8v& b) {
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ constexpr int ring_b_joff(int j) { return ((j >> 1) * 32 + (j & 1) * 4) * 64; }
template <int V> struct IntC { static constexpr int value = V; };

template <bool BF16>
__global__ __launch_bounds__(W4_THREADS, 1) void k_gemm_tn_ring(GemmLaunch L) {
    constexpr int NJ = RING_NJ;
    __shared__ __attribute__((aligned(1024))) char lds[RING_S * RING_STG];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 1, wc = w & 1;
    const int ntn = g.n / RING_BN, grid = (int)gridDim.x, nk = g.k / RING_BK;
    constexpr int G4 = 0x1230;            // G = {0, 3, 2, 1} as nibbles: G[i] = (G4 >> 4 i) & 3
    auto Gf = [](int i) { return (G4 >> (4 * i)) & 3; };

    // staging: piece c = 4 j + w (A: j < 4, B: j < 3) is rows 16 c .. 16 c + 15; this lane fills slot (lane & 3) of row 16 c + (lane >> 2)
    const int srow = w * 16 + (lane >> 2);
    const int f_a = Gf((lane >> 4) & 3);                                  // (row >> 2) & 3 = lane >> 4
    const int f_b = Gf((2 * (w & 1) + (lane >> 5)) & 3);                  // (row >> 3) & 3 = (2 c + (lane >> 5)) & 3, c & 1 = w & 1
    const int voff_a = srow * g.lda * 2 + (((lane & 3) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 3) ^ f_b) << 4);
    const int rowstep_a = 64 * g.lda * 2, rowstep_b = 64 * g.ldb * 2;
    // fragment reads: lane = (operand row rho, k-group kg)
    const int rho = lane & 15, kg = lane >> 4;
    const int swz = (kg ^ Gf(rho >> 2)) << 4;
    const int a_rd = (wr * 128 + rho) * 64 + swz;                                                   // + i * 1024
    const int b_rd = RING_TA + (wc * (16 * NJ) + (rho >> 2) * 8 + (rho & 3)) * 64 + swz;            // + ring_b_joff(j)
    const int c_lane = (wc * (16 * NJ) + kg * 8) * 2;                                               // byte offset of this lane's columns in a row of the tile

    auto rsrc_a = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.a) + (int64_t)it.m0 * g.lda * 2), 0,
                                                 it.rows * g.lda * 2, 0x00020000);
    };
    auto rsrc_b = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.b) + (int64_t)it.n0 * g.ldb * 2), 0,
                                                 RING_BN * g.ldb * 2, 0x00020000);
    };
    auto rsrc_c = [&](const GemmItem& it) {          // the tile's rows of C from column n0 on: rows beyond `rows` are out of range (stores dropped)
        return __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(g.c) + ((int64_t)it.m0 * g.ldc + it.n0) * 2, 0,
                                                 (it.rows - 1) * g.ldc * 2 + RING_BN * 2, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), 0, 0, 0x00020000);

    int id = (int)blockIdx.x;
    if (id >= L.nitems) return;
    GemmItem it = w4_item(L, id, ntn, grid, RING_BN);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(it), rb = rsrc_b(it);

    auto piece = [&](__amdgpu_buffer_rsrc_t ra_s, __amdgpu_buffer_rsrc_t rb_s, int kbyte, char* dst, int n) {      // n < 4: A, else B
        if (n < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_s, (lds_ptr_t)(dst + (n * 4 + w) * 1024), 16, voff_a, kbyte + n * rowstep_a, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_s, (lds_ptr_t)(dst + RING_TA + ((n - 4) * 4 + w) * 1024), 16, voff_b,
                                                      kbyte + (n - 4) * rowstep_b, 0, 0);
    };
    s8v bf[2][NJ], af[2][8];
    auto read_n = [&](int set, const char* stg, int n) {        // weight fragments first: a fragment row needs all NJ of them
        if (n < NJ) bf[set][n] = lds_frag(stg, b_rd + ring_b_joff(n));
        else af[set][n - NJ] = lds_frag(stg, a_rd + (n - NJ) * 1024);
    };

    // ---- fill the ring: steps 0 .. 4 of the first tile; 24 stores that the range check drops, so that the first tile's waits count the
    //      same operations as every other tile's (behind its steps' pieces: the previous tile's 24 stores)
    for (int s = 0; s < RING_S; ++s)
#pragma unroll
        for (int n = 0; n < RING_NDMA; ++n) piece(ra, rb, s * (RING_BK * 2), lds + s * RING_STG, n);
#pragma unroll
    for (int q = 0; q < RING_NST; ++q) __builtin_amdgcn_raw_buffer_store_b128(u4v{0u, 0u, 0u, 0u}, r_null, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * RING_NDMA + RING_NST) : "memory");        // step 0 landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int n = 0; n < 8 + NJ; ++n) read_n(0, lds, n);
    int cur = 0;                  // the stage of the step that is about to run

    f4v acc[8][NJ];
    while (true) {
        const int nid = id + grid;
        const bool has_next = nid < L.nitems;
        GemmItem itn = it;
        if (has_next) itn = w4_item(L, nid, ntn, grid, RING_BN);
        const __amdgpu_buffer_rsrc_t ra_n = has_next ? rsrc_a(itn) : r_null, rb_n = has_next ? rsrc_b(itn) : r_null;
        const __amdgpu_buffer_rsrc_t rc = rsrc_c(it);
        const int ldc2 = g.ldc * 2;

        // one step.  MODE 0: a tile's first step (zero addend), 1: a middle step, 2: the last step (rows rounded and stored between the MFMAs)
        auto step = [&](auto mode_c, auto wait_c, auto set_c, int t) {
            constexpr int MODE = decltype(mode_c)::value, WAIT = decltype(wait_c)::value, SET = decltype(set_c)::value;
            if (AUM_RING_ABL & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAIT) : "memory");
            if (!(AUM_RING_ABL & 4)) __builtin_amdgcn_s_barrier();
            const int s5 = t + RING_S;                                    // the step whose pieces are requested now
            const bool same = s5 < nk;
            const __amdgpu_buffer_rsrc_t ra_s = same ? ra : ra_n, rb_s = same ? rb : rb_n;
            const int kb = (same ? s5 : s5 - nk) * (RING_BK * 2);
            char* fill = lds + cur * RING_STG;
            const int nx = cur + 1 == RING_S ? 0 : cur + 1;
            const char* nxt = lds + nx * RING_STG;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (AUM_RING_ABL & 8) {          // no MFMAs: the DMA stream (and the stores) alone
                        if constexpr (MODE == 0) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
                    } else if constexpr (MODE == 0) mfma_first<BF16>(acc[i][j], bf[SET][j], af[SET][i]);
                    else mfma_acc<BF16>(acc[i][j], bf[SET][j], af[SET][i]);
                    const int m = i * NJ + j;
                    // side operations, one per two MFMAs: the 7 pieces first (they have the longest way), then the 14 fragment reads.  In a
                    // tile's last step every piece is issued before the first store (after fragment row 1): the next tile's counted
                    // waits rely on "all of a step's pieces, THEN the 24 stores"
                    if constexpr (!(AUM_RING_ABL & 1)) {
                        if constexpr (MODE == 2) {
                            if (m < RING_NDMA) piece(ra_s, rb_s, kb, fill, m);
                        } else {
                            if (m % 2 == 0 && m / 2 < RING_NDMA) piece(ra_s, rb_s, kb, fill, m / 2);
                        }
                    }
                    if constexpr (!(AUM_RING_ABL & 2))
                        if (m % 2 == 0 && m / 2 >= RING_NDMA && m / 2 < RING_NDMA + 8 + NJ) read_n(SET ^ 1, nxt, m / 2 - RING_NDMA);
                }
                if constexpr (MODE == 2) {
                    if (i > 0) {          // fragment row i - 1 is final (its last MFMAs were issued NJ MFMAs ago): round it and store it
                        // (its vector-ALU reads must stay BEHIND row i's MFMAs -- the compiler knows nothing about the latency of the
                        // assembly that produced the values and would hoist them right behind it: an empty volatile statement that
                        // "rewrites" the row pins them here)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i - 1][j]));
                        const int voff = (wr * 128 + (i - 1) * 16 + rho) * ldc2 + c_lane;
#pragma unroll
                        for (int jp = 0; jp < NJ / 2; ++jp) {
                            u4v v;
                            v.x = pack2<BF16>(acc[i - 1][2 * jp][0], acc[i - 1][2 * jp][1]);
                            v.y = pack2<BF16>(acc[i - 1][2 * jp][2], acc[i - 1][2 * jp][3]);
                            v.z = pack2<BF16>(acc[i - 1][2 * jp + 1][0], acc[i - 1][2 * jp + 1][1]);
                            v.w = pack2<BF16>(acc[i - 1][2 * jp + 1][2], acc[i - 1][2 * jp + 1][3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rc, voff + jp * 64, 0, 0);
                        }
                    }
                }
            }
            if constexpr (MODE == 2) {
                // last MFMA -> vector-ALU reads of its accumulators: 18 wait states, and the reads pinned behind them
                asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]));
                const int voff = (wr * 128 + 7 * 16 + rho) * ldc2 + c_lane;
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    u4v v;
                    v.x = pack2<BF16>(acc[7][2 * jp][0], acc[7][2 * jp][1]);
                    v.y = pack2<BF16>(acc[7][2 * jp][2], acc[7][2 * jp][3]);
                    v.z = pack2<BF16>(acc[7][2 * jp + 1][0], acc[7][2 * jp + 1][1]);
                    v.w = pack2<BF16>(acc[7][2 * jp + 1][2], acc[7][2 * jp + 1][3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rc, voff + jp * 64, 0, 0);
                }
            }
            cur = nx;
        };
        constexpr int W0 = 3 * RING_NDMA + RING_NST, W1 = 3 * RING_NDMA;       // younger operations behind the pieces a step waits for
        step(IntC<0>{}, IntC<W0>{}, IntC<0>{}, 0);
        step(IntC<1>{}, IntC<W0>{}, IntC<1>{}, 1);
        step(IntC<1>{}, IntC<W0>{}, IntC<0>{}, 2);
        step(IntC<1>{}, IntC<W0>{}, IntC<1>{}, 3);
        for (int t = 4; t + 2 < nk; t += 2) {            // nk is even: steps 4 .. nk - 3 in pairs
            step(IntC<1>{}, IntC<W1>{}, IntC<0>{}, t);
            step(IntC<1>{}, IntC<W1>{}, IntC<1>{}, t + 1);
        }
        step(IntC<1>{}, IntC<W1>{}, IntC<0>{}, nk - 2);
        step(IntC<2>{}, IntC<W1>{}, IntC<1>{}, nk - 1);
        if (!has_next) break;
        id = nid;
        it = itn;
        ra = ra_n;
        rb = rb_n;
    }
}

}  // namespace aumg
