// gemm_w4_kernels.h -- aum_gemm_tn on FOUR waves (round 5; AUM_GEMM_W4): the same 256 x 256 x 64 workgroup tile, LDS image, swizzles and MFMA
// roles as gemm_kernels.h, divided the other way.
//
//   gemm_kernels.h   8 waves, 2 per SIMD, wave tile 128 x 64   (128 accumulator registers, 24 fragment reads per 64 MFMAs and K-step)
//   here             4 waves, 1 per SIMD, wave tile 128 x 128  (256 accumulator registers = the AGPR half of the 512-register file a lone
//                    wave owns, 32 fragment reads per 128 MFMAs and K-step: a third less LDS traffic per flop; nobody to share the SIMD's
//                    matrix pipe with, so the wave has to cover its own fragment reads -- the next half K-step's fragments are read under
//                    the current half's 64 MFMAs, two register sets)
// Persistent (one workgroup per CU walking tiles round-robin, the 32 workgroups of an XCD on consecutive tiles), the next tile's first K-step
// fetched during the current tile's last one, the tile's first wait counted so that the previous tile's 32 stores per wave stay in flight.
// Ragged last row block: a full item whose rows beyond M read as zero (buffer range) and are not stored.
#pragma once
#include "gemm_kernels.h"

namespace aumg {

constexpr int W4_THREADS = 256;

// acc += a . b with the accumulator NAMED as an AGPR tile updated in place.  Through the builtin, hipcc selects the AGPR form of the
// instruction with an early-clobber destination and then rotates the 192 accumulators of the K loop through copies (264 v_accvgpr moves
// per 96 MFMAs, half of the MFMAs writing a different tile than they read): the tied "+a" operand leaves it no choice.  Inline assembly is
// outside the compiler's hazard recogniser: the only hazards here are a vector-ALU write of an accumulator before its first MFMA (the
// zero fill) and the vector-ALU reads of the finished tile behind the last MFMA -- both covered by explicit s_nop at those two places.
template <bool BF16> __device__ __forceinline__ void mfma_acc(f4v& c, const s8v& a, const s8v& b) {
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// the 16 pieces (8 of A, 8 of B) wave w of four contributes to one K-step: piece c = j * 4 + w is rows 8 c .. 8 c + 7 of the tile
// work item id -> tile: every item is a 256-row block (the last one may be ragged) x one column tile of `bnw` columns; in a complete round
// of `grid` items XCD x (= workgroup % 8) takes grid / 8 consecutive tiles (tiles of one row block next to each other: A comes from HBM once
// per XCD-resident row block, the weight stays in L2 / MALL)
__device__ __forceinline__ GemmItem w4_item(const GemmLaunch& L, int id, int ntn, int grid, int bnw) {
    int tile = id;
    const int r0 = id / grid * grid;
    if ((grid & 7) == 0 && r0 + grid <= L.nitems) {
        const int q = id - r0;
        tile = r0 + (q & 7) * (grid >> 3) + (q >> 3);
    }
    GemmItem it;
    const int tm = tile / ntn;
    it.m0 = tm * BM;
    it.n0 = (tile - tm * ntn) * bnw;
    it.rows = L.g.m - it.m0 < BM ? L.g.m - it.m0 : BM;
    it.half = false;
    return it;
}

template <int NJ>
__device__ __forceinline__ void stage4(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voff_a, int voff_b, int kbyte, int rowstep_a,
                                       int rowstep_b, char* lds_stage, int w) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds_stage + (j * 4 + w) * 1024), 16, voff_a, kbyte + j * rowstep_a, 0, 0);
        if (j < NJ)         // the weight tile has 32 NJ rows: NJ pieces per wave
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds_stage + TILE_BYTES + (j * 4 + w) * 1024), 16, voff_b,
                                                     kbyte + j * rowstep_b, 0, 0);
    }
}

// NJ: weight fragments per wave -- 8: 256-column workgroup tiles (256 accumulator registers: every AGPR, and the register allocator starts
// moving accumulators through the vector registers); 6: 192-column tiles (192 accumulators, 2 x 56 KB of LDS): the form that is used.
// N = 768 / 1536 / 3072 are 4 / 8 / 16 tiles of 192 columns: at 64 x 513 tokens N = 768 is 516 tiles on 256 CUs -- two rounds of
// three-quarter tiles instead of one and a half rounds of whole ones (the half-empty round that kept these GEMMs with the library).
template <bool BF16, int NJ>
__global__ __launch_bounds__(W4_THREADS, 1) void k_gemm_tn_w4(GemmLaunch L) {
    constexpr int BNW = 32 * NJ;          // columns of a workgroup tile (two waves side by side, 16 NJ each)
    constexpr int W4_STAGE = TILE_BYTES + BNW * BK * 2;
    __shared__ __attribute__((aligned(1024))) char lds[2 * W4_STAGE];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 1, wc = w & 1;
    const int ntn = g.n / BNW, grid = (int)gridDim.x, nk = g.k / BK;

    // staging: this lane fills slot (lane & 7) of row 8 c + (lane >> 3), c = 4 j + w: f_A / f_B of that row do not depend on j
    const int srow = w * 8 + (lane >> 3);
    const int f_a = ((w & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = ((w & 3) << 1) | ((lane >> 4) & 1);
    const int voff_a = srow * g.lda * 2 + (((lane & 7) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 7) ^ f_b) << 4);
    const int rowstep_a = 32 * g.lda * 2, rowstep_b = 32 * g.ldb * 2;
    // fragment reads: lane = (operand row rho, k-group kg)
    const int rho = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + rho) * 128 + ((kg ^ ((lane >> 1) & 7)) << 4);                                   // + i * 2048, ^ 64 for the second half of K
    const int b_row = wc * (16 * NJ) + (rho >> 2) * 8 + (rho & 3);                                                     // + (j >> 1) * 32 + (j & 1) * 4
    const int b_rd = TILE_BYTES + b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);      // + b_joff(j), ^ 64

    auto rsrc_a = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.a) + (int64_t)it.m0 * g.lda * 2), 0,
                                                 it.rows * g.lda * 2, 0x00020000);
    };
    auto rsrc_b = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.b) + (int64_t)it.n0 * g.ldb * 2), 0,
                                                 BNW * g.ldb * 2, 0x00020000);
    };

    int id = (int)blockIdx.x;
    if (id >= L.nitems) return;
    GemmItem it = w4_item(L, id, ntn, grid, BNW);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(it), rb = rsrc_b(it);
    int par = 0;                    // the buffer that holds (or is receiving) step 0 of the current tile
    bool prev_full = false;         // the previous tile of this workgroup issued all of its 32 stores per wave
    stage4<NJ>(ra, rb, voff_a, voff_b, 0, rowstep_a, rowstep_b, lds, w);
    while (true) {
        const int nid = id + grid;
        const bool has_next = nid < L.nitems;
        GemmItem itn = it;
        if (has_next) itn = w4_item(L, nid, ntn, grid, BNW);
        // (no next tile: range 0 -- the fetch behind the last-but-one step then reads zeros into the free buffer)
        const __amdgpu_buffer_rsrc_t ra_n = has_next ? rsrc_a(itn) : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), 0, 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_n = has_next ? rsrc_b(itn) : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.b), 0, 0, 0x00020000);

        f4v acc[8][NJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 7" ::: "memory");          // zero fill (vector ALU) -> first MFMA reading the tile as its addend

        // step 0 of this tile has landed (memory operations retire in issue order: behind its pieces there are at most the previous
        // tile's 4 NJ stores of this wave), and everybody is done with the buffer step 1 goes into
        if (prev_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage4<NJ>(ra, rb, voff_a, voff_b, BK * 2, rowstep_a, rowstep_b, lds + (par ^ 1) * W4_STAGE, w);
        s8v bf[2][NJ], af[2][8];
        {
            const char* s0 = lds + par * W4_STAGE;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[0][j] = lds_frag(s0, b_rd + b_joff(j));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[0][i] = lds_frag(s0, a_rd + i * 2048);
        }
        // Steps that have a successor: ONE basic block (a branch inside the loop makes the register allocator move accumulators around).
        // What is fetched behind the barrier -- step t + 2, or the next tile's step 0 during the last-but-one step, or nothing (a descriptor
        // of range 0 reads as zeros into the free buffer) -- is a scalar select.
        // A lone wave has nobody to cover its issue slots: every fragment read and every DMA piece is PLACED between MFMAs (volatile
        // assembly keeps its order, and memory operations do not move across it): one side operation per three MFMAs.
        //   first half  (fragments 0: 8 NJ MFMAs)   reads of fragments 1 (the second 32 k of this step)
        //   barrier: step t + 1 has landed everywhere, step t is read
        //   second half (fragments 1: 8 NJ MFMAs)   the 8 + NJ pieces of step t + 2 and the reads of fragments 0 of step t + 1, alternating
        constexpr int NSIDE = 8 + NJ;
        auto read_n = [&](int set, const char* stg, int kx, int n) {       // weight fragments first: a fragment row needs all NJ of them
            if (n < NJ) bf[set][n] = lds_frag(stg, (b_rd ^ kx) + b_joff(n));
            else af[set][n - NJ] = lds_frag(stg, (a_rd ^ kx) + (n - NJ) * 2048);
        };
        auto piece_n = [&](__amdgpu_buffer_rsrc_t ra_s, __amdgpu_buffer_rsrc_t rb_s, int kbyte, char* dst, int n) {
            if (n < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_s, (lds_ptr_t)(dst + (n * 4 + w) * 1024), 16, voff_a, kbyte + n * rowstep_a, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_s, (lds_ptr_t)(dst + TILE_BYTES + ((n - 8) * 4 + w) * 1024), 16, voff_b,
                                                          kbyte + (n - 8) * rowstep_b, 0, 0);
        };
        for (int t = 0; t + 1 < nk; ++t) {
            const char* st = lds + ((par + t) & 1) * W4_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    mfma_acc<BF16>(acc[i][j], bf[0][j], af[0][i]);
                    const int m = i * NJ + j;
                    if (m % 3 == 1 && m / 3 < NSIDE) read_n(1, st, 64, m / 3);
                }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of step t + 1 landed, its reads of step t returned
            __builtin_amdgcn_s_barrier();                                    // ... everybody's
            const bool more2 = t + 2 < nk;
            const __amdgpu_buffer_rsrc_t ra_s = more2 ? ra : ra_n, rb_s = more2 ? rb : rb_n;
            const int kb = more2 ? (t + 2) * (BK * 2) : 0;
            char* fr = lds + ((par + t) & 1) * W4_STAGE;                     // the buffer step t was read from
            const char* sn = lds + ((par + t + 1) & 1) * W4_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    mfma_acc<BF16>(acc[i][j], bf[1][j], af[1][i]);
                    const int m = i * NJ + j;
                    if (m % 3 == 0 && m / 3 < NSIDE) piece_n(ra_s, rb_s, kb, fr, m / 3);
                    if (m % 3 == 1 && m / 3 < NSIDE) read_n(0, sn, 0, m / 3);
                }
        }
        {   // the last step: nothing left to fetch or to wait for
            const char* st = lds + ((par + nk - 1) & 1) * W4_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    mfma_acc<BF16>(acc[i][j], bf[0][j], af[0][i]);
                    const int m = i * NJ + j;
                    if (m % 3 == 1 && m / 3 < NSIDE) read_n(1, st, 64, m / 3);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mfma_acc<BF16>(acc[i][j], bf[1][j], af[1][i]);
        }
        // last MFMA -> vector-ALU reads of the accumulators (at most 18 wait states); the last fragment row is named so that its reads
        // cannot be scheduled in front of the wait (the rows before it are older than that anyway)
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][NJ - 2]), "+a"(acc[7][NJ - 1]));
        // store: lane holds, for fragment row i, columns wc * 128 + 32 (j >> 1) + 8 kg + 4 (j & 1) + r (r = 0..3) of row wr * 128 + 16 i + rho
        char* c_rows = static_cast<char*>(g.c) + ((int64_t)it.m0 * g.ldc + it.n0 + wc * (16 * NJ) + kg * 8) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wr * 128 + i * 16 + rho;
            if (row < it.rows) {
                u4v* dst = reinterpret_cast<u4v*>(c_rows + (int64_t)row * g.ldc * 2);
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    u4v v;
                    v.x = pack2<BF16>(acc[i][2 * jp][0], acc[i][2 * jp][1]);
                    v.y = pack2<BF16>(acc[i][2 * jp][2], acc[i][2 * jp][3]);
                    v.z = pack2<BF16>(acc[i][2 * jp + 1][0], acc[i][2 * jp + 1][1]);
                    v.w = pack2<BF16>(acc[i][2 * jp + 1][2], acc[i][2 * jp + 1][3]);
                    dst[4 * jp] = v;
                }
            }
        }
        if (!has_next) break;
        prev_full = it.rows == BM;
        id = nid;
        it = itn;
        ra = ra_n;
        rb = rb_n;
        par = (par + nk) & 1;
    }
}

}  // namespace aumg
