// gemm_kernels.h -- the dense in_proj / out_proj GEMMs of the Mamba block as a hand-written MFMA kernel for gfx950 (round 3; ABI 9).
//
//   C[M][N] (16-bit) = A[M][K] . B[N][K]^T        A, B row-major with K contiguous ("TN"), fp32 accumulation, one rounding at the store
//
// Every operand of the forward GEMMs is K-contiguous in the token-major layout (hidden [tokens][D] . W_in[2E][D]^T, out_z [tokens][E] .
// W_out[D][E]^T -- MS:185-189, SSI:517); the data-gradient GEMMs (SSI:540; d hidden = dxz . W_in) take the same form with the weight's
// transpose, which the host caches once per step next to its 16-bit cast.  M = batch * len tokens (32 832 for the bench), N and K are
// the model's widths (768 / 1536 / 3072): N % 256 == 0 and K % 64 == 0 are required, M is arbitrary.
//
// Division of the work
//   * one workgroup = 8 waves = one 256 x 256 tile of C, K walked in steps of 64; wave (wr, wc) of the 2 x 4 grid owns 128 rows x 64
//     columns = 8 x 4 accumulator fragments of v_mfma_f32_16x16x32_bf16 (128 accumulator registers).
//   * operands go HBM -> LDS by buffer_load_dwordx4 ... lds (no registers, no ds_write pass): a wave instruction moves 8 rows x 128
//     bytes; a K-step of A plus B is 64 such pieces, 8 per wave.  Two LDS buffers of 64 KB: the pieces of step t + 1 are in flight
//     while step t computes; one workgroup barrier per step.  Rows of A beyond M are outside the buffer descriptor's range (the
//     fetch never leaves the tensor; the rows they would produce are not stored).
//   * the LDS image is lane-linear (the DMA writes lane l's 16 bytes at base + 16 l), so the bank swizzle lives in the SOURCE address:
//     the lane that fills 16-byte slot s of row r fetches k-slot s ^ f(r), and the fragment reads apply the same XOR.
//     ds_read_b128 serves 16 lanes per LDS cycle over a 256-byte bank row (two 128-byte tile rows): A fragments read rows r0 .. r0+15,
//     f_A(r) = (r >> 1) & 7; B fragments read rows {8 q + e + 32 (j >> 1) + 4 (j & 1)} (below), f_B(r) = ((r >> 3) & 3) << 1 | ((r >> 1) & 1): both give
//     16 distinct slots per lane group (checked exhaustively on the host: tests/test_gemm_layout.py).
//   * MFMA roles are swapped -- the weight fragment is the MFMA's A operand, the activation fragment its B operand -- so a lane's four
//     accumulator values of a fragment are four consecutive COLUMNS of C, and fragment j of a wave reads weight rows
//     {8 q + e + 32 (j >> 1) + 4 (j & 1) : q, e = 0..3}: fragments 0, 1 of lane group kg are columns 8 kg .. 8 kg + 7 and fragments 2, 3
//     columns 32 + 8 kg ..: two 16-byte stores per lane and fragment row, each store instruction writing 64 contiguous bytes per row of C
//     (the first layout -- 16 consecutive columns per lane, 16-byte pieces 32 bytes apart per instruction -- cost 7 us of a 26 us tile).
//   * launch: blockIdx -> tile with the 8 XCDs each taking a contiguous range of tiles (tiles of one row block of A next to each other:
//     A is fetched from HBM once per XCD-resident row block, B (the weight, <= 4.7 MB) stays in L2 / MALL).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_args.h"

namespace aumg {

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

#ifndef AUM_GEMM_PRIO
#define AUM_GEMM_PRIO 0     // A/B builds: 1 the second-dispatched half of the waves (4-7, the younger wave of every SIMD) holds priority 1; 2 the two waves of a SIMD take turns K-step by K-step
#endif
#ifndef AUM_GEMM_ABL
#define AUM_GEMM_ABL 0      // timing experiments on the persistent kernel only (wrong results; tools/gemm_abl_probe.py): 1 no tile-end stores, 2 two dummy
#endif                      // stores per wave in each of a tile's first eight K-steps (the paced-store pattern), 4 no DMA pieces inside the K loop, 8 no reads / MFMAs
constexpr int NWAVES = 8, THREADS = NWAVES * 64;
constexpr int TILE_BYTES = BM * BK * 2;                 // one operand, one K-step: 32 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;             // A | B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;              // two K-steps: 128 KB of the CU's 160 KB

template <bool BF16> __device__ __forceinline__ f4v mfma(s8v a, s8v b, f4v c) {
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <bool BF16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, bf2v));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, h2v));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// the 8 pieces (4 of A, 4 of B) this wave contributes to one K-step: piece c = j * 8 + w is rows 8 c .. 8 c + 7 of the tile
__device__ __forceinline__ void stage(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voff_a, int voff_b, int kbyte, int rowstep_a,
                                      int rowstep_b, char* lds_stage, int w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds_stage + (j * 8 + w) * 1024), 16, voff_a, kbyte + j * rowstep_a, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds_stage + TILE_BYTES + (j * 8 + w) * 1024), 16, voff_b,
                                                 kbyte + j * rowstep_b, 0, 0);
    }
}

// weight fragment j of a wave reads tile rows {8 q + e + 32 (j >> 1) + 4 (j & 1) : q, e = 0..3} of the wave's 64
__device__ __forceinline__ constexpr int b_joff(int j) { return ((j >> 1) * 32 + (j & 1) * 4) * 128; }
__device__ __forceinline__ s8v lds_frag(const char* lds, int byte_off) { return *reinterpret_cast<const s8v*>(lds + byte_off); }

// blockIdx -> tile id with each XCD (blockIdx % 8) working through a contiguous range of ids (bijective for any grid size)
__device__ __forceinline__ int xcd_tile(int orig, int nwg) {
    const int xcd = orig & 7, idx = orig >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// SCHED 0: every wave stages, reads and multiplies K-step by K-step in lockstep, one barrier per step (the first version; kept for A/B:
//          AUM_GEMM_LOCKSTEP).
// SCHED 1: the two waves of a SIMD (wave w and w + 4 = the row halves wr = 0 / 1 of the tile) run half a K-step apart.  A phase = half a
//          K-step (32 of the 64 k) = a LOAD segment (12 fragment reads, and on one phase per step the wave's 8 DMA pieces of the next
//          step) and an MFMA segment (32 MFMAs at raised priority), a workgroup barrier after each; wr = 1 starts one barrier late, so
//          between two barriers one wave of every SIMD feeds the matrix pipe while the other fetches.  With barriers B_0 .. B_2P
//          (P = 2 nk phases) and segment s = the code between B_{s-1} and B_s:
//              wr = 0:  load(p) in segment 2p,     MFMA(p) in segment 2p + 1        wr = 1:  load(p) in 2p + 1,  MFMA(p) in 2p + 2
//          K-step t + 1 goes into the buffer step t - 1 was read from.  Its last reads (phase 2t - 1) have retired before B_{4t-1}
//          (wr = 0: lgkmcnt(0) at the head of MFMA(2t-1), segment 4t - 1) and before B_{4t} (wr = 1, segment 4t), so pieces may be issued
//          from segment 4t + 1 on; its first reads (phase 2t + 2) are in segment 4t + 4 (wr = 0), so the pieces must have landed before
//          B_{4t+3}.  wr = 1 issues in segment 4t + 1 (its load(2t)) and waits at the end of segment 4t + 3 (its load(2t+1)); wr = 0 issues
//          in segment 4t + 2 (its load(2t+1)) and waits at the end of segment 4t + 3 (its MFMA(2t+1)).
template <bool BF16, int SCHED>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn(AumGemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    if (AUM_GEMM_PRIO == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    const int ntn = g.n / BN;
    const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int rows_a = g.m - m0 < BM ? g.m - m0 : BM;

    const char* a_base = static_cast<const char*>(g.a) + (int64_t)m0 * g.lda * 2;
    const char* b_base = static_cast<const char*>(g.b) + (int64_t)n0 * g.ldb * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a_base), 0, rows_a * g.lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b_base), 0, BN * g.ldb * 2, 0x00020000);

    // staging: this lane fills slot (lane & 7) of row 8 c + (lane >> 3); f_A / f_B of that row do not depend on j
    const int srow = w * 8 + (lane >> 3);
    const int f_a = ((w & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = ((w & 3) << 1) | ((lane >> 4) & 1);
    const int voff_a = srow * g.lda * 2 + (((lane & 7) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 7) ^ f_b) << 4);
    const int rowstep_a = 64 * g.lda * 2, rowstep_b = 64 * g.ldb * 2;

    // fragment reads: lane = (operand row rho = lane & 15, k-group kg = lane >> 4)
    const int rho = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + rho) * 128 + ((kg ^ ((lane >> 1) & 7)) << 4);                                  // + i * 2048, ^ 64 for the second half of K
    const int b_row = wc * 64 + (rho >> 2) * 8 + (rho & 3);                                                     // + (j >> 1) * 32 + (j & 1) * 4
    const int b_rd = TILE_BYTES + b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);     // + b_joff(j), ^ 64

    f4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

    const int nk = g.k / BK;
    stage(ra, rb, voff_a, voff_b, 0, rowstep_a, rowstep_b, lds, w);
    if constexpr (SCHED == 0) {
        for (int t = 0; t < nk; ++t) {
            // step t has landed (this wave's pieces: vmcnt; everybody's: the barrier), and everybody is done reading step t - 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < nk) stage(ra, rb, voff_a, voff_b, (t + 1) * (BK * 2), rowstep_a, rowstep_b, lds + ((t + 1) & 1) * STAGE_BYTES, w);
            const char* st = lds + (t & 1) * STAGE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s8v bf[4], af[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = lds_frag(st, (b_rd ^ (kk * 64)) + b_joff(j));
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = lds_frag(st, (a_rd ^ (kk * 64)) + i * 2048);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[j], af[i], acc[i][j]);
            }
        }
    } else if constexpr (SCHED == 2) {
        // SCHED 2 (round 4, AUM_GEMM_PIPELINED; the default where a CU gets at most two tiles): same box, us, lockstep -> pipelined:
        //   N x K = 3072 x 768: 165.0 -> 156.5, 768 x 1536: 81.4 -> 74.6, 1536 x 768: 93.3 -> 87.4, 768 x 3072: 145.6 -> 134.7 (bitwise the same
        //   results).  The same loop inside the persistent kernel needs 243 registers and ran 2-18 % SLOWER than its 170-register loop: not kept.
        // The fragments of the NEXT half K-step are read while the current half's 32 MFMAs
        // run (two register sets), so the matrix pipe does not wait out a fragment read at the head of every half step; the barrier of
        // a K-step sits between its halves, where the next step's pieces are needed, and the pieces of step t + 2 are issued right behind it
        s8v bf[2][4], af[2][8];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // step 0 is in LDS
        if (1 < nk) stage(ra, rb, voff_a, voff_b, BK * 2, rowstep_a, rowstep_b, lds + STAGE_BYTES, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[0][j] = lds_frag(lds, b_rd + b_joff(j));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[0][i] = lds_frag(lds, a_rd + i * 2048);
        for (int t = 0; t < nk; ++t) {
            if (AUM_GEMM_PRIO == 2) { if ((t + (w >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            const char* st = lds + (t & 1) * STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[1][j] = lds_frag(st, (b_rd ^ 64) + b_joff(j));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[1][i] = lds_frag(st, (a_rd ^ 64) + i * 2048);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[0][j], af[0][i], acc[i][j]);
#pragma unroll
            for (int q = 0; q < 12; ++q) {                         // one fragment read per two MFMAs, the remaining eight MFMAs behind them
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (t + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of step t + 1 landed, its reads of step t returned
                __builtin_amdgcn_s_barrier();                                    // ... everybody's
                if (t + 2 < nk) stage(ra, rb, voff_a, voff_b, (t + 2) * (BK * 2), rowstep_a, rowstep_b, lds + (t & 1) * STAGE_BYTES, w);
                const char* sn = lds + ((t + 1) & 1) * STAGE_BYTES;
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[0][j] = lds_frag(sn, b_rd + b_joff(j));
#pragma unroll
                for (int i = 0; i < 8; ++i) af[0][i] = lds_frag(sn, a_rd + i * 2048);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[1][j], af[1][i], acc[i][j]);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // step 0 is in LDS
        if (wr == 1) __builtin_amdgcn_s_barrier();                 // B_0: the second wave of every SIMD runs one segment behind
        for (int t = 0; t < nk; ++t) {
            const char* st = lds + (t & 1) * STAGE_BYTES;
            const bool more = t + 1 < nk;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---- load segment
                s8v bf[4], af[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = lds_frag(st, (b_rd ^ (kk * 64)) + b_joff(j));
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = lds_frag(st, (a_rd ^ (kk * 64)) + i * 2048);
                if (more && kk == 1 - wr) stage(ra, rb, voff_a, voff_b, (t + 1) * (BK * 2), rowstep_a, rowstep_b, lds + ((t + 1) & 1) * STAGE_BYTES, w);
                if (more && kk == 1 && wr == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA segment
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[j], af[i], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                if (more && kk == 1 && wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();                 // B_2P
    }

    // store: lane holds, for fragment row i, columns wc * 64 + 32 (j >> 1) + 8 kg + 4 (j & 1) + r (j, r = 0..3) of row wr * 128 + 16 i + rho
    char* c_base = static_cast<char*>(g.c) + ((int64_t)m0 * g.ldc + n0 + wc * 64 + kg * 8) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wr * 128 + i * 16 + rho;
        if (row < rows_a) {
            u4v lo, hi;
            lo.x = pack2<BF16>(acc[i][0][0], acc[i][0][1]);
            lo.y = pack2<BF16>(acc[i][0][2], acc[i][0][3]);
            lo.z = pack2<BF16>(acc[i][1][0], acc[i][1][1]);
            lo.w = pack2<BF16>(acc[i][1][2], acc[i][1][3]);
            hi.x = pack2<BF16>(acc[i][2][0], acc[i][2][1]);
            hi.y = pack2<BF16>(acc[i][2][2], acc[i][2][3]);
            hi.z = pack2<BF16>(acc[i][3][0], acc[i][3][1]);
            hi.w = pack2<BF16>(acc[i][3][2], acc[i][3][3]);
            u4v* dst = reinterpret_cast<u4v*>(c_base + (int64_t)row * g.ldc * 2);
            dst[0] = lo;
            dst[4] = hi;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Persistent form (the default).  One workgroup per CU walks a list of work items round-robin (item = blockIdx + round * gridDim):
//   * full items: 256 x 256 tiles of the row blocks that are complete; in each round the 32 workgroups of an XCD take 32 consecutive
//     tiles (2.7 row blocks of A x the column tiles: A comes from HBM once per XCD, the weight stays in L2 / MALL);
//   * half items: the last m % 256 rows when they are at most 128 (the bench: 64 x 513 tokens = 128 row blocks + 64 rows) as 128 x 256
//     tiles -- the two row halves of the wave grid take 64 rows each (4 fragment rows instead of 8), a half with no rows left skips its
//     reads and MFMAs -- instead of a round of full-price tiles for a quarter of a tile's work (7 -> 6.3 rounds at N = 3072).
//   * the K loop is the lockstep one (SCHED 0 above); in its LAST step, where a tile has nothing left to fetch, the wave issues the
//     pieces of the NEXT item's first step, so the next tile's HBM round trip runs under this tile's stores, and the next tile's first
//     wait is counted so that those stores stay in flight.
// What bounds it (measured, profiles/r03_gemm_*): a K-step takes 1.54 us (1.39 PFLOP/s across the chip) whatever the shape, and every tile
// pays 5.7 us on top -- its 128 KB of C leave the CU at HBM's pace while every matrix pipe idles, in all 256 CUs at the same moment
// (tiles take equal time).  Tried: rounding a finished tile to 64 packed registers and storing it two stores per K-step under the next
// tile (exact counted waits): 256 VGPRs + spills, the compiler falls back to one fragment read per four MFMAs, 210 us instead of 159;
// deferring half of the tile (230 VGPRs, no spills): 162 us, the basic loop loses what the stores gain.  Not kept.
// Buffer parity: step t of a tile uses buffer (par + t) & 1; the next tile starts at par' = (par + nk) & 1, the buffer the last-but-one
// step was read from (free since the barrier of the last step).
// ------------------------------------------------------------------------------------------------------------------------------------
struct GemmLaunch {
    AumGemmArgs g;
    int full_rb;        // complete 256-row blocks handled as full items
    int half_rb;        // 128-row blocks behind them (the last one may be ragged)
    int nitems;
};

struct GemmItem {
    int m0, n0, rows;
    bool half;
};
__device__ __forceinline__ GemmItem gemm_item(const GemmLaunch& L, int id, int ntn, int grid) {
    GemmItem it;
    const int nfull = L.full_rb * ntn;
    if (id < nfull) {
        int tile = id;
        const int r0 = id / grid * grid;
        if ((grid & 7) == 0 && r0 + grid <= nfull) {            // a complete round: XCD x (= workgroup % 8) takes tiles r0 + x * grid/8 ...
            const int q = id - r0;
            tile = r0 + (q & 7) * (grid >> 3) + (q >> 3);
        }
        const int tm = tile / ntn;
        it.m0 = tm * BM;
        it.n0 = (tile - tm * ntn) * BN;
        it.rows = L.g.m - it.m0 < BM ? L.g.m - it.m0 : BM;       // < BM only for a last row block of 129 .. 255 rows (taken as a full item)
        it.half = false;
    } else {
        const int h = id - nfull, hb = h / ntn;
        it.m0 = L.full_rb * BM + hb * 128;
        it.n0 = (h - hb * ntn) * BN;
        it.rows = L.g.m - it.m0 < 128 ? L.g.m - it.m0 : 128;
        it.half = true;
    }
    return it;
}

// A tile's K-steps.  stores16: the previous item of this workgroup was a full tile, i.e. behind the pieces of this tile's first step there
// are exactly that tile's 16 stores per wave (memory operations retire in issue order) -- the first wait leaves them in flight.
// (The body is written out here rather than in a helper: the same statements behind a function boundary compile to a 220-register
// schedule that runs the persistent kernel 10 % slower -- check .vgpr_count = 170 and the step A/B after touching this.)
template <bool BF16, int NI>
__device__ __forceinline__ void gemm_steps(f4v (&acc)[8][4], int nk, int par, bool active, char* lds, int a_rd, int b_rd, __amdgpu_buffer_rsrc_t ra,
                                           __amdgpu_buffer_rsrc_t rb, bool has_next, __amdgpu_buffer_rsrc_t ra_n, __amdgpu_buffer_rsrc_t rb_n,
                                           int voff_a, int voff_b, int rowstep_a, int rowstep_b, int w, bool stores16, uint32_t L_flags,
                                           char* abl_c = nullptr, int64_t abl_ldc2 = 0, int abl_row0 = 0, int abl_rows = 0) {
    for (int t = 0; t < nk; ++t) {
        if (AUM_GEMM_PRIO == 2) { if ((t + (w >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (t == 0 && stores16 && !(L_flags & (AUM_GEMM_NO_COUNTED_WAIT | AUM_GEMM_NO_PREFETCH))) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if ((AUM_GEMM_ABL & 2) && t >= 1 && t <= NI) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the previous step's two stores stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (AUM_GEMM_ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's ds_write of the step's pieces
        __builtin_amdgcn_s_barrier();
        char* nxt = lds + ((par + t + 1) & 1) * STAGE_BYTES;
        u4v rg_a[4], rg_b[4];                        // AUM_GEMM_ABL & 16: the next step's pieces through registers (buffer_load -> ds_write_b128) instead of LDS-DMA
        if (AUM_GEMM_ABL & 16) {
            const bool nx = t + 1 < nk;
            const __amdgpu_buffer_rsrc_t qa = nx ? ra : ra_n, qb = nx ? rb : rb_n;
            const int kb = nx ? (t + 1) * (BK * 2) : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rg_a[j] = __builtin_amdgcn_raw_buffer_load_b128(qa, voff_a, kb + j * rowstep_a, 0);
                rg_b[j] = __builtin_amdgcn_raw_buffer_load_b128(qb, voff_b, kb + j * rowstep_b, 0);
            }
        } else if (AUM_GEMM_ABL & 4) {
        } else if (t + 1 < nk) stage(ra, rb, voff_a, voff_b, (t + 1) * (BK * 2), rowstep_a, rowstep_b, nxt, w);
        else if (has_next && !(L_flags & AUM_GEMM_NO_PREFETCH)) stage(ra_n, rb_n, voff_a, voff_b, 0, rowstep_a, rowstep_b, nxt, w);
        const char* st = lds + ((par + t) & 1) * STAGE_BYTES;
        if ((AUM_GEMM_ABL & 2) && t < NI) {           // the paced-store pattern with whatever two registers hold: row block t of the tile, both halves
            const int row = abl_row0 + t * 16;
            if (row < abl_rows) {
                u4v* dst = reinterpret_cast<u4v*>(abl_c + (int64_t)row * abl_ldc2);
                dst[0] = __builtin_bit_cast(u4v, acc[0][0]);
                dst[4] = __builtin_bit_cast(u4v, acc[0][1]);
            }
        }
        if (active && !(AUM_GEMM_ABL & 8)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s8v bf[4], af[NI];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = lds_frag(st, (b_rd ^ (kk * 64)) + b_joff(j));
#pragma unroll
                for (int i = 0; i < NI; ++i) af[i] = lds_frag(st, (a_rd ^ (kk * 64)) + i * 2048);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[j], af[i], acc[i][j]);
            }
        }
        if (AUM_GEMM_ABL & 16) {
            const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<u4v*>(nxt + (j * 8 + w) * 1024 + lane * 16) = rg_a[j];
                *reinterpret_cast<u4v*>(nxt + TILE_BYTES + (j * 8 + w) * 1024 + lane * 16) = rg_b[j];
            }
        }
    }
}
template <bool BF16, int NI>
__device__ __forceinline__ void gemm_store(const f4v (&acc)[8][4], char* c_rows, int64_t ldc2, int row0, int rows) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = row0 + i * 16;
        if (row < rows) {
            u4v lo, hi;
            lo.x = pack2<BF16>(acc[i][0][0], acc[i][0][1]);
            lo.y = pack2<BF16>(acc[i][0][2], acc[i][0][3]);
            lo.z = pack2<BF16>(acc[i][1][0], acc[i][1][1]);
            lo.w = pack2<BF16>(acc[i][1][2], acc[i][1][3]);
            hi.x = pack2<BF16>(acc[i][2][0], acc[i][2][1]);
            hi.y = pack2<BF16>(acc[i][2][2], acc[i][2][3]);
            hi.z = pack2<BF16>(acc[i][3][0], acc[i][3][1]);
            hi.w = pack2<BF16>(acc[i][3][2], acc[i][3][3]);
            u4v* dst = reinterpret_cast<u4v*>(c_rows + (int64_t)row * ldc2);
            dst[0] = lo;
            dst[4] = hi;
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn_persistent(GemmLaunch L) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    if (AUM_GEMM_PRIO == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    const int ntn = g.n / BN, grid = (int)gridDim.x, nk = g.k / BK;

    const int srow = w * 8 + (lane >> 3);
    const int f_a = ((w & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = ((w & 3) << 1) | ((lane >> 4) & 1);
    const int voff_a = srow * g.lda * 2 + (((lane & 7) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 7) ^ f_b) << 4);
    const int rowstep_a = 64 * g.lda * 2, rowstep_b = 64 * g.ldb * 2;
    const int rho = lane & 15, kg = lane >> 4;
    const int a_swz = (kg ^ ((lane >> 1) & 7)) << 4;
    const int b_row = wc * 64 + (rho >> 2) * 8 + (rho & 3);
    const int b_rd = TILE_BYTES + b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);

    auto rsrc_a = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.a) + (int64_t)it.m0 * g.lda * 2), 0,
                                                 it.rows * g.lda * 2, 0x00020000);
    };
    auto rsrc_b = [&](const GemmItem& it) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.b) + (int64_t)it.n0 * g.ldb * 2), 0,
                                                 BN * g.ldb * 2, 0x00020000);
    };

    int id = (int)blockIdx.x;
    if (id >= L.nitems) return;
    GemmItem it = gemm_item(L, id, ntn, grid);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(it), rb = rsrc_b(it);
    int par = 0;
    bool stores16 = false;          // the previous item of this workgroup was a full tile
    stage(ra, rb, voff_a, voff_b, 0, rowstep_a, rowstep_b, lds, w);
    while (true) {
        const int nid = id + grid;
        const bool has_next = nid < L.nitems;
        GemmItem itn = it;
        if (has_next) itn = gemm_item(L, nid, ntn, grid);
        const __amdgpu_buffer_rsrc_t ra_n = rsrc_a(itn), rb_n = rsrc_b(itn);

        f4v acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
        const int row_base = it.half ? wr * 64 : wr * 128;                // this wave's first row of the tile
        const int a_rd = (row_base + rho) * 128 + a_swz;
        char* c_rows = static_cast<char*>(g.c) + ((int64_t)it.m0 * g.ldc + it.n0 + wc * 64 + kg * 8) * 2;
        if (!it.half) {
            gemm_steps<BF16, 8>(acc, nk, par, true, lds, a_rd, b_rd, ra, rb, has_next, ra_n, rb_n, voff_a, voff_b, rowstep_a, rowstep_b, w,
                                (AUM_GEMM_ABL & 1) ? false : stores16, g.flags, c_rows, (int64_t)g.ldc * 2, row_base + rho, it.rows);
            if (!(AUM_GEMM_ABL & 1)) gemm_store<BF16, 8>(acc, c_rows, (int64_t)g.ldc * 2, row_base + rho, it.rows);
            else {          // the accumulators stay live (no code): without a consumer the compiler deletes the MFMAs and their reads
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
            }
        } else {
            gemm_steps<BF16, 4>(acc, nk, par, row_base < it.rows, lds, a_rd, b_rd, ra, rb, has_next, ra_n, rb_n, voff_a, voff_b, rowstep_a,
                                rowstep_b, w, stores16, g.flags);
            gemm_store<BF16, 4>(acc, c_rows, (int64_t)g.ldc * 2, row_base + rho, it.rows);
        }
        if (!has_next) break;
        stores16 = !it.half && it.rows == BM;         // every wave of a complete full tile issued its 16 stores
        id = nid;
        it = itn;
        ra = ra_n;
        rb = rb_n;
        par = (par + nk) & 1;
        if (g.flags & AUM_GEMM_NO_PREFETCH) stage(ra, rb, voff_a, voff_b, 0, rowstep_a, rowstep_b, lds + par * STAGE_BYTES, w);      // A/B: fetch at the tile's head
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Split tail (round 4; aum_gemm_tn_sk).  N = 768 at 64 x 513 tokens is 387 tiles on 256 CUs: a second round in which half of the CUs idle
// (75 % of the chip over the launch), and the reason the two N = 768 GEMMs stayed with the library.  Here the complete rounds run as whole
// tiles (one workgroup per CU, tile = blockIdx + round * grid) and the remaining `tail` tiles are split ALONG K between the workgroups:
// the tail's K-steps (tail tiles x K / 64) are dealt out in equal contiguous ranges -- XCD x takes tiles [x T / 8, (x + 1) T / 8), its 32
// workgroups equal shares of their K-steps -- so a workgroup computes the end of one tile and the beginning of the next.  The workgroup that
// holds a tile's FIRST K-step finishes the tile: every workgroup walks its range upwards, so that part is the last thing it computes, while
// the other contributors (one or two: ranges are at least a third of a tile) computed theirs first, wrote fp32 partial tiles to the workspace
// (write-through stores: visible to every XCD) and raised a flag.  The finisher adds the partials to its accumulators and stores the tile.
// Flags carry the launch's epoch (the host counts launches per workspace): nothing is cleared between launches.  No workgroup waits for
// one that waits (contributors never wait), every workgroup is resident (grid <= CUs, one per CU), the wait is bounded all the same.
// MEASURED (profiles/r04_gemm_split_tail.txt), parity-green and NOT faster: N = 768, K = 1536 / 3072: 113 / 187 us against 87 / 160 us for
// whole tiles.  Ablations: the complete round alone 50 / 104 us; + the finishers' half tiles and their stores 88 / 152 us; + the hand-over
// 113 / 187 us -- writing 256 KB per contributor, the flag, and reading it back is a serial chain of ~25 us behind half a tile's MFMAs, more
// than the idle half round it replaces.  Kept as an opt-in entry point (aum_gemm_tn_sk) with its tests; the default stays whole tiles.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int SK_SLOTS = 2;
#ifndef AUM_SK_ABL
#define AUM_SK_ABL 0        // timing experiments only (wrong results): 1 no partial exchange, 2 no tail at all, 3 no stores of finished tail tiles either
#endif
#ifndef AUM_SK_AUX
#define AUM_SK_AUX 16       // cache policy of the partial-tile exchange: 16 = sc1 (agent scope: through the L2 to the memory side), 17 = system scope, 0 = plain
#endif
struct GemmSkLaunch {
    AumGemmArgs g;
    float* part;           // [tail][SK_SLOTS][BM * BN] fp32
    uint32_t* flags;       // [tail][SK_SLOTS]
    uint32_t* err;         // set to 1 when a wait ran out (the result is then incomplete)
    uint32_t epoch;
    int rounds;            // complete rounds of whole tiles: tiles [0, rounds * grid)
    int tail;              // tiles after them
};

template <bool BF16>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn_sk(GemmSkLaunch L) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const AumGemmArgs& g = L.g;
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    const int ntn = g.n / BN, grid = (int)gridDim.x, nk = g.k / BK;
    const int srow = w * 8 + (lane >> 3);
    const int f_a = ((w & 1) * 4 + (lane >> 4)) & 7;
    const int f_b = ((w & 3) << 1) | ((lane >> 4) & 1);
    const int voff_a = srow * g.lda * 2 + (((lane & 7) ^ f_a) << 4);
    const int voff_b = srow * g.ldb * 2 + (((lane & 7) ^ f_b) << 4);
    const int rowstep_a = 64 * g.lda * 2, rowstep_b = 64 * g.ldb * 2;
    const int rho = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + rho) * 128 + ((kg ^ ((lane >> 1) & 7)) << 4);
    const int b_row = wc * 64 + (rho >> 2) * 8 + (rho & 3);
    const int b_rd = TILE_BYTES + b_row * 128 + ((kg ^ ((((rho >> 2) & 3) << 1) | ((rho >> 1) & 1))) << 4);
    int par = 0;

    f4v acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
    };
    // K-steps [k0, k1) of tile `tile` into acc
    auto run = [&](int tile, int k0, int k1) {
        const int tm = tile / ntn, m0 = tm * BM, n0 = (tile - tm * ntn) * BN;
        const int rows = g.m - m0 < BM ? g.m - m0 : BM;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.a) + (int64_t)m0 * g.lda * 2), 0,
                                                                            rows * g.lda * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(g.b) + (int64_t)n0 * g.ldb * 2), 0,
                                                                            BN * g.ldb * 2, 0x00020000);
        // (the buffer written here was last read two steps ago: every wave has passed the barrier of the step in between)
        stage(ra, rb, voff_a, voff_b, k0 * (BK * 2), rowstep_a, rowstep_b, lds + (par & 1) * STAGE_BYTES, w);
        for (int t = k0; t < k1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < k1) stage(ra, rb, voff_a, voff_b, (t + 1) * (BK * 2), rowstep_a, rowstep_b, lds + ((par + 1) & 1) * STAGE_BYTES, w);
            const char* st = lds + (par & 1) * STAGE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s8v bf[4], af[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = lds_frag(st, (b_rd ^ (kk * 64)) + b_joff(j));
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = lds_frag(st, (a_rd ^ (kk * 64)) + i * 2048);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma<BF16>(bf[j], af[i], acc[i][j]);
            }
            par ^= 1;
        }
    };
    auto store_tile = [&](int tile) {
        const int tm = tile / ntn, m0 = tm * BM, n0 = (tile - tm * ntn) * BN;
        const int rows = g.m - m0 < BM ? g.m - m0 : BM;
        char* c_rows = static_cast<char*>(g.c) + ((int64_t)m0 * g.ldc + n0 + wc * 64 + kg * 8) * 2;
        gemm_store<BF16, 8>(acc, c_rows, (int64_t)g.ldc * 2, wr * 128 + rho, rows);
    };
    // this lane's piece of an fp32 partial tile: row wr * 128 + 16 i + rho, columns wc * 64 + 32 (j >> 1) + 8 kg + 4 (j & 1) .. + 3
    const int p_lane = ((wr * 128 + rho) * BN + wc * 64 + kg * 8) * 4;
    auto part_rsrc = [&](int tt, int slot) {
        return __builtin_amdgcn_make_buffer_rsrc(L.part + ((int64_t)tt * SK_SLOTS + slot) * (BM * BN), 0, BM * BN * 4, 0x00020000);
    };

    // ---- complete rounds: whole tiles; the 32 workgroups of an XCD take 32 consecutive tiles of a round (A rows shared in its L2) ----
    for (int r = 0; r < L.rounds; ++r) {
        const int tile = r * grid + ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3);
        zero_acc();
        run(tile, 0, nk);
        store_tile(tile);
    }
    if (L.tail <= 0 || AUM_SK_ABL == 2) return;
    // ---- the tail, split along K ------------------------------------------------------------------------------------------------
    const int x = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3, cpx = grid >> 3;
    const int tx0 = x * L.tail / 8, tx1 = (x + 1) * L.tail / 8;
    const int S = (tx1 - tx0) * nk;                     // K-steps of this XCD's tail tiles
    const int q = (S + cpx - 1) / cpx;                  // per workgroup
    int pos = c * q;
    const int end = pos + q < S ? pos + q : S;
    const int tile0 = L.rounds * grid;
    while (pos < end) {
        const int tl = pos / nk, k0 = pos - tl * nk;
        const int k1 = k0 + (end - pos) < nk ? k0 + (end - pos) : nk;
        const int tt = tx0 + tl, tile = tile0 + tt;
        zero_acc();
        run(tile, k0, k1);
        if (k0 == 0) {
            // finisher: the other contributors are the workgroups c + 1 .. c_last of this XCD
            const int c_last = ((tl + 1) * nk - 1) / q;
            const int ncontrib = c_last - c < SK_SLOTS ? c_last - c : SK_SLOTS;
            for (int sl = 0; sl < (AUM_SK_ABL ? 0 : ncontrib); ++sl) {
                if (threadIdx.x == 0) {
                    const uint32_t* f = L.flags + (int64_t)tt * SK_SLOTS + sl;
                    int it = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != L.epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++it > (1 << 22)) {          // seconds: something is wrong -- report it instead of hanging the queue
                            __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();
                const __amdgpu_buffer_rsrc_t rp = part_rsrc(tt, sl);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u4v v = __builtin_amdgcn_raw_buffer_load_b128(rp, p_lane + (i * 16 * BN + (j >> 1) * 32 + (j & 1) * 4) * 4, 0, AUM_SK_AUX);
                        acc[i][j] = acc[i][j] + __builtin_bit_cast(f4v, v);
                    }
            }
            if (AUM_SK_ABL != 3) store_tile(tile);
        } else {
            // contributor: slot = position among the workgroups behind the finisher
            const int c_first = (tl * nk) / q;
            const int sl = c - c_first - 1;
            if (AUM_SK_ABL) {
            } else if (sl >= 0 && sl < SK_SLOTS) {
                const __amdgpu_buffer_rsrc_t rp = part_rsrc(tt, sl);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, acc[i][j]), rp, p_lane + (i * 16 * BN + (j >> 1) * 32 + (j & 1) * 4) * 4, 0, AUM_SK_AUX);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (threadIdx.x == 0) __hip_atomic_store(L.flags + (int64_t)tt * SK_SLOTS + sl, L.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (threadIdx.x == 0) {
                __hip_atomic_store(L.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // more contributors than slots: the host's split rule was violated
            }
        }
        pos += k1 - k0;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM (round 4; ABI 10, aum_gemm_wgrad):   C[n][k] = sum over tokens t of Y[t][n] * X[t][k]
// -- dW_in = dxz^T . hidden (autograd of MS:185-189), dW_out = dout^T . out_z (SSI:563).  Both operands are token-major, i.e. the
// contraction index is the ROW index of both stored matrices: the MFMA wants 8 consecutive t per lane, memory has them a row pitch apart.
//   * tiles go HBM -> LDS as they are stored ([64 tokens][256 columns] per operand and K-step, 16-byte chunks by buffer_load ... lds: a wave
//     instruction is two token rows of 512 bytes), and the fragments are read with ds_read_b64_tr_b16: inside a 16-lane group lane
//     4 j + q hands in the address of 4 consecutive columns (4 q ..) of token j, and gets back, as lane i, column i of the four tokens --
//     the 4 x 16 block transposed.  Two reads (tokens 8 g .. + 3, + 4 .. + 7 of lane group g) are one 16 x 32 operand fragment.
//   * bank conflicts: a token row is 512 bytes, so the eight tokens two lane groups read at once would all sit on the same bank columns;
//     (a transposing read is served 32 lanes at a time: tokens j and 8 + j, j = 0..3).  The 16-byte chunk c of token t is stored at chunk
//     c ^ 2 ((t & 3) + 4 ((t >> 3) & 1)) (the swizzle is in the DMA's per-lane SOURCE column, the image stays lane-linear) and the reads apply
//     the same XOR: the 32 lanes cover the 64 banks once (first version, XOR with 2 (t & 7): tokens j and 8 + j collided, SQ_LDS_BANK_CONFLICT
//     50 %).  Per-lane address registers for the wave's 8 + 4 fragment columns, token and stage offsets as instruction immediates (Y stages
//     at 0 / 32 KB, X stages at 64 / 96 KB: every immediate below 64 KB).
//   * MFMA roles: X fragment = A operand (rows = k), Y fragment = B operand (columns = n): a lane's four accumulator values are four
//     consecutive k of one n -- 16 contiguous bytes of the fp32 result row.
//   * K = 32 832 tokens against 36 / 18 output tiles: the token range is split over `splits` workgroups per tile (7 / 14: 252 workgroups),
//     each writes its fp32 partial tile, the caller sums them in a fixed order (aum_sum_rows): no atomics, bitwise repeatable.
// ------------------------------------------------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v lds_s4v;
constexpr int W_YOFF = 0, W_XOFF = 2 * TILE_BYTES, W_STAGE = 32 * 512, W_NST = 4;          // four 16 KB stages of Y from 0, of X from 64 KB

// One 1 KB piece global -> LDS, written as inline assembly: behind the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of every
// transposing read (it cannot tell which LDS-DMA the read depends on), i.e. waits for the pieces it has just requested for LATER stages;
// with the load opaque to it, the counted waits below are the only ones (first version: 2.2 us per 64-token step with the fragment reads
// removed altogether -- the loop ran at the memory latency).
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t r, const char* lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"((int)(size_t)(__attribute__((address_space(3))) const char*)lds_dst), "v"(voff), "s"(r), "s"(soff)
                 : "memory", "m0");
}

template <bool BF16>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_wgrad(AumGemmWArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
    if (AUM_GEMM_PRIO == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    const int ntn = g.n / 256, ntk = g.k / 256;
    // work item (split, tile) of this workgroup.  Every workgroup of a split streams the same token rows, so the items are numbered split-major
    // and each XCD (blockIdx % 8: its CUs share an L2) takes a contiguous run of them: its ~32 workgroups walk one or two token ranges in step and
    // fetch each row from HBM once per XCD instead of once per workgroup (numbered tile-major, the kernel ran at HBM's pace: 1.2 GB per launch).
    const int nitems = ntn * ntk * g.splits, per = (nitems + 7) / 8;
    const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || item >= nitems) return;
    const int s = item / (ntn * ntk);
    const int id = item - s * (ntn * ntk);
    const int tk = id % ntk, tn = id / ntk;
    const int n0 = tn * 256, k0 = tk * 256;
    const int64_t chunk = ((((int64_t)g.t + g.splits - 1) / g.splits) + 63) / 64 * 64;      // tokens per split, a multiple of the K-step
    const int64_t t0 = (int64_t)s * chunk;
    const int rows = (int)(t0 >= g.t ? 0 : (g.t - t0 < chunk ? g.t - t0 : chunk));
    const int nk = (rows + 31) / 32;            // K-steps of 32 tokens

    const char* y_base = static_cast<const char*>(g.y) + (t0 * g.ldy + n0) * 2;
    const char* x_base = static_cast<const char*>(g.x) + (t0 * g.ldx + k0) * 2;
    // rows beyond the split's range are outside the descriptor: they read as zero and add nothing
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(y_base), 0, rows > 0 ? (int)((int64_t)(rows - 1) * g.ldy * 2 + 512) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x_base), 0, rows > 0 ? (int)((int64_t)(rows - 1) * g.ldx * 2 + 512) : 0, 0x00020000);

    // staging: piece c = jj * 8 + w is tokens 2 c, 2 c + 1 of the K-step; lane l fills physical chunk l & 31 of token 2 c + (l >> 5) with
    // logical chunk (l & 31) ^ swizzle(token)
    const int trow = 2 * w + (lane >> 5);                          // + 16 jj: token inside the K-step
    const int csrc = (lane & 31) ^ (2 * ((trow & 3) + 4 * ((trow >> 3) & 1)));
    const int voff_y = trow * (int)g.ldy * 2 + csrc * 16;
    const int voff_x = trow * (int)g.ldx * 2 + csrc * 16;
    const int step_y = 16 * (int)g.ldy * 2, step_x = 16 * (int)g.ldx * 2;           // jj -> jj + 1; a K-step is two of them
    // a stage = 32 tokens of both operands (16 + 16 KB: two Y and two X pieces per wave), four stages in a ring: the pieces of stage t + 3
    // are requested when stage t is computed -- three steps (~3 us) of flight time.  (Two 64-token stages, fetched one step ahead, ran at
    // the memory latency: 2.2 us per step with the fragment reads removed altogether, against 1.5 us for the same bytes in aum_gemm_tn,
    // whose rows are mostly L2 hits; here every token row is a first touch for its XCD.)
    auto stage_w = [&](int kstep, int st) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            wg_dma16(ry, lds + W_YOFF + st * W_STAGE + (jj * 8 + w) * 1024, voff_y, (kstep * 2 + jj) * step_y);
            wg_dma16(rx, lds + W_XOFF + st * W_STAGE + (jj * 8 + w) * 1024, voff_x, (kstep * 2 + jj) * step_x);
        }
    };

    // fragment reads: lane = 16 gq + 4 j + q hands in columns 4 q .. 4 q + 3 (chunk q >> 1, half q & 1) of token 8 gq + j (+ 4 for the second
    // read, + 32 for the second half of the K-step).  Address of logical chunk c: token * 512 + ((c ^ 2 (token & 7)) * 16).
    const int gq = lane >> 4, j = (lane >> 2) & 3, q = lane & 3;
    const int swz = 2 * j + 8 * (gq & 1);          // 2 ((token & 3) + 4 ((token >> 3) & 1)): the same for both reads of a fragment and both halves of a K-step
    const int lane_lo = (8 * gq + j) * 512 + (q >> 1) * 16 + (q & 1) * 8;
    int ay[8], ax[4];
#pragma unroll
    for (int m = 0; m < 8; ++m) ay[m] = W_YOFF + lane_lo + (((16 * wr + 2 * m) ^ swz) * 16);
#pragma unroll
    for (int f = 0; f < 4; ++f) ax[f] = W_XOFF + lane_lo + (((8 * wc + 2 * f) ^ swz) * 16);

    f4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f4v{0.f, 0.f, 0.f, 0.f};

    // fragment loads of one phase = half a K-step's columns of Y: X fragments once per 32 tokens, Y fragments in two halves of four
#ifndef AUM_WGRAD_ABL
#define AUM_WGRAD_ABL 0        // timing experiments only (wrong results): 1 plain ds_read_b64, 2 one ds_read_b128 per fragment, 3 no fragment reads
#endif
    auto frag = [&](int addr) -> s8v {
#if AUM_WGRAD_ABL == 1
        const s4v lo = *reinterpret_cast<const s4v*>(lds + addr), hi = *reinterpret_cast<const s4v*>(lds + addr + 4 * 512);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#elif AUM_WGRAD_ABL == 2
        return *reinterpret_cast<const s8v*>(lds + (addr & ~15));
#elif AUM_WGRAD_ABL == 3
        s8v r;
        for (int i = 0; i < 8; ++i) r[i] = (short)(addr + i);
        return r;
#else
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(lds + addr));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(lds + addr + 4 * 512));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
    };
    auto load_x = [&](s8v (&xf)[4], int st, int ks) {
#pragma unroll
        for (int f = 0; f < 4; ++f) xf[f] = frag(ax[f] + st + ks * 32 * 512);
    };
    auto load_y = [&](s8v (&yf)[4], int st, int ks, int half) {
#pragma unroll
        for (int f = 0; f < 4; ++f) yf[f] = frag(ay[half * 4 + f] + st + ks * 32 * 512);
    };
    // (Round 4 also tried the software pipeline of aum_gemm_tn's SCHED 2 here -- next step's x / first-half y fragments read under the second
    // half's MFMAs, 80 fragment registers: 155.2 vs 154.1 us for dW_in, 91.2 vs 89.3 us for dW_out, the step 63.57 vs 63.54 ms, and 196 bytes
    // of spills at the joins of its tail.  Not the limit of this loop: not kept.)
    // Memory operations retire in issue order: with the four pieces of each of the next two stages behind them, the pieces of stage t have
    // landed when at most 8 are outstanding (fewer stages were requested near the end: wait for everything there).
#pragma unroll
    for (int p = 0; p < W_NST - 1; ++p)
        if (p < nk) stage_w(p, p);
    for (int t = 0; t < nk; ++t) {
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everybody's pieces of stage t are in LDS, and everybody is done reading stage t - 1
        if (t + W_NST - 1 < nk) stage_w(t + W_NST - 1, (t + W_NST - 1) & (W_NST - 1));
        const int st = (t & (W_NST - 1)) * W_STAGE;
        // two phases of 16 MFMAs; the second phase's Y fragments are requested before the first phase's MFMAs
        s8v xf[4], yf[2][4];
        load_x(xf, st, 0);
        load_y(yf[0], st, 0, 0);
        load_y(yf[1], st, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[half * 4 + fn][f] = mfma<BF16>(xf[f], yf[half][fn], acc[half * 4 + fn][f]);
    }


    // store the fp32 partial tile: lane (gq, i = lane & 15) holds C[n0 + wr * 128 + fn * 16 + i][k0 + wc * 64 + f * 16 + 4 gq + r], r = 0..3
    float* c_base = g.part + ((int64_t)s * g.n + n0 + wr * 128 + (lane & 15)) * g.k + k0 + wc * 64 + 4 * gq;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn)
#pragma unroll
        for (int f = 0; f < 4; ++f) *reinterpret_cast<f4v*>(c_base + (int64_t)fn * 16 * g.k + f * 16) = acc[fn][f];
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same product with a SKINNY second operand (round 4): k = 48 or 80 columns -- the weight gradients of dt_proj and x_proj,
//   d W_dt [E][48] = ddelta^T . x_dbl[:, :48]  (SSI:586),      d W_x^T [E][80] = conv_out^T . dx_dbl  (SSI:589, stored transposed),
// one 100 MB activation tensor streamed against 3 / 5 MB.  As strided batched library GEMMs they ran at a third of the copy rate
// (52 / 55 us for 16 us of traffic).  Arrangement: a workgroup owns a 256-channel slab of y and a token split; the y stages are the big
// kernel's ([32 tokens][256 columns], swizzled 512-byte rows), the x stage is [32 tokens][16 chunks] with the same swizzle in 256-byte rows
// (6 / 10 chunks are real, the DMA lanes of the others fetch chunk 0 again: 3 % of the traffic); a wave multiplies its 32 channels (two
// fragments) by all XC = 3 / 5 column fragments of x.  Three DMA pieces per wave and K-step, four stages in flight.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int W_XSTAGE_S = 32 * 256;          // bytes of one skinny x stage

template <bool BF16, int XC>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_wgrad_skinny(AumGemmWArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntn = g.n / 256;
    const int nitems = ntn * g.splits, per = (nitems + 7) / 8;         // split-major items, one contiguous run per XCD (see k_gemm_wgrad)
    const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || item >= nitems) return;
    const int s = item / ntn;
    const int n0 = (item - s * ntn) * 256;
    const int64_t chunk = ((((int64_t)g.t + g.splits - 1) / g.splits) + 63) / 64 * 64;
    const int64_t t0 = (int64_t)s * chunk;
    const int rows = (int)(t0 >= g.t ? 0 : (g.t - t0 < chunk ? g.t - t0 : chunk));
    const int nk = (rows + 31) / 32;

    const char* y_base = static_cast<const char*>(g.y) + (t0 * g.ldy + n0) * 2;
    const char* x_base = static_cast<const char*>(g.x) + (t0 * g.ldx) * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(y_base), 0, rows > 0 ? (int)((int64_t)(rows - 1) * g.ldy * 2 + 512) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x_base), 0, rows > 0 ? (int)((int64_t)(rows - 1) * g.ldx * 2 + XC * 32) : 0, 0x00020000);

    // y: piece c = jj * 8 + w is tokens 2 c, 2 c + 1 of the K-step (as in k_gemm_wgrad); x: piece w is tokens 4 w .. 4 w + 3, lane l fills
    // physical chunk l & 15 of token 4 w + (l >> 4) with logical chunk (l & 15) ^ swizzle(token), or with chunk 0 where that is past the row
    const int trow = 2 * w + (lane >> 5);
    const int csrc = (lane & 31) ^ (2 * ((trow & 3) + 4 * ((trow >> 3) & 1)));
    const int voff_y = trow * (int)g.ldy * 2 + csrc * 16;
    const int step_y = 16 * (int)g.ldy * 2;
    const int xrow = 4 * w + (lane >> 4);
    const int xc_l = (lane & 15) ^ (2 * ((xrow & 3) + 4 * ((xrow >> 3) & 1)));
    const int voff_x = xrow * (int)g.ldx * 2 + (xc_l < 2 * XC ? xc_l : 0) * 16;
    const int step_x = 32 * (int)g.ldx * 2;
    auto stage_w = [&](int kstep, int st) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) wg_dma16(ry, lds + W_YOFF + st * W_STAGE + (jj * 8 + w) * 1024, voff_y, (kstep * 2 + jj) * step_y);
        wg_dma16(rx, lds + W_XOFF + st * W_XSTAGE_S + w * 1024, voff_x, kstep * step_x);
    };

    const int gq = lane >> 4, j = (lane >> 2) & 3, q = lane & 3;
    const int swz = 2 * j + 8 * (gq & 1);
    const int lane_y = (8 * gq + j) * 512 + (q >> 1) * 16 + (q & 1) * 8;
    const int lane_x = (8 * gq + j) * 256 + (q >> 1) * 16 + (q & 1) * 8;
    int ay[2], ax[XC];
#pragma unroll
    for (int m = 0; m < 2; ++m) ay[m] = W_YOFF + lane_y + (((4 * w + 2 * m) ^ swz) * 16);
#pragma unroll
    for (int f = 0; f < XC; ++f) ax[f] = W_XOFF + lane_x + (((2 * f) ^ swz) * 16);

    f4v acc[2][XC];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int f = 0; f < XC; ++f) acc[m][f] = f4v{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](int addr, int tok4) -> s8v {
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(lds + addr));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(lds + addr + tok4));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int p = 0; p < W_NST - 1; ++p)
        if (p < nk) stage_w(p, p);
    for (int t = 0; t < nk; ++t) {
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + W_NST - 1 < nk) stage_w(t + W_NST - 1, (t + W_NST - 1) & (W_NST - 1));
        const int sy = (t & (W_NST - 1)) * W_STAGE, sx = (t & (W_NST - 1)) * W_XSTAGE_S;
        s8v xf[XC], yf[2];
#pragma unroll
        for (int f = 0; f < XC; ++f) xf[f] = frag(ax[f] + sx, 4 * 256);
#pragma unroll
        for (int m = 0; m < 2; ++m) yf[m] = frag(ay[m] + sy, 4 * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int f = 0; f < XC; ++f) acc[m][f] = mfma<BF16>(xf[f], yf[m], acc[m][f]);
    }
    // lane (gq, i = lane & 15) holds C[n0 + 32 w + 16 m + i][16 f + 4 gq + r], r = 0..3
    float* c_base = g.part + ((int64_t)s * g.n + n0 + 32 * w + (lane & 15)) * g.k + 4 * gq;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int f = 0; f < XC; ++f) *reinterpret_cast<f4v*>(c_base + (int64_t)m * 16 * g.k + f * 16) = acc[m][f];
}

}  // namespace aumg
