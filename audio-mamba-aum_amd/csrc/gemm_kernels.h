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

// SCHED 0: every wave stages, reads and multiplies K-step by K-step in lockstep, one barrier per step (the first version: AUM_GEMM_LOCKSTEP).
// SCHED 2: the fragments of the next half K-step are read under the current half's MFMAs (AUM_GEMM_PIPELINED, the default for products too
//          short for the paced-store kernel of gemm_ps_kernels.h).
// (Round 3 also had a staggered schedule -- the two waves of a SIMD half a K-step apart, 3-6 % slower -- and a persistent form of SCHED 0;
// round 5 a four-wave and a ring division; round 4 a split-K tail: all measured slower than what is left here, removed in round 6 -- HISTORY.md.)
template <bool BF16, int SCHED>
__global__ __launch_bounds__(THREADS, 1) void k_gemm_tn(AumGemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int lane = (int)(threadIdx.x & 63u);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 2, wc = w & 3;
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
    } else {
        static_assert(SCHED == 2, "schedules: 0 lockstep, 2 pipelined");
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


// work list of the paced-store kernel (gemm_ps_kernels.h).  Items [0, nwhole) are whole tiles: a 256-row block (the last one may be
// ragged) x one column tile.  When the tile count leaves a last round at most half full (n = 768 at 64 x 513 tokens: 384 + 3 tiles on 256
// CUs), the tiles of that round are SPLIT into two items of 128 rows each (items nwhole + 2 q, + 1 = the halves of tile nwhole + q): a
// half runs as a ragged tile -- its rows beyond 128 read as zero and are not stored, it moves 3/4 of a tile's bytes and runs at the
// matrix pipes' pace (0.65 of a whole tile) -- so the last round costs 0.65 instead of 1.  `fold` ragged rows (M mod 256, at most 64)
// then belong to the second half of the last 256-row block (128 + fold rows) instead of being a row block of their own.
struct GemmLaunch {
    AumGemmArgs g;
    int nitems;         // nwhole + 2 * (split tiles)
    int nwhole;
    int fold;
};

// the work list for (m, n) on `ncu` CUs (host side; gemm.hip).  Splitting applies when the whole tiles fill complete rounds and the rest
// fits, as halves, into one: 0 < rem, 2 rem <= ncu.
static inline void gemm_ps_items(int64_t m, int n, int ncu, GemmLaunch* L) {
    const int ntn = n / BN;
    const int full = (int)(m / BM), r = (int)(m % BM);
    const bool can_fold = r > 0 && r <= 64 && full >= 1;
    const int rb_all = full + (r > 0 ? 1 : 0);
    const int rb = can_fold ? full : rb_all;
    const int t = rb * ntn, rem = t % ncu;
    // a ragged row block of 65 .. 128 rows would leave its second half empty: such shapes are not split
#ifndef AUM_PS_SPLIT_TAIL
#define AUM_PS_SPLIT_TAIL 1     // 0: A/B build (tools/build_gemm_variant.sh nosplit -DAUM_PS_SPLIT_TAIL=0): whole tiles only
#endif
    const bool ok = AUM_PS_SPLIT_TAIL && t > ncu && rem > 0 && 2 * rem <= ncu && (can_fold ? rem >= ntn : (r == 0 || r > 128));
    if (ok) {
        L->nwhole = t - rem;
        L->nitems = t + rem;
        L->fold = can_fold ? r : 0;
    } else {
        L->nwhole = L->nitems = rb_all * ntn;
        L->fold = 0;
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
