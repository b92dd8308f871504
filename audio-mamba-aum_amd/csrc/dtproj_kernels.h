// dtproj_kernels.h -- the dt projection of the token-major block as a hand-written MFMA kernel (round 3; ABI 9 aum_dtproj_tm_fwd).
//
//   delta[M][E] (16-bit) = x_dbl[M][0:R] . W_dt[E][R]^T          SSI:468 (delta = dt_proj.weight @ x_dbl[:, :R].t(), without the transposes)
//
// K = R = dt_rank is tiny (48 for AuM-Base) and the result is the size of an activation tensor (100 MB at the bench shape): the
// product is bound by its WRITE.  The library's GEMM for this shape moves it at 1.4 TB/s (72 us); a wave here owns 32 tokens, keeps their
// x_dbl rows in registers as the MFMA's B operand for the whole kernel (4 fragments), streams W_dt through as the A operand (147 KB,
// L2-resident, 16 bytes per lane straight from memory: nothing is shared between waves, so nothing is staged through LDS), and stores
// 16 bytes per lane: the weight-row order inside a pair of 16-channel fragments is chosen so that a lane's 4 + 4 accumulator values are 8
// consecutive channels and each store instruction writes 64 contiguous bytes per token row (the layout of gemm_kernels.h).
// v_mfma_f32_16x16x32: rows = channels (A operand = W_dt fragment), columns = tokens (B operand = x_dbl fragment), K = 32 per instruction;
// R <= 32 is one K-step, R <= 64 two; k beyond R reads as zero on both sides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dtproj_args.h"

namespace aumd {

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <bool BF16> __device__ __forceinline__ f4v mfma(s8v a, s8v b, f4v c) {
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <bool BF16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, bf2v));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, h2v));
}

constexpr int TOK_PER_WAVE = 32, WAVES = 4;

template <bool BF16, int KS>       // KS = K-steps of 32: 1 (R <= 32) or 2 (R <= 64)
__global__ __launch_bounds__(WAVES * 64) void k_dtproj_tm(AumDtProjArgs g) {
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t wave = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t t0 = wave * TOK_PER_WAVE;
    if (t0 >= g.ntok) return;
    const int rho = lane & 15, kg = lane >> 4;
    const s8v zero = {0, 0, 0, 0, 0, 0, 0, 0};

    // the wave's tokens: B operand fragments [token fragment][K-step], held for the whole kernel
    s8v xf[2][KS];
    const char* xb = static_cast<const char*>(g.x);
#pragma unroll
    for (int tf = 0; tf < 2; ++tf)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int64_t t = t0 + tf * 16 + rho;
            const int k = ks * 32 + kg * 8;
            xf[tf][ks] = (t < g.ntok && k < g.rank) ? *reinterpret_cast<const s8v*>(xb + (t * g.ldx + k) * 2) : zero;
        }

    // fragment j of a channel pair reads weight rows c0 + 8 (rho >> 2) + 4 j + (rho & 3): the lane's accumulator rows 4 kg + r of
    // fragments 0 and 1 are then channels c0 + 8 kg + 0..7
    const char* wb = static_cast<const char*>(g.w) + ((int64_t)((rho >> 2) * 8 + (rho & 3)) * g.ldw + kg * 8) * 2;
    const bool k_ok[2] = {kg * 8 < g.rank, 32 + kg * 8 < g.rank};
    const int64_t wstep = (int64_t)32 * g.ldw * 2, wj = (int64_t)4 * g.ldw * 2;
    char* ob = static_cast<char*>(g.out) + ((t0 + rho) * g.ldo + kg * 8) * 2;
    const bool t_ok[2] = {t0 + rho < g.ntok, t0 + 16 + rho < g.ntok};
    const int64_t otf = (int64_t)16 * g.ldo * 2;

    const int npairs = g.dim / 32;
#pragma unroll 2
    for (int p = 0; p < npairs; ++p) {
        s8v wf[2][KS];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][ks] = k_ok[ks] ? *reinterpret_cast<const s8v*>(wb + p * wstep + j * wj + ks * 64) : zero;
#pragma unroll
        for (int tf = 0; tf < 2; ++tf) {
            f4v acc[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[j] = mfma<BF16>(wf[j][ks], xf[tf][ks], acc[j]);
            if (t_ok[tf]) {
                u4v o;
                o.x = pack2<BF16>(acc[0][0], acc[0][1]);
                o.y = pack2<BF16>(acc[0][2], acc[0][3]);
                o.z = pack2<BF16>(acc[1][0], acc[1][1]);
                o.w = pack2<BF16>(acc[1][2], acc[1][3]);
                *reinterpret_cast<u4v*>(ob + tf * otf + (int64_t)p * 64) = o;
            }
        }
    }
}

}  // namespace aumd
