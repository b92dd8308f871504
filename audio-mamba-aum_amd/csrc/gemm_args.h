// gemm_args.h -- argument rules of aum_gemm_tn (include/aum_hip.h, ABI 9), shared by the device library (gemm.hip) and the tests-only
// host build (tests/emu/aum_emu.cpp): no HIP dependency.
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

namespace aumg {
constexpr int BM = 256, BN = 256, BK = 64;          // C tile of a workgroup, K-step
inline int gemm_check(const AumGemmArgs* p) {
    if (!p || !p->a || !p->b || !p->c) return AUM_E_NULL;
    const AumGemmArgs& g = *p;
    if (g.m <= 0 || g.n <= 0 || g.k <= 0 || g.lda < g.k || g.ldb < g.k || g.ldc < g.n) return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    if (g.n % BN || g.k % BK || g.lda % 8 || g.ldb % 8 || g.ldc % 8) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.a | (uintptr_t)g.b | (uintptr_t)g.c) & 15u) return AUM_E_UNSUPPORTED;
    if ((int64_t)BM * g.lda * 2 >= (1ll << 31) || (int64_t)BN * g.ldb * 2 >= (1ll << 31)) return AUM_E_UNSUPPORTED;   // 32-bit buffer offsets
    return AUM_OK;
}
// aum_gemm_wgrad (ABI 10): both operands token-major, the result a stack of fp32 partial tiles
inline int gemm_wgrad_check(const AumGemmWArgs* p) {
    if (!p || !p->y || !p->x || !p->part) return AUM_E_NULL;
    const AumGemmWArgs& g = *p;
    if (g.t <= 0 || g.n <= 0 || g.k <= 0 || g.splits <= 0 || g.ldy < g.n || g.ldx < g.k) return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    // k: multiples of 256 (the projections), or 48 / 80 (the skinny operands x_dbl[:, :48] and dx_dbl: dt_proj's and x_proj's weight gradients)
    if (g.n % 256 || (g.k % 256 && g.k != 48 && g.k != 80) || g.ldy % 8 || g.ldx % 8 || g.splits > 64) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.y | (uintptr_t)g.x | (uintptr_t)g.part) & 15u) return AUM_E_UNSUPPORTED;
    const int64_t chunk = (((g.t + g.splits - 1) / g.splits) + 63) / 64 * 64;
    if (chunk * g.ldy * 2 >= (1ll << 31) || chunk * g.ldx * 2 >= (1ll << 31)) return AUM_E_UNSUPPORTED;   // 32-bit buffer offsets inside a split
    return AUM_OK;
}
}  // namespace aumg
