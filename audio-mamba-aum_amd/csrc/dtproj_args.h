// dtproj_args.h -- argument rules of aum_dtproj_tm_fwd (include/aum_hip.h, ABI 9), shared by the device library (gemm.hip) and the
// tests-only host build (tests/emu/aum_emu.cpp): no HIP dependency.
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

namespace aumd {
inline int dtproj_check(const AumDtProjArgs* p) {
    if (!p || !p->x || !p->w || !p->out) return AUM_E_NULL;
    const AumDtProjArgs& g = *p;
    if (g.ntok <= 0 || g.dim <= 0 || g.rank <= 0 || g.ldx < g.rank || g.ldw < g.rank || g.ldo < g.dim) return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    if (g.dim % 32 || g.rank % 8 || g.rank > 64 || g.ldx % 8 || g.ldw % 8 || g.ldo % 8) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.x | (uintptr_t)g.w | (uintptr_t)g.out) & 15u) return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
}  // namespace aumd
