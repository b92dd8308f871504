// xdt_args.h -- argument rules of aum_xdt_tm_fwd (include/aum_hip.h, ABI 9), shared by the device library (gemm.hip) and the tests-only
// host build (tests/emu/aum_emu.cpp): no HIP dependency.
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

#define XDT_COLS 80          // columns of x_dbl = dt_rank + 2 d_state the kernel is built for (AuM-Base: 48 + 32)
#ifndef XDT_TOK_W
#define XDT_TOK_W 16         // tokens per wave
#endif
#ifndef XDT_WAVES
#define XDT_WAVES 8
#endif
#define XDT_TOK_WG (XDT_TOK_W * XDT_WAVES)
#define XDT_MAX_DIM 1536     // W_x passes through LDS in two K-halves of XDT_COLS rows: 80 x (768 x 2 + 16) bytes

namespace aumx {
inline int xdt_check(const AumXdtArgs* p) {
    if (!p || !p->u || !p->wx || !p->wdt || !p->x_dbl || !p->delta) return AUM_E_NULL;
    const AumXdtArgs& g = *p;
    if (g.ntok <= 0 || g.dim <= 0 || g.rank <= 0 || g.ncols <= 0 || g.ldu < g.dim || g.ldwx < g.dim || g.ldwdt < g.rank || g.ldx < g.ncols ||
        g.ldd < g.dim)
        return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    if (g.ncols != XDT_COLS || g.rank % 8 || g.rank > 64 || g.rank > g.ncols || g.dim % 256 || g.dim > XDT_MAX_DIM) return AUM_E_UNSUPPORTED;
    if (g.ldu % 8 || g.ldwx % 8 || g.ldwdt % 8 || g.ldx % 8 || g.ldd % 8) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.u | (uintptr_t)g.wx | (uintptr_t)g.wdt | (uintptr_t)g.x_dbl | (uintptr_t)g.delta) & 15u) return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
}  // namespace aumx
