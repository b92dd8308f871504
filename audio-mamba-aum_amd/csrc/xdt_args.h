// xdt_args.h -- argument rules of aum_xdt_tm_fwd (include/aum_hip.h, ABI 9), shared by the device library (gemm.hip) and the tests-only
// host build (tests/emu/aum_emu.cpp): no HIP dependency.
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

#define XDT_COLS 80          // columns of x_dbl = dt_rank + 2 d_state the kernels are built for (AuM-Base: 48 + 32)
#define XDT_COLS_SMALL 56    // ... and the forward kernel also for AuM-Small's 24 + 32
#ifndef XDT_TOK_W
#define XDT_TOK_W 16         // tokens per wave
#endif
#define XDT_WAVES_MIN 8      // waves (of XDT_TOK_W tokens) per workgroup: xdt_waves picks 8 .. 12 from the token count
#define XDT_WAVES_MAX 12
#define XDT_MAX_DIM 1536     // W_x passes through LDS in two K-halves of XDT_COLS rows: 80 x (768 x 2 + 16) bytes

namespace aumx {
inline int xdt_check(const AumXdtArgs* p) {
    if (!p || !p->u || !p->wx || !p->wdt || !p->x_dbl || !p->delta) return AUM_E_NULL;
    const AumXdtArgs& g = *p;
    if (g.ntok <= 0 || g.dim <= 0 || g.rank <= 0 || g.ncols <= 0 || g.ldu < g.dim || g.ldwx < g.dim || g.ldwdt < g.rank || g.ldx < g.ncols ||
        g.ldd < g.dim)
        return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    if ((g.ncols != XDT_COLS && g.ncols != XDT_COLS_SMALL) || g.rank % 8 || g.rank > 64 || g.rank > g.ncols || g.dim % 256 || g.dim > XDT_MAX_DIM) return AUM_E_UNSUPPORTED;
    if (g.ldu % 8 || g.ldwx % 8 || g.ldwdt % 8 || g.ldx % 8 || g.ldd % 8) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.u | (uintptr_t)g.wx | (uintptr_t)g.wdt | (uintptr_t)g.x_dbl | (uintptr_t)g.delta) & 15u) return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
inline int xdt_bwd_check(const AumXdtBwdArgs* p) {
    if (!p || !p->ddelta || !p->dbc || !p->wdt_t || !p->wx_t || !p->du || !p->dx_dbl) return AUM_E_NULL;
    const AumXdtBwdArgs& g = *p;
    if (g.ntok <= 0 || g.dim <= 0 || g.rank <= 0 || g.ncols <= 0 || g.ldd < g.dim || g.ldu < g.dim || g.ldwdt < g.dim || g.ldwx < g.ncols ||
        g.ldx < g.ncols || g.lddbc < g.ncols - g.rank)
        return AUM_E_SHAPE;
    if (g.dtype != AUM_BF16 && g.dtype != AUM_F16) return AUM_E_DTYPE;
    if (g.ncols != XDT_COLS || g.rank != XDT_COLS - 32 || g.dim % 256 || g.dim > XDT_MAX_DIM) return AUM_E_UNSUPPORTED;
    if (g.ldd % 8 || g.ldu % 8 || g.ldwdt % 8 || g.ldwx % 8 || g.ldx % 8 || g.lddbc % 4) return AUM_E_UNSUPPORTED;
    if (((uintptr_t)g.ddelta | (uintptr_t)g.dbc | (uintptr_t)g.wdt_t | (uintptr_t)g.wx_t | (uintptr_t)g.du | (uintptr_t)g.dx_dbl) & 15u)
        return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
// waves per workgroup for `ntok` tokens on `ncu` CUs (one workgroup per CU at a time): the count whose launch takes the fewest
// wave-rounds, rounds x waves -- 2052 fragments on 256 CUs: 8 waves = 257 workgroups = 2 rounds (16), 9 waves = 228 = 1 round (9)
inline int xdt_waves(int64_t ntok, int ncu) {
    const int64_t frags = (ntok + XDT_TOK_W - 1) / XDT_TOK_W;
    int best = XDT_WAVES_MIN;
    int64_t best_cost = -1;
    for (int nw = XDT_WAVES_MIN; nw <= XDT_WAVES_MAX; ++nw) {
        const int64_t wgs = (frags + nw - 1) / nw, cost = (wgs + ncu - 1) / ncu * nw;
        if (best_cost < 0 || cost < best_cost) {
            best = nw;
            best_cost = cost;
        }
    }
    return best;
}
}  // namespace aumx
