// decode_args.h -- argument rules of the single-token (streaming inference) kernels aum_causal_conv1d_update / aum_selective_state_update
// (include/aum_hip.h, ABI 10), shared by the device library (gemm.hip) and the tests-only host build (tests/emu/aum_emu.cpp).
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

namespace aumdec {
inline int conv_update_check(const AumConvUpdateArgs* p) {
    if (!p || !p->x || !p->conv_state || !p->weight || !p->out) return AUM_E_NULL;
    if (p->batch <= 0 || p->dim <= 0 || p->width <= 0) return AUM_E_SHAPE;
    if (p->dtype < 0 || p->dtype > 2) return AUM_E_DTYPE;
    if (p->width > 8) return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
inline int state_update_check(const AumStateUpdateArgs* p) {
    if (!p || !p->state || !p->x || !p->dt || !p->A || !p->B || !p->C || !p->out) return AUM_E_NULL;
    if (p->batch <= 0 || p->dim <= 0 || p->dstate <= 0) return AUM_E_SHAPE;
    if (p->dtype < 0 || p->dtype > 2) return AUM_E_DTYPE;
    if (p->dstate > 256) return AUM_E_UNSUPPORTED;
    return AUM_OK;
}
}  // namespace aumdec
