// cast_args.h -- argument rules of aum_cast_bank (include/aum_hip.h, ABI 13), shared by the device library (gemm.hip) and the tests-only host
// build (tests/emu/aum_emu.cpp): no HIP dependency.
#pragma once
#include <stdint.h>

#include "../../include/aum_hip.h"

namespace aumc {
constexpr int CT = 64;                  // tile edge of the kernel
inline int cast_bank_check(const uint64_t* src, const void* bank, const void* bank_t, int32_t n, int32_t rows, int32_t cols, int32_t dtype) {
    if (!src || !bank) return AUM_E_NULL;
    if (n <= 0 || rows <= 0 || cols <= 0 || (rows & 7) || (cols & 3)) return AUM_E_SHAPE;
    if (dtype != AUM_BF16 && dtype != AUM_F16) return AUM_E_DTYPE;
    if (((uintptr_t)bank | (uintptr_t)bank_t) & 15u) return AUM_E_UNSUPPORTED;
    const int64_t tiles = (int64_t)((rows + CT - 1) / CT) * ((cols + CT - 1) / CT) * n;
    if (tiles > 0x7fffffff) return AUM_E_SHAPE;
    return AUM_OK;
}
}  // namespace aumc
