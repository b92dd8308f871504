// gemm.hip -- translation unit of the dense projection GEMM (gemm_kernels.h) and its C-ABI entry point aum_gemm_tn (include/aum_hip.h, ABI 9).
#include "gemm_kernels.h"

extern "C" int aum_gemm_tn(const AumGemmArgs* p, void* stream) {
    const int rc = aumg::gemm_check(p);
    if (rc != AUM_OK) return rc;
    const AumGemmArgs& g = *p;
    const int tiles = (g.m + aumg::BM - 1) / aumg::BM * (g.n / aumg::BN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lockstep = (g.flags & AUM_GEMM_LOCKSTEP) != 0;
    if (g.dtype == AUM_BF16) {
        if (lockstep) hipLaunchKernelGGL((aumg::k_gemm_tn<true, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<true, 1>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    } else {
        if (lockstep) hipLaunchKernelGGL((aumg::k_gemm_tn<false, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<false, 1>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    }
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}
