// gemm.hip -- translation unit of the plain-HIP MFMA kernels: the dense projection GEMM (gemm_kernels.h, aum_gemm_tn) and the dt projection of
// the token-major block (dtproj_kernels.h, aum_dtproj_tm_fwd); include/aum_hip.h, ABI 9.
#include <atomic>
#include "gemm_kernels.h"
#include "gemm_ps_kernels.h"
#include "dtproj_kernels.h"
#include "xdt_kernels.h"
#include "decode_kernels.h"
#include "cast_kernels.h"

// CU count of the CURRENT device (one process may drive several devices from several threads): a small per-device cache, filled with
// relaxed atomics -- racing fillers write the same value
static int cu_count() {
    static std::atomic<int> ncu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int ncu = ncu_of[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return 0;
        ncu_of[dev].store(ncu, std::memory_order_relaxed);
    }
    return ncu;
}

extern "C" int aum_gemm_tn(const AumGemmArgs* p, void* stream) {
    const int rc = aumg::gemm_check(p);
    if (rc != AUM_OK) return rc;
    const AumGemmArgs& g = *p;
    const int tiles = (g.m + aumg::BM - 1) / aumg::BM * (g.n / aumg::BN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ncu = cu_count();
    if (ncu <= 0) return AUM_E_LAUNCH;
    (void)hipGetLastError();        // a stale error of an earlier, unrelated launch must not be reported as this one's
    // Which kernel: the paced-store kernel (round 6, gemm_ps_kernels.h: one workgroup per CU walking a tile list, a tile's stores under the
    // next tile's K-steps) whenever a tile has the seven K-steps its store pacing needs -- every projection of the model (k >= 768);
    // shorter products: one workgroup per tile, software-pipelined K loop (round 4).  AUM_GEMM_LOCKSTEP / _PIPELINED / _PACED name one.
    uint32_t flags = g.flags & (AUM_GEMM_LOCKSTEP | AUM_GEMM_PIPELINED | AUM_GEMM_PACED);
    if (!flags) flags = g.k >= (aumg::PS_NST + 1) * aumg::BK ? AUM_GEMM_PACED : AUM_GEMM_PIPELINED;
    if (flags & AUM_GEMM_PACED) {
        if (g.k < (aumg::PS_NST + 1) * aumg::BK) return AUM_E_UNSUPPORTED;
        aumg::GemmLaunch L;
        L.g = g;
        aumg::gemm_ps_items(g.m, g.n, ncu, &L);
        const int grid = L.nitems < ncu ? L.nitems : ncu;
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL(aumg::k_gemm_tn_ps<true>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        else hipLaunchKernelGGL(aumg::k_gemm_tn_ps<false>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    const bool bf = g.dtype == AUM_BF16;
    if (flags & AUM_GEMM_PIPELINED) {
        if (bf) hipLaunchKernelGGL((aumg::k_gemm_tn<true, 2>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<false, 2>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    } else {
        if (bf) hipLaunchKernelGGL((aumg::k_gemm_tn<true, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<false, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    }
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

extern "C" int aum_gemm_wgrad(const AumGemmWArgs* p, void* stream) {
    const int rc = aumg::gemm_wgrad_check(p);
    if (rc != AUM_OK) return rc;
    const AumGemmWArgs& g = *p;
    const bool skinny = g.k % 256 != 0;
    const int nitems = (g.n / 256) * (skinny ? 1 : g.k / 256) * g.splits;
    const int grid = (nitems + 7) / 8 * 8;          // eight XCD runs of equal length; the surplus workgroups return at once
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const bool bf = g.dtype == AUM_BF16;
    if (skinny && g.k == 48) {
        if (bf) hipLaunchKernelGGL((aumg::k_gemm_wgrad_skinny<true, 3>), dim3(grid), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_wgrad_skinny<false, 3>), dim3(grid), dim3(aumg::THREADS), 0, s, g);
    } else if (skinny) {
        if (bf) hipLaunchKernelGGL((aumg::k_gemm_wgrad_skinny<true, 5>), dim3(grid), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_wgrad_skinny<false, 5>), dim3(grid), dim3(aumg::THREADS), 0, s, g);
    } else if (bf) hipLaunchKernelGGL(aumg::k_gemm_wgrad<true>, dim3(grid), dim3(aumg::THREADS), 0, s, g);
    else hipLaunchKernelGGL(aumg::k_gemm_wgrad<false>, dim3(grid), dim3(aumg::THREADS), 0, s, g);
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

extern "C" int aum_dtproj_tm_fwd(const AumDtProjArgs* p, void* stream) {
    const int rc = aumd::dtproj_check(p);
    if (rc != AUM_OK) return rc;
    const AumDtProjArgs& g = *p;
    const int64_t nwaves = (g.ntok + aumd::TOK_PER_WAVE - 1) / aumd::TOK_PER_WAVE;
    const dim3 grid((unsigned)((nwaves + aumd::WAVES - 1) / aumd::WAVES)), block(aumd::WAVES * 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool bf = g.dtype == AUM_BF16, one = g.rank <= 32;
    if (bf && one) hipLaunchKernelGGL((aumd::k_dtproj_tm<true, 1>), grid, block, 0, s, g);
    else if (bf) hipLaunchKernelGGL((aumd::k_dtproj_tm<true, 2>), grid, block, 0, s, g);
    else if (one) hipLaunchKernelGGL((aumd::k_dtproj_tm<false, 1>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((aumd::k_dtproj_tm<false, 2>), grid, block, 0, s, g);
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}


template <int NW, int NC> static void xdt_launch_nc(const AumXdtArgs& g, hipStream_t s) {
    const dim3 grid((unsigned)((g.ntok + NW * XDT_TOK_W - 1) / (NW * XDT_TOK_W))), block(NW * 64);
    const bool bf = g.dtype == AUM_BF16, one = g.rank <= 32;
    if (bf && one) hipLaunchKernelGGL((aumx::k_xdt_tm_fwd<true, 1, NW, NC>), grid, block, 0, s, g);
    else if (bf) hipLaunchKernelGGL((aumx::k_xdt_tm_fwd<true, 2, NW, NC>), grid, block, 0, s, g);
    else if (one) hipLaunchKernelGGL((aumx::k_xdt_tm_fwd<false, 1, NW, NC>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((aumx::k_xdt_tm_fwd<false, 2, NW, NC>), grid, block, 0, s, g);
}
template <int NW> static void xdt_launch(const AumXdtArgs& g, hipStream_t s) {
    if (g.ncols == XDT_COLS) xdt_launch_nc<NW, XDT_COLS>(g, s);
    else xdt_launch_nc<NW, XDT_COLS_SMALL>(g, s);
}

extern "C" int aum_xdt_tm_fwd(const AumXdtArgs* p, void* stream) {
    const int rc = aumx::xdt_check(p);
    if (rc != AUM_OK) return rc;
    const AumXdtArgs& g = *p;
    const int ncu = cu_count();
    if (ncu <= 0) return AUM_E_LAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    switch (aumx::xdt_waves(g.ntok, ncu)) {
        case 8: xdt_launch<8>(g, s); break;
        case 9: xdt_launch<9>(g, s); break;
        case 10: xdt_launch<10>(g, s); break;
        case 11: xdt_launch<11>(g, s); break;
        default: xdt_launch<12>(g, s); break;
    }
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

template <int NW> static void xdt_bwd_launch(const AumXdtBwdArgs& g, hipStream_t s) {
    const dim3 grid((unsigned)((g.ntok + NW * XDT_TOK_W - 1) / (NW * XDT_TOK_W))), block(NW * 64);
    const bool bf = g.dtype == AUM_BF16, deep = g.dim % 512 == 0;          // du read-ahead of four channel pairs where a quarter of the channels holds a multiple of four
    if (bf && deep) hipLaunchKernelGGL((aumx::k_xdt_tm_bwd<true, 3, NW, 4>), grid, block, 0, s, g);
    else if (bf) hipLaunchKernelGGL((aumx::k_xdt_tm_bwd<true, 3, NW, 2>), grid, block, 0, s, g);
    else if (deep) hipLaunchKernelGGL((aumx::k_xdt_tm_bwd<false, 3, NW, 4>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((aumx::k_xdt_tm_bwd<false, 3, NW, 2>), grid, block, 0, s, g);
}

extern "C" int aum_xdt_tm_bwd(const AumXdtBwdArgs* p, void* stream) {
    const int rc = aumx::xdt_bwd_check(p);
    if (rc != AUM_OK) return rc;
    const AumXdtBwdArgs& g = *p;
    const int ncu = cu_count();
    if (ncu <= 0) return AUM_E_LAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    switch (aumx::xdt_waves(g.ntok, ncu)) {
        case 8: xdt_bwd_launch<8>(g, s); break;
        case 9: xdt_bwd_launch<9>(g, s); break;
        case 10: xdt_bwd_launch<10>(g, s); break;
        case 11: xdt_bwd_launch<11>(g, s); break;
        default: xdt_bwd_launch<12>(g, s); break;
    }
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

extern "C" int aum_causal_conv1d_update(const AumConvUpdateArgs* p, void* stream) {
    const int rc = aumdec::conv_update_check(p);
    if (rc != AUM_OK) return rc;
    const int64_t n = (int64_t)p->batch * p->dim;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (p->dtype == AUM_F32) hipLaunchKernelGGL(aumdec::k_conv_update<float>, grid, block, 0, s, *p);
    else if (p->dtype == AUM_BF16) hipLaunchKernelGGL(aumdec::k_conv_update<__bf16>, grid, block, 0, s, *p);
    else hipLaunchKernelGGL(aumdec::k_conv_update<_Float16>, grid, block, 0, s, *p);
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

extern "C" int aum_selective_state_update(const AumStateUpdateArgs* p, void* stream) {
    const int rc = aumdec::state_update_check(p);
    if (rc != AUM_OK) return rc;
    const int64_t n = (int64_t)p->batch * p->dim;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (p->dtype == AUM_F32) hipLaunchKernelGGL(aumdec::k_state_update<float>, grid, block, 0, s, *p);
    else if (p->dtype == AUM_BF16) hipLaunchKernelGGL(aumdec::k_state_update<__bf16>, grid, block, 0, s, *p);
    else hipLaunchKernelGGL(aumdec::k_state_update<_Float16>, grid, block, 0, s, *p);
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

extern "C" int aum_cast_bank(const uint64_t* src, int32_t n, int32_t rows, int32_t cols, void* bank, void* bank_t, int32_t dtype, void* stream) {
    const int rc = aumc::cast_bank_check(src, bank, bank_t, n, rows, cols, dtype);
    if (rc != AUM_OK) return rc;
    aumc::CastBank a;
    a.src = src, a.bank = bank, a.bank_t = bank_t, a.n = n, a.rows = rows, a.cols = cols;
    a.tiles_r = (rows + aumc::CT - 1) / aumc::CT, a.tiles_c = (cols + aumc::CT - 1) / aumc::CT;
    const unsigned grid = (unsigned)((int64_t)a.tiles_r * a.tiles_c * n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (dtype == AUM_BF16) hipLaunchKernelGGL(aumc::k_cast_bank<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(aumc::k_cast_bank<false>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}
