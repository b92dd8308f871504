// gemm.hip -- translation unit of the plain-HIP MFMA kernels: the dense projection GEMM (gemm_kernels.h, aum_gemm_tn) and the dt projection of
// the token-major block (dtproj_kernels.h, aum_dtproj_tm_fwd); include/aum_hip.h, ABI 9.
#include <atomic>
#include "gemm_kernels.h"
#include "gemm_w4_kernels.h"
#include "gemm_ring_kernels.h"
#include "gemm_ps_kernels.h"
#include "dtproj_kernels.h"
#include "xdt_kernels.h"
#include "decode_kernels.h"

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
    // CU count of the CURRENT device (one process may drive several devices from several threads): a small per-device cache, filled
    // with relaxed atomics -- racing fillers write the same value
    static std::atomic<int> ncu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return AUM_E_LAUNCH;
    int ncu = ncu_of[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return AUM_E_LAUNCH;
        ncu_of[dev].store(ncu, std::memory_order_relaxed);
    }
    (void)hipGetLastError();        // a stale error of an earlier, unrelated launch must not be reported as this one's
    // Which kernel: the persistent one pays off when a CU gets several tiles (cheap ragged row block, next tile prefetched under the stores:
    // 774 / 1548 tiles at N = 1536 / 3072: 84.6 vs 94.2 us, 158.6 vs 169.1 us); with at most two tiles per CU (N = 768: 387 tiles) one
    // workgroup per tile is as fast or faster (84.3 vs 87.4 us, 159.4 vs 162.2 us; profiles/r03_gemm_probe.txt) -- since round 4 with the
    // software-pipelined K loop (SCHED 2: 74.6 / 134.7 us on the box where the persistent kernel took 82.7 / 154.2)
    uint32_t flags = g.flags;
    // default (no schedule named): the paced-store kernel of round 6 whenever a tile has the seven K-steps its store pacing needs (every
    // projection of the model: k >= 768); shorter products keep the round-3 / round-4 kernels below
    if (!(flags & (AUM_GEMM_LOCKSTEP | AUM_GEMM_STAGGERED | AUM_GEMM_PERSISTENT | AUM_GEMM_PIPELINED | AUM_GEMM_W4 | AUM_GEMM_RING | AUM_GEMM_PACED))
        && g.k >= (aumg::PS_NST + 1) * aumg::BK)
        flags |= AUM_GEMM_PACED;
    if (flags & AUM_GEMM_PACED) {
        if (g.k < (aumg::PS_NST + 1) * aumg::BK) return AUM_E_UNSUPPORTED;
        aumg::GemmLaunch L;
        L.g = g;
        L.full_rb = (g.m + aumg::BM - 1) / aumg::BM;
        L.half_rb = 0;
        L.nitems = L.full_rb * (g.n / aumg::BN);
        const int grid = L.nitems < ncu ? L.nitems : ncu;
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL(aumg::k_gemm_tn_ps<true>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        else hipLaunchKernelGGL(aumg::k_gemm_tn_ps<false>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    if (flags & AUM_GEMM_RING) {
        if (g.n % aumg::RING_BN || g.k < 8 * aumg::RING_BK) return AUM_E_UNSUPPORTED;
        aumg::GemmLaunch L;
        L.g = g;
        L.full_rb = (g.m + aumg::BM - 1) / aumg::BM;
        L.half_rb = 0;
        L.nitems = L.full_rb * (g.n / aumg::RING_BN);
        const int grid = L.nitems < ncu ? L.nitems : ncu;
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL(aumg::k_gemm_tn_ring<true>, dim3(grid), dim3(aumg::W4_THREADS), 0, s, L);
        else hipLaunchKernelGGL(aumg::k_gemm_tn_ring<false>, dim3(grid), dim3(aumg::W4_THREADS), 0, s, L);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    if (flags & AUM_GEMM_W4) {
        if (g.n % 192 || g.k < 2 * aumg::BK) return AUM_E_UNSUPPORTED;
        aumg::GemmLaunch L;
        L.g = g;
        L.full_rb = (g.m + aumg::BM - 1) / aumg::BM;
        L.half_rb = 0;
        L.nitems = L.full_rb * (g.n / 192);
        const int grid = L.nitems < ncu ? L.nitems : ncu;
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL((aumg::k_gemm_tn_w4<true, 6>), dim3(grid), dim3(aumg::W4_THREADS), 0, s, L);
        else hipLaunchKernelGGL((aumg::k_gemm_tn_w4<false, 6>), dim3(grid), dim3(aumg::W4_THREADS), 0, s, L);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    if (!(flags & (AUM_GEMM_LOCKSTEP | AUM_GEMM_STAGGERED | AUM_GEMM_PERSISTENT | AUM_GEMM_PIPELINED))) flags |= tiles > 2 * ncu ? AUM_GEMM_PERSISTENT : AUM_GEMM_PIPELINED;
    if (flags & AUM_GEMM_PERSISTENT) {
        aumg::GemmLaunch L;
        L.g = g;
        const int rem = g.m % aumg::BM;
        L.full_rb = g.m / aumg::BM + (rem > 128 ? 1 : 0);            // a remainder above 128 rows is a full item whose last rows are out of range
        L.half_rb = rem > 0 && rem <= 128 ? 1 : 0;
        L.nitems = (L.full_rb + L.half_rb) * (g.n / aumg::BN);
        const int grid = L.nitems < ncu ? L.nitems : ncu;
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL(aumg::k_gemm_tn_persistent<true>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        else hipLaunchKernelGGL(aumg::k_gemm_tn_persistent<false>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    const bool lockstep = (flags & AUM_GEMM_LOCKSTEP) != 0;
    if (flags & AUM_GEMM_PIPELINED) {
        if (g.dtype == AUM_BF16) hipLaunchKernelGGL((aumg::k_gemm_tn<true, 2>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<false, 2>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
    }
    if (g.dtype == AUM_BF16) {
        if (lockstep) hipLaunchKernelGGL((aumg::k_gemm_tn<true, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<true, 1>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    } else {
        if (lockstep) hipLaunchKernelGGL((aumg::k_gemm_tn<false, 0>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
        else hipLaunchKernelGGL((aumg::k_gemm_tn<false, 1>), dim3(tiles), dim3(aumg::THREADS), 0, s, g);
    }
    return hipGetLastError() == hipSuccess ? AUM_OK : AUM_E_LAUNCH;
}

// split-tail geometry for (m, n) on `ncu` CUs: grid (a multiple of 8 workgroups), complete rounds, tail tiles; applies when the tail is
// at least half a round (every tile then has at most SK_SLOTS + 1 contributors)
static bool sk_geometry(int64_t m, int n, int ncu, int* grid, int* rounds, int* tail) {
    if (m <= 0 || n <= 0 || n % aumg::BN) return false;
    const int g8 = ncu / 8 * 8;
    if (g8 < 8) return false;
    const int64_t tiles = (m + aumg::BM - 1) / aumg::BM * (n / aumg::BN);
    *grid = g8;
    *rounds = (int)(tiles / g8);
    *tail = (int)(tiles - (int64_t)*rounds * g8);
    return *tail * 2 >= g8 && *tail < g8;
}
static int64_t sk_flag_bytes(int tail) { return ((int64_t)(tail * aumg::SK_SLOTS + 1) * 4 + 255) / 256 * 256; }

extern "C" int64_t aum_gemm_tn_sk_workspace_bytes(int64_t m, int32_t n) {
    const int ncu = cu_count();
    int grid, rounds, tail;
    if (ncu <= 0 || !sk_geometry(m, n, ncu, &grid, &rounds, &tail)) return 0;
    return sk_flag_bytes(tail) + (int64_t)tail * aumg::SK_SLOTS * aumg::BM * aumg::BN * 4;
}

extern "C" int aum_gemm_tn_sk(const AumGemmSkArgs* p, void* stream) {
    if (!p) return AUM_E_NULL;
    const int rc = aumg::gemm_check(&p->base);
    if (rc != AUM_OK) return rc;
    if (!p->workspace || !p->epoch) return AUM_E_NULL;
    const AumGemmArgs& g = p->base;
    const int ncu = cu_count();
    int grid, rounds, tail;
    if (ncu <= 0) return AUM_E_LAUNCH;
    if (!sk_geometry(g.m, g.n, ncu, &grid, &rounds, &tail)) return AUM_E_UNSUPPORTED;
    const int64_t fb = sk_flag_bytes(tail);
    if (((uintptr_t)p->workspace & 255u) || p->workspace_bytes < fb + (int64_t)tail * aumg::SK_SLOTS * aumg::BM * aumg::BN * 4) return AUM_E_WORKSPACE;
    aumg::GemmSkLaunch L;
    L.g = g;
    L.err = static_cast<uint32_t*>(p->workspace);                  // word 0: set when a bounded wait ran out; flags behind it
    L.flags = L.err + 1;
    L.part = reinterpret_cast<float*>(static_cast<char*>(p->workspace) + fb);
    L.epoch = p->epoch;
    L.rounds = rounds;
    L.tail = tail;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (g.dtype == AUM_BF16) hipLaunchKernelGGL(aumg::k_gemm_tn_sk<true>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
    else hipLaunchKernelGGL(aumg::k_gemm_tn_sk<false>, dim3(grid), dim3(aumg::THREADS), 0, s, L);
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
