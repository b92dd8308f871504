// TEST INFRASTRUCTURE ONLY.  Lane-array build of the kernel sources in audio-mamba-aum_amd/csrc: every
// wavefront is stepped on the host with gfx950 lane semantics (csrc/wave.h, AUM_EMU) so the kernels' index
// arithmetic, tails, carries and reductions can be checked against the oracle on a machine with no GPU.
// Exports the same C ABI as libaum_hip.so but takes HOST pointers.  Never loaded by the product path.
#define AUM_EMU 1
#include "../../audio-mamba-aum_amd/csrc/aum_api.inc"

// aum_gemm_tn on host pointers: the device kernel (csrc/gemm_kernels.h) is plain HIP around MFMA and LDS-DMA instructions and has no
// lane-array build; the host tests of the projection dispatch get the same contract -- argument rules (gemm_args.h), fp32
// accumulation, one rounding at the store -- from this loop.  The kernel's tile layout is checked on its own in tests/test_gemm_layout.py.
#include "../../audio-mamba-aum_amd/csrc/gemm_args.h"
extern "C" int aum_gemm_tn(const AumGemmArgs* p, void*) {
    const int rc = aumg::gemm_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* a = static_cast<const T*>(p->a);
        const T* b = static_cast<const T*>(p->b);
        T* c = static_cast<T*>(p->c);
        for (int i = 0; i < p->m; ++i)
            for (int j = 0; j < p->n; ++j) {
                float acc = 0.f;
                for (int k = 0; k < p->k; ++k) acc += aum::elem_to_f32(a[(int64_t)i * p->lda + k]) * aum::elem_to_f32(b[(int64_t)j * p->ldb + k]);
                aum::f32_to_elem(acc, c[(int64_t)i * p->ldc + j]);
            }
    };
    if (p->dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}

// aum_gemm_wgrad on host pointers: same arrangement (shared argument rules and split boundaries, arithmetic as a plain loop).
extern "C" int aum_gemm_wgrad(const AumGemmWArgs* p, void*) {
    const int rc = aumg::gemm_wgrad_check(p);
    if (rc != AUM_OK) return rc;
    const int64_t chunk = (((p->t + p->splits - 1) / p->splits) + 63) / 64 * 64;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* y = static_cast<const T*>(p->y);
        const T* x = static_cast<const T*>(p->x);
        for (int s = 0; s < p->splits; ++s) {
            const int64_t t0 = s * chunk, t1 = t0 + chunk < p->t ? t0 + chunk : p->t;
            for (int n = 0; n < p->n; ++n)
                for (int k = 0; k < p->k; ++k) {
                    float acc = 0.f;
                    for (int64_t t = t0; t < t1; ++t) acc += aum::elem_to_f32(y[t * p->ldy + n]) * aum::elem_to_f32(x[t * p->ldx + k]);
                    p->part[((int64_t)s * p->n + n) * p->k + k] = acc;
                }
        }
    };
    if (p->dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}

// aum_dtproj_tm_fwd on host pointers: same arrangement as aum_gemm_tn above (argument rules shared, arithmetic as a plain loop).
#include "../../audio-mamba-aum_amd/csrc/dtproj_args.h"
extern "C" int aum_dtproj_tm_fwd(const AumDtProjArgs* p, void*) {
    const int rc = aumd::dtproj_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* x = static_cast<const T*>(p->x);
        const T* w = static_cast<const T*>(p->w);
        T* o = static_cast<T*>(p->out);
        for (int64_t t = 0; t < p->ntok; ++t)
            for (int e = 0; e < p->dim; ++e) {
                float acc = 0.f;
                for (int k = 0; k < p->rank; ++k) acc += aum::elem_to_f32(x[t * p->ldx + k]) * aum::elem_to_f32(w[(int64_t)e * p->ldw + k]);
                aum::f32_to_elem(acc, o[t * p->ldo + e]);
            }
    };
    if (p->dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}

// aum_xdt_tm_fwd on host pointers: same arrangement (shared argument rules; x_dbl rounded once, delta from the rounded x_dbl).
#include "../../audio-mamba-aum_amd/csrc/xdt_args.h"
extern "C" int aum_xdt_tm_fwd(const AumXdtArgs* p, void*) {
    const int rc = aumx::xdt_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* u = static_cast<const T*>(p->u);
        const T* wx = static_cast<const T*>(p->wx);
        const T* wd = static_cast<const T*>(p->wdt);
        T* x = static_cast<T*>(p->x_dbl);
        T* d = static_cast<T*>(p->delta);
        for (int64_t t = 0; t < p->ntok; ++t) {
            for (int c = 0; c < p->ncols; ++c) {
                float acc = 0.f;
                for (int k = 0; k < p->dim; ++k) acc += aum::elem_to_f32(u[t * p->ldu + k]) * aum::elem_to_f32(wx[(int64_t)c * p->ldwx + k]);
                aum::f32_to_elem(acc, x[t * p->ldx + c]);
            }
            for (int e = 0; e < p->dim; ++e) {
                float acc = 0.f;
                for (int k = 0; k < p->rank; ++k) acc += aum::elem_to_f32(x[t * p->ldx + k]) * aum::elem_to_f32(wd[(int64_t)e * p->ldwdt + k]);
                aum::f32_to_elem(acc, d[t * p->ldd + e]);
            }
        }
    };
    if (p->dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}
