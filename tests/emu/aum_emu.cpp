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

// aum_xdt_tm_bwd on host pointers: dx_dbl = [ddelta . W_dt rounded | dB | dC], du += dx_dbl . W_x from the ROUNDED dx_dbl
extern "C" int aum_xdt_tm_bwd(const AumXdtBwdArgs* p, void*) {
    const int rc = aumx::xdt_bwd_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* dd = static_cast<const T*>(p->ddelta);
        const T* wd = static_cast<const T*>(p->wdt_t);
        const T* wx = static_cast<const T*>(p->wx_t);
        T* du = static_cast<T*>(p->du);
        T* dx = static_cast<T*>(p->dx_dbl);
        for (int64_t t = 0; t < p->ntok; ++t) {
            for (int r = 0; r < p->rank; ++r) {
                float acc = 0.f;
                for (int e = 0; e < p->dim; ++e) acc += aum::elem_to_f32(dd[t * p->ldd + e]) * aum::elem_to_f32(wd[(int64_t)r * p->ldwdt + e]);
                aum::f32_to_elem(acc, dx[t * p->ldx + r]);
            }
            for (int c = p->rank; c < p->ncols; ++c) aum::f32_to_elem(p->dbc[t * p->lddbc + c - p->rank], dx[t * p->ldx + c]);
            for (int e = 0; e < p->dim; ++e) {
                float acc = 0.f;
                for (int c = 0; c < p->ncols; ++c) acc += aum::elem_to_f32(dx[t * p->ldx + c]) * aum::elem_to_f32(wx[(int64_t)e * p->ldwx + c]);
                aum::f32_to_elem(acc + aum::elem_to_f32(du[t * p->ldu + e]), du[t * p->ldu + e]);
            }
        }
    };
    if (p->dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}

// the per-token decode kernels on host pointers: shared argument rules (decode_args.h), the arithmetic of csrc/decode_kernels.h as plain loops
#include "../../audio-mamba-aum_amd/csrc/decode_args.h"
extern "C" int aum_causal_conv1d_update(const AumConvUpdateArgs* p, void*) {
    const int rc = aumdec::conv_update_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* x = static_cast<const T*>(p->x);
        T* out = static_cast<T*>(p->out);
        for (int64_t i = 0; i < (int64_t)p->batch * p->dim; ++i) {
            const int d = (int)(i % p->dim);
            float* win = p->conv_state + i * p->width;
            const float* w = p->weight + (int64_t)d * p->width;
            float acc = p->bias ? p->bias[d] : 0.f;
            for (int k = 0; k + 1 < p->width; ++k) { win[k] = win[k + 1]; acc += win[k] * w[k]; }
            win[p->width - 1] = aum::elem_to_f32(x[i]);
            acc += win[p->width - 1] * w[p->width - 1];
            if (p->flags & AUM_CONV_SILU) acc = acc / (1.f + std::exp(-acc));
            aum::f32_to_elem(acc, out[i]);
        }
    };
    if (p->dtype == AUM_F32) run(float{}); else if (p->dtype == AUM_BF16) run(aum::bf16_t{}); else run(aum::f16_t{});
    return AUM_OK;
}
extern "C" int aum_selective_state_update(const AumStateUpdateArgs* p, void*) {
    const int rc = aumdec::state_update_check(p);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T *x = static_cast<const T*>(p->x), *dtp = static_cast<const T*>(p->dt), *z = static_cast<const T*>(p->z);
        const T *Bm = static_cast<const T*>(p->B), *Cm = static_cast<const T*>(p->C);
        T* out = static_cast<T*>(p->out);
        for (int64_t i = 0; i < (int64_t)p->batch * p->dim; ++i) {
            const int d = (int)(i % p->dim);
            const int64_t b = i / p->dim;
            float dt = aum::elem_to_f32(dtp[i]) + (p->dt_bias ? p->dt_bias[d] : 0.f);
            if (p->flags & AUM_SCAN_SOFTPLUS) dt = dt > 20.f ? dt : std::log1p(std::exp(dt));
            const float xv = aum::elem_to_f32(x[i]);
            float y = 0.f;
            for (int n = 0; n < p->dstate; ++n) {
                float& h = p->state[i * p->dstate + n];
                h = std::exp(dt * p->A[(int64_t)d * p->dstate + n]) * h + dt * xv * aum::elem_to_f32(Bm[b * p->dstate + n]);
                y += h * aum::elem_to_f32(Cm[b * p->dstate + n]);
            }
            if (p->D) y += p->D[d] * xv;
            if (z) { const float zv = aum::elem_to_f32(z[i]); y *= zv / (1.f + std::exp(-zv)); }
            aum::f32_to_elem(y, out[i]);
        }
    };
    if (p->dtype == AUM_F32) run(float{}); else if (p->dtype == AUM_BF16) run(aum::bf16_t{}); else run(aum::f16_t{});
    return AUM_OK;
}

// aum_cast_bank on host pointers: argument rules (cast_args.h) and one rounding per element
#include "../../audio-mamba-aum_amd/csrc/cast_args.h"
extern "C" int aum_cast_bank(const uint64_t* src, int32_t n, int32_t rows, int32_t cols, void* bank, void* bank_t, int32_t dtype, void*) {
    const int rc = aumc::cast_bank_check(src, bank, bank_t, n, rows, cols, dtype);
    if (rc != AUM_OK) return rc;
    auto run = [&](auto tag) {
        typedef decltype(tag) T;
        T* b = static_cast<T*>(bank);
        T* bt = static_cast<T*>(bank_t);
        for (int m = 0; m < n; ++m) {
            const float* a = reinterpret_cast<const float*>(src[m]);
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols; ++c) {
                    aum::f32_to_elem(a[(int64_t)r * cols + c], b[((int64_t)m * rows + r) * cols + c]);
                    if (bt) bt[((int64_t)m * cols + c) * rows + r] = b[((int64_t)m * rows + r) * cols + c];
                }
        }
    };
    if (dtype == AUM_BF16) run(aum::bf16_t{});
    else run(aum::f16_t{});
    return AUM_OK;
}
