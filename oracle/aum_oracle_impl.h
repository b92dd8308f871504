/*
 * ORACLE (test infrastructure only) -- CPU restatement of the reference's hot-path arithmetic.
 *
 * This file is included twice by aum_oracle.c, once with REAL=float (suffix _f32: the
 * reference's own fp32 internal arithmetic, SSI:101-103,117-118) and once with REAL=double
 * (suffix _f64: a higher-precision "truth" used to separate kernel error from fp32 round-off).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into it.
 * The product path (libaum_hip.so + the mamba_ssm host package) never links or imports it.
 *
 * Citations: SSI = /root/reference/vim-mamba_ssm/mamba_ssm/ops/selective_scan_interface.py
 *            MS  = /root/reference/vim-mamba_ssm/mamba_ssm/modules/mamba_simple.py
 *            LN  = /root/reference/vim-mamba_ssm/mamba_ssm/ops/triton/layernorm.py
 * All tensors are dense, row-major, with the logical shapes given per function.
 */

#ifndef REAL
#error "include from aum_oracle.c"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* torch.nn.functional.softplus(beta=1, threshold=20): x if x > 20 else log1p(exp(x))  (SSI:106-107) */
static inline REAL FN(softplus_)(REAL x) { return x > (REAL)20 ? x : (REAL)log1p(exp((double)x)); }
static inline REAL FN(sigmoid_)(REAL x) { return (REAL)1 / ((REAL)1 + (REAL)exp(-(double)x)); }

/*
 * Selective scan forward, one direction.  Restates selective_scan_ref (SSI:86-152), real A,
 * variable B/C given as (batch, dstate, len) [the G=1 squeeze of (B,1,N,L), SSI:125-131].
 *   u, delta, z : (batch, dim, len)     A : (dim, dstate)     Bm, Cm : (batch, dstate, len)
 *   D, delta_bias : (dim) or NULL       z : NULL => no gate
 *   reverse != 0 : run the recurrence from t = len-1 down to 0 (equivalent to flipping every
 *                  time-indexed input, scanning, and flipping the output -- SSI:503-507, 707-708)
 * Outputs (any may be NULL): y_pre = y + D*u (the reference kernel's `out`), out = y_pre*silu(z)
 * (or y_pre when z is NULL), last_state (batch, dim, dstate) = x after the final step (SSI:142-143).
 */
void FN(aum_oracle_scan_fwd)(const REAL* u, const REAL* delta, const REAL* A, const REAL* Bm,
                             const REAL* Cm, const REAL* D, const REAL* z, const REAL* delta_bias,
                             int delta_softplus, int reverse, int batch, int dim, int len, int dstate,
                             REAL* y_pre, REAL* out, REAL* last_state) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            REAL x[256];
            for (int n = 0; n < dstate; ++n) x[n] = 0; /* SSI:119 */
            const size_t row = ((size_t)b * dim + d) * len;
            for (int step = 0; step < len; ++step) {
                const int t = reverse ? len - 1 - step : step;
                REAL dt = delta[row + t];
                if (delta_bias) dt += delta_bias[d]; /* SSI:104-105 */
                if (delta_softplus) dt = FN(softplus_)(dt);
                const REAL ut = u[row + t];
                REAL y = 0;
                for (int n = 0; n < dstate; ++n) {
                    const REAL a = (REAL)exp((double)(dt * A[(size_t)d * dstate + n])); /* SSI:121 */
                    const REAL bu = dt * Bm[((size_t)b * dstate + n) * len + t] * ut;   /* SSI:126 */
                    x[n] = a * x[n] + bu;                                               /* SSI:134 */
                    y += x[n] * Cm[((size_t)b * dstate + n) * len + t];                 /* SSI:139 */
                }
                REAL o = D ? y + ut * D[d] : y; /* SSI:148 */
                if (y_pre) y_pre[row + t] = o;
                if (z) {
                    const REAL zt = z[row + t];
                    o = o * (zt * FN(sigmoid_)(zt)); /* SSI:150 */
                }
                if (out) out[row + t] = o;
            }
            if (last_state)
                for (int n = 0; n < dstate; ++n) last_state[((size_t)b * dim + d) * dstate + n] = x[n];
        }
    }
}

/*
 * Selective scan backward, one direction: analytic adjoint of the function above (what
 * selective_scan_cuda.bwd returns at SSI:62-65; derivation in SURVEY.md 8(a')).
 * dout is the gradient w.r.t. `out` (the gated output when z != NULL).
 * Gradient outputs (NULL to skip): du, ddelta, dz : (batch,dim,len); dA : (dim,dstate);
 * dB, dC : (batch,dstate,len); dD, ddelta_bias : (dim).  dA,dB,dC,dD,ddelta_bias are ACCUMULATED
 * into (caller zero-initialises), which lets the bidirectional sum of SSI:554-559 reuse them.
 * Not thread-parallel over d for dB/dC (race-free version: parallel over batch only).
 */
void FN(aum_oracle_scan_bwd)(const REAL* u, const REAL* delta, const REAL* A, const REAL* Bm,
                             const REAL* Cm, const REAL* D, const REAL* z, const REAL* delta_bias,
                             const REAL* dout, int delta_softplus, int reverse, int batch, int dim,
                             int len, int dstate, REAL* du, REAL* ddelta, REAL* dA, REAL* dB, REAL* dC,
                             REAL* dD, REAL* dz, REAL* ddelta_bias) {
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    /* per-thread private accumulators for the (dim,*) reductions */
    REAL* dA_priv = (REAL*)calloc((size_t)nthreads * dim * dstate, sizeof(REAL));
    REAL* dD_priv = (REAL*)calloc((size_t)nthreads * dim, sizeof(REAL));
    REAL* db_priv = (REAL*)calloc((size_t)nthreads * dim, sizeof(REAL));
#pragma omp parallel for schedule(static)
    for (int b = 0; b < batch; ++b) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        REAL* xs = (REAL*)malloc((size_t)(len + 1) * dstate * sizeof(REAL)); /* x_{-1..len-1} in scan order */
        REAL* as = (REAL*)malloc((size_t)len * dstate * sizeof(REAL));
        REAL* dts = (REAL*)malloc((size_t)len * sizeof(REAL));
        REAL g[256];
        for (int d = 0; d < dim; ++d) {
            const size_t row = ((size_t)b * dim + d) * len;
            for (int n = 0; n < dstate; ++n) xs[n] = 0;
            /* forward recompute in scan order (step index s; time index t) */
            for (int s = 0; s < len; ++s) {
                const int t = reverse ? len - 1 - s : s;
                REAL dt = delta[row + t];
                if (delta_bias) dt += delta_bias[d];
                if (delta_softplus) dt = FN(softplus_)(dt);
                dts[s] = dt;
                for (int n = 0; n < dstate; ++n) {
                    const REAL a = (REAL)exp((double)(dt * A[(size_t)d * dstate + n]));
                    as[(size_t)s * dstate + n] = a;
                    xs[(size_t)(s + 1) * dstate + n] =
                        a * xs[(size_t)s * dstate + n] + dt * Bm[((size_t)b * dstate + n) * len + t] * u[row + t];
                }
            }
            for (int n = 0; n < dstate; ++n) g[n] = 0;
            for (int s = len - 1; s >= 0; --s) {
                const int t = reverse ? len - 1 - s : s;
                const REAL ut = u[row + t];
                const REAL dt = dts[s];
                REAL y = 0;
                for (int n = 0; n < dstate; ++n)
                    y += xs[(size_t)(s + 1) * dstate + n] * Cm[((size_t)b * dstate + n) * len + t];
                const REAL ypre = D ? y + ut * D[d] : y;
                REAL dy = dout[row + t];
                if (z) {
                    const REAL zt = z[row + t];
                    const REAL sg = FN(sigmoid_)(zt);
                    if (dz) dz[row + t] = dy * ypre * sg * ((REAL)1 + zt * ((REAL)1 - sg));
                    dy = dy * zt * sg;
                }
                if (D) dD_priv[(size_t)tid * dim + d] += dy * ut;
                REAL ddt = 0, dut = D ? D[d] * dy : (REAL)0;
                for (int n = 0; n < dstate; ++n) {
                    const size_t bn = ((size_t)b * dstate + n) * len + t;
                    const REAL a_next = (s + 1 < len) ? as[(size_t)(s + 1) * dstate + n] : (REAL)0;
                    g[n] = dy * Cm[bn] + a_next * g[n]; /* g_s = dy_s C_s + a_{s+1} g_{s+1} */
                    const REAL xprev = xs[(size_t)s * dstate + n];
                    const REAL a = as[(size_t)s * dstate + n];
                    const REAL An = A[(size_t)d * dstate + n];
                    if (dC) dC[bn] += dy * xs[(size_t)(s + 1) * dstate + n];
                    if (dB) dB[bn] += g[n] * dt * ut;
                    dut += dt * g[n] * Bm[bn];
                    ddt += g[n] * (Bm[bn] * ut + An * a * xprev);
                    dA_priv[((size_t)tid * dim + d) * dstate + n] += g[n] * dt * a * xprev;
                }
                if (du) du[row + t] = dut;
                REAL draw = ddt;
                if (delta_softplus) {
                    REAL raw = delta[row + t];
                    if (delta_bias) raw += delta_bias[d];
                    draw = raw > (REAL)20 ? ddt : ddt * FN(sigmoid_)(raw);
                }
                if (ddelta) ddelta[row + t] = draw;
                db_priv[(size_t)tid * dim + d] += draw;
            }
        }
        free(xs); free(as); free(dts);
    }
    for (int t = 0; t < nthreads; ++t) {
        if (dA) for (size_t i = 0; i < (size_t)dim * dstate; ++i) dA[i] += dA_priv[(size_t)t * dim * dstate + i];
        if (dD) for (int i = 0; i < dim; ++i) dD[i] += dD_priv[(size_t)t * dim + i];
        if (ddelta_bias) for (int i = 0; i < dim; ++i) ddelta_bias[i] += db_priv[(size_t)t * dim + i];
    }
    free(dA_priv); free(dD_priv); free(db_priv);
}

/*
 * Depthwise causal conv1d (+bias, optional SiLU): restates `act(conv1d(x)[..., :seqlen])`
 * (MS:272; nn.Conv1d groups=d_inner, padding=d_conv-1, MS:76-84), i.e.
 *   y[b,e,l] = act(bias[e] + sum_{w<W} weight[e,w] * x[b,e,l-(W-1)+w]),  x[...,<0] = 0.
 * reverse != 0 gives the anti-causal form obtained by conv(flip(x)) then flip (MS:229-241 on
 * xz.flip(-1)):  y[l] = act(bias + sum_w weight[w] * x[l+(W-1)-w]),  x[...,>=L] = 0.
 *   x, y : (batch, dim, len)    weight : (dim, W)    bias : (dim) or NULL
 */
void FN(aum_oracle_conv1d_fwd)(const REAL* x, const REAL* weight, const REAL* bias, int silu,
                               int reverse, int batch, int dim, int len, int width, REAL* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b)
        for (int e = 0; e < dim; ++e) {
            const size_t row = ((size_t)b * dim + e) * len;
            for (int l = 0; l < len; ++l) {
                REAL acc = bias ? bias[e] : (REAL)0;
                for (int w = 0; w < width; ++w) {
                    const int src = reverse ? l + (width - 1) - w : l - (width - 1) + w;
                    if (src >= 0 && src < len) acc += weight[(size_t)e * width + w] * x[row + src];
                }
                y[row + l] = silu ? acc * FN(sigmoid_)(acc) : acc;
            }
        }
}

/* Adjoint of the above: dx (batch,dim,len); dweight (dim,W), dbias (dim) accumulated into. */
void FN(aum_oracle_conv1d_bwd)(const REAL* x, const REAL* weight, const REAL* bias, const REAL* dy,
                               int silu, int reverse, int batch, int dim, int len, int width,
                               REAL* dx, REAL* dweight, REAL* dbias) {
    for (size_t i = 0; i < (size_t)batch * dim * len; ++i) dx[i] = 0;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < dim; ++e)
        for (int b = 0; b < batch; ++b) {
            const size_t row = ((size_t)b * dim + e) * len;
            for (int l = 0; l < len; ++l) {
                REAL acc = bias ? bias[e] : (REAL)0;
                for (int w = 0; w < width; ++w) {
                    const int src = reverse ? l + (width - 1) - w : l - (width - 1) + w;
                    if (src >= 0 && src < len) acc += weight[(size_t)e * width + w] * x[row + src];
                }
                REAL dpre = dy[row + l];
                if (silu) {
                    const REAL sg = FN(sigmoid_)(acc);
                    dpre *= sg * ((REAL)1 + acc * ((REAL)1 - sg));
                }
                if (dbias) dbias[e] += dpre;
                for (int w = 0; w < width; ++w) {
                    const int src = reverse ? l + (width - 1) - w : l - (width - 1) + w;
                    if (src >= 0 && src < len) {
                        dx[row + src] += weight[(size_t)e * width + w] * dpre;
                        if (dweight) dweight[(size_t)e * width + w] += x[row + src] * dpre;
                    }
                }
            }
        }
}

/*
 * Fused residual-add + RMSNorm forward: restates rms_norm_ref(upcast=True) (LN:35-48) with the
 * fused kernel's extra outputs (LN:86-120): res_out = x (+ residual), rstd = 1/sqrt(mean(res_out^2)+eps),
 * y = res_out * rstd * weight (+ bias).   x, residual, y, res_out : (rows, cols)
 */
void FN(aum_oracle_rmsnorm_fwd)(const REAL* x, const REAL* residual, const REAL* weight, const REAL* bias,
                                REAL eps, int rows, int cols, REAL* y, REAL* res_out, REAL* rstd) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const size_t o = (size_t)r * cols;
        REAL ss = 0;
        for (int c = 0; c < cols; ++c) {
            REAL v = x[o + c];
            if (residual) v += residual[o + c];
            if (res_out) res_out[o + c] = v;
            ss += v * v;
        }
        const REAL rs = (REAL)1 / (REAL)sqrt((double)(ss / cols + eps));
        if (rstd) rstd[r] = rs;
        for (int c = 0; c < cols; ++c) {
            REAL v = x[o + c];
            if (residual) v += residual[o + c];
            REAL out = v * rs * weight[c];
            if (bias) out += bias[c];
            y[o + c] = out;
        }
    }
}

/*
 * Backward (LN:254-277, rms branch): xhat = res_out*rstd, c = mean_j(xhat_j w_j dy_j),
 * dx = (w*dy - xhat*c)*rstd (+ dres_out), dweight += sum_rows dy*xhat, dbias += sum_rows dy.
 * dx doubles as the gradient for both `x` and `residual` (LN:372-375).
 */
void FN(aum_oracle_rmsnorm_bwd)(const REAL* dy, const REAL* dres_out, const REAL* res_out,
                                const REAL* weight, const REAL* rstd, int rows, int cols, REAL* dx,
                                REAL* dweight, REAL* dbias) {
    for (int r = 0; r < rows; ++r) {
        const size_t o = (size_t)r * cols;
        const REAL rs = rstd[r];
        REAL c1 = 0;
        for (int c = 0; c < cols; ++c) c1 += res_out[o + c] * rs * weight[c] * dy[o + c];
        c1 /= cols;
        for (int c = 0; c < cols; ++c) {
            const REAL xhat = res_out[o + c] * rs;
            REAL g = (weight[c] * dy[o + c] - xhat * c1) * rs;
            if (dres_out) g += dres_out[o + c];
            dx[o + c] = g;
            if (dweight) dweight[c] += dy[o + c] * xhat;
            if (dbias) dbias[c] += dy[o + c];
        }
    }
}

#undef FN
#undef CAT
#undef CAT_
