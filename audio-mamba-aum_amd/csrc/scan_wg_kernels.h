// scan_wg_kernels.h -- the production selective-scan kernels (dstate <= 16): 8-wave workgroups.
//
// Same per-wave algorithm as scan_kernels.h (lanes along time, K steps per lane, DPP/readlane associative
// scan, reverse direction as the suffix form, both directions fused in MODE 2) but organised so that the
// traffic that dominated the single-wave version never leaves the CU (measured on MI355X, profiles/):
//   * one workgroup = 8 waves = up to 64 channel rows of ONE batch element.  The B/C tile of the current
//     chunk ([N][64K] fp32) is loaded ONCE per workgroup with coalesced loads into LDS and read from there
//     by every (row pair, state) -- instead of every wave re-reading B/C through the vector L1;
//   * backward: dB/dC are accumulated across the workgroup's rows in LDS with ds_add_f32 and written as ONE
//     partial tile per workgroup; a second tiny kernel sums the partials.  No global fp32 atomics at all
//     (device-scope atomics execute at the memory side on this chip: the atomic version spent >95% of its
//     time there).  dA/dD/ddelta_bias leave the kernel as per-(batch,row) partials reduced by the same
//     second kernel.
// Workgroup phases are separated by s_barrier; see AUM_FOR_EACH_WAVE in wave.h.
#pragma once
#include "scan_kernels.h"

namespace aum {

constexpr int SCANWG_NW = 8;          // waves per workgroup
constexpr int SCANWG_MAX_N = 16;      // dstate limit of this path (LDS tile height)
constexpr int SCANWG_MAX_ROWS = 64;   // rows per workgroup (8 waves x 4 pairs x 2 rows)
// debug-only ablation bits (upper half of `flags`; set through AUM_ABLATE in the Python binding, never by the product)
constexpr uint32_t AUM_DBG_SKIP_STATES = 1u << 16, AUM_DBG_SKIP_LDS_ATOMICS = 1u << 17, AUM_DBG_SKIP_PARTIALS = 1u << 18,
                   AUM_DBG_SKIP_EPILOGUE = 1u << 19;
constexpr int SCANWG_MAX_K = 9;       // 64*9 = 576 steps per chunk keeps the backward's 4 tiles in 160 KB of LDS

template <int K> constexpr int scanwg_fwd_lds_floats() { return 2 * SCANWG_MAX_N * WAVE * K + 2 * SCANWG_MAX_ROWS * SCANWG_MAX_N; }
template <int K> constexpr int scanwg_bwd_lds_floats() { return 4 * SCANWG_MAX_N * WAVE * K + 3 * SCANWG_MAX_ROWS * SCANWG_MAX_N; }

// Workspace layout of the backward (floats), shared by host dispatch, kernel and the reduce kernel.
struct ScanWgWs {
    int64_t pB, pC, pA, pAb, pD, pbias, xck, total;
    int gpb, rows_per_wg, nchunks;
};
AUM_HOSTDEV ScanWgWs scanwg_ws_layout(int batch, int dim, int len, int N, int rows_per_wg, int nchunks, bool bidir) {
    ScanWgWs w;
    w.rows_per_wg = rows_per_wg;
    w.nchunks = nchunks;
    w.gpb = (dim + rows_per_wg - 1) / rows_per_wg;
    const int64_t tile = (int64_t)w.gpb * batch * N * len;
    int64_t o = 0;
    w.pB = o; o += tile;
    w.pC = o; o += tile;
    w.pA = o; o += (int64_t)batch * dim * N;
    w.pAb = o; o += bidir ? (int64_t)batch * dim * N : 0;
    w.pD = o; o += (int64_t)batch * dim;
    w.pbias = o; o += (int64_t)batch * dim;
    w.xck = o; o += nchunks > 1 ? (int64_t)batch * w.gpb * rows_per_wg * nchunks * N : 0;
    w.total = o;
    return w;
}

// Cooperative load of one [N][S] tile of B (or C) into LDS as fp32; t outside [0,len) -> 0.
template <class T, int K>
AUM_DEV void scanwg_load_tile(const T* src, int64_t n_stride, int N, int base, int len, float* tile, int w) {
    constexpr int S = WAVE * K;
    const vi lane = lane_id();
    for (int i0 = w * WAVE; i0 < N * S; i0 += SCANWG_NW * WAVE) {
        const vi idx = lane + i0;
        const vi n = idx / S;
        const vi tt = idx - n * S;
        const vi t = tt + base;
        const vm m = t < len;          // idx < N*S always: N*S is a multiple of 64
        lds_write(tile, idx, gload(src, n * (int)n_stride + t, m));
    }
}

template <class T, int K>
AUM_DEV void scanwg_load_rows(const void* u, int64_t u_bs, int64_t u_ds, const void* delta, int64_t d_bs, int64_t d_ds,
                              const float* delta_bias, bool softplus, int b, int e0, int dim, const vi (&t)[K],
                              const vm (&valid)[K], vf (&dl)[SCAN_R][K], vf (&dlu)[SCAN_R][K], vf (&sumd)[SCAN_R]) {
    AUM_UNROLL
    for (int r = 0; r < SCAN_R; ++r) {
        const int e = e0 + r;
        const bool rowok = e < dim;
        const int ec = rowok ? e : dim - 1;
        const T* up = row_ptr<T>(u, (int64_t)b * u_bs + (int64_t)ec * u_ds);
        const T* dp = row_ptr<T>(delta, (int64_t)b * d_bs + (int64_t)ec * d_ds);
        const float bias = delta_bias ? delta_bias[ec] : 0.f;
        vf sd = splat(0.f);
        AUM_UNROLL
        for (int k = 0; k < K; ++k) {
            const vm m = valid[k] && rowok;
            const vf uu = gload(up, t[k], m);
            vf d = gload(dp, t[k], m) + bias;
            if (softplus) d = vsoftplus(d);
            d = vsel(m, d, splat(0.f));
            dl[r][k] = d;
            dlu[r][k] = d * uu;
            sd = sd + d;
        }
        sumd[r] = sd;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <class T, int K, bool REV>
AUM_DEV void scanwg_fwd_dir(const AumScanFwdArgs& p, int b, int e0, int rloc, const float* Aptr, const float* Bt,
                            const float* Ct, const vf (&dl)[SCAN_R][K], const vf (&dlu)[SCAN_R][K],
                            const vf (&sumd)[SCAN_R], float* carry, bool multi, bool write_last, vf (&y)[SCAN_R][K]) {
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    const vi lane = lane_id();
    for (int n = 0; n < N; ++n) {
        vf Bn[K], Cn[K];
        AUM_UNROLL
        for (int k = 0; k < K; ++k) {
            Bn[k] = lds_read(Bt, lane * K + (n * S + k));
            Cn[k] = lds_read(Ct, lane * K + (n * S + k));
        }
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            const int e = e0 + r;
            if (e < p.dim) {
                const float An = Aptr[(int64_t)e * N + n] * LOG2E;
                vf a[K], bb[K], x[K];
                AUM_UNROLL
                for (int k = 0; k < K; ++k) {
                    a[k] = vexp2(dl[r][k] * An);
                    bb[k] = dlu[r][k] * Bn[k];
                }
                const vf Ptot = vexp2(sumd[r] * An);
                vf cin = splat(0.f);
                if (multi) cin = lds_read(carry, spl_i((rloc + r) * SCANWG_MAX_N + n));
                vf xin, cout;
                affine_scan_states<K, REV>(a, bb, Ptot, cin, x, xin, cout);
                AUM_UNROLL
                for (int k = 0; k < K; ++k) y[r][k] = vfma(Cn[k], x[k], y[r][k]);
                if (multi) lds_write(carry, spl_i((rloc + r) * SCANWG_MAX_N + n), cout);
                if (write_last) gstore(p.last_state + ((int64_t)b * p.dim + e) * N + n, spl_i(0), cout, lane == 0);
            }
        }
    }
}

template <class T, int K, int MODE>
AUM_DEV void scanwg_fwd(const AumScanFwdArgs& p, int wg, float* lds, int rows_per_wg) {
    constexpr bool BI = MODE == 2;
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + SCANWG_MAX_N * S;
    float* carry = lds + 2 * SCANWG_MAX_N * S;     // [2][SCANWG_MAX_ROWS][SCANWG_MAX_N]
    const int gpb = (p.dim + rows_per_wg - 1) / rows_per_wg;
    const int b = wg / gpb;
    const int eb = (wg % gpb) * rows_per_wg;
    const int nchunks = (p.len + S - 1) / S;
    const bool multi = nchunks > 1;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);

    if (multi) {
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            for (int i0 = w * WAVE; i0 < 2 * SCANWG_MAX_ROWS * SCANWG_MAX_N; i0 += SCANWG_NW * WAVE)
                lds_write(carry, lane_id() + i0, splat(0.f));
        }
        AUM_WG_BARRIER();
    }
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = (MODE == 1) ? nchunks - 1 - ci : ci;
        const int base = c * S;
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            scanwg_load_tile<T, K>(Bsrc, p.B_ns, N, base, p.len, Bt, w);
            scanwg_load_tile<T, K>(Csrc, p.C_ns, N, base, p.len, Ct, w);
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            const vi lane = lane_id();
            for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                const int rloc = 2 * pair;
                const int e0 = eb + rloc;
                if (e0 >= p.dim) break;
                vi t[K];
                vm valid[K];
                AUM_UNROLL
                for (int k = 0; k < K; ++k) {
                    t[k] = lane * K + (base + k);
                    valid[k] = t[k] < p.len;
                }
                vf dl[SCAN_R][K], dlu[SCAN_R][K], y[SCAN_R][K], sumd[SCAN_R];
                scanwg_load_rows<T, K>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                       e0, p.dim, t, valid, dl, dlu, sumd);
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    AUM_UNROLL
                    for (int k = 0; k < K; ++k) y[r][k] = splat(0.f);
                }
                const bool wl = (ci == nchunks - 1) && p.last_state != nullptr;
                if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                if (MODE == 0 || BI)
                    scanwg_fwd_dir<T, K, false>(p, b, e0, rloc, p.A, Bt, Ct, dl, dlu, sumd, carry, multi, wl && !BI, y);
                if (MODE == 1)
                    scanwg_fwd_dir<T, K, true>(p, b, e0, rloc, p.A, Bt, Ct, dl, dlu, sumd, carry, multi, wl, y);
                if (BI)
                    scanwg_fwd_dir<T, K, true>(p, b, e0, rloc, p.A_b, Bt, Ct, dl, dlu, sumd,
                                               carry + SCANWG_MAX_ROWS * SCANWG_MAX_N, multi, false, y);
                }
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    const int e = e0 + r;
                    if (e < p.dim) {
                        const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds);
                        const float Dn = p.D ? (BI ? 2.f : 1.f) * p.D[e] : 0.f;
                        const int64_t ooff = (int64_t)b * p.out_bs + (int64_t)e * p.out_ds;
                        AUM_UNROLL
                        for (int k = 0; k < K; ++k) {
                            vf o = y[r][k];
                            if (p.D) o = vfma(gload(up, t[k], valid[k]), splat(Dn), o);
                            if (p.out_pre) gstore(row_ptr_w<T>(p.out_pre, ooff), t[k], o, valid[k]);
                            if (p.z) {
                                const T* zp = row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)e * p.z_ds);
                                const vf zz = gload(zp, t[k], valid[k]);
                                o = o * (zz * vsigmoid(zz));
                            }
                            gstore(row_ptr_w<T>(p.out, ooff), t[k], o, valid[k]);
                        }
                    }
                }
            }
        }
        AUM_WG_BARRIER();
    }
}

// ------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------
template <class T, int K, bool REV>
AUM_DEV void scanwg_bwd_dir_state(const AumScanBwdArgs& p, int n, int b, int e0, int rloc, const float* Aptr,
                                  float* pA, const vf (&Bn)[K], const vf (&Cn)[K], const vf (&dl)[SCAN_R][K],
                                  const vf (&dlu)[SCAN_R][K], const vf (&dy)[SCAN_R][K], const vf (&sumd)[SCAN_R],
                                  const vf (&sumd_next)[SCAN_R], const vf (&dnf)[SCAN_R], const float* xck,
                                  int64_t xck_row_stride, int chunk, bool first_visit, float* gcarry, bool multi,
                                  vf (&G)[SCAN_R][K], vf (&DA)[SCAN_R][K], vf (&dBacc)[K], vf (&dCacc)[K]) {
    const int N = p.dstate;
    AUM_UNROLL
    for (int r = 0; r < SCAN_R; ++r) {
        const int e = e0 + r;
        if (e < p.dim) {
            const float Araw = Aptr[(int64_t)e * N + n];
            const float An = Araw * LOG2E;
            vf a[K], bb[K], x[K];
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                a[k] = vexp2(dl[r][k] * An);
                bb[k] = dlu[r][k] * Bn[k];
            }
            const vf Ptot = vexp2(sumd[r] * An);
            vf cin = splat(0.f);
            if (multi) cin = gload_coherent(xck + (rloc + r) * xck_row_stride + (int64_t)chunk * N + n, spl_i(0), lane_id() >= 0);
            vf xin, cout;
            affine_scan_states<K, REV>(a, bb, Ptot, cin, x, xin, cout);
            vf an[K], cc[K], g[K];
            const vf a_nf = vexp2(dnf[r] * An);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                cc[k] = dy[r][k] * Cn[k];
                if (!REV) an[k] = (k + 1 < K) ? a[k + 1 < K ? k + 1 : 0] : dpp_wave_shl1(a[0], a_nf);
                else      an[k] = (k > 0) ? a[k > 0 ? k - 1 : 0] : dpp_wave_shr1(a[K - 1], a_nf);
            }
            const vf Pn = vexp2(sumd_next[r] * An);
            vf gin_c = splat(0.f);
            if (multi) gin_c = lds_read(gcarry, spl_i((rloc + r) * SCANWG_MAX_N + n));
            vf gin, gout;
            affine_scan_states<K, !REV>(an, cc, Pn, gin_c, g, gin, gout);
            if (multi) lds_write(gcarry, spl_i((rloc + r) * SCANWG_MAX_N + n), gout);
            vf dAl = splat(0.f);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                const vf xprev = REV ? ((k == K - 1) ? xin : x[k + 1 < K ? k + 1 : 0]) : ((k == 0) ? xin : x[k > 0 ? k - 1 : 0]);
                const vf h = g[k] * a[k] * xprev;
                G[r][k] = vfma(g[k], Bn[k], G[r][k]);
                DA[r][k] = vfma(splat(Araw), h, DA[r][k]);
                dBacc[k] = vfma(g[k], dlu[r][k], dBacc[k]);
                dCacc[k] = vfma(dy[r][k], x[k], dCacc[k]);
                dAl = vfma(dl[r][k], h, dAl);
            }
            // per-(batch,row,state) partial: this wave is the only writer; chunks after the first accumulate
            if (!(p.flags & AUM_DBG_SKIP_PARTIALS)) {
            float dAsum = wave_sum(dAl);
            float* slot = pA + ((int64_t)b * p.dim + e) * N + n;
            if (!first_visit) dAsum += readlane(gload_coherent(slot, spl_i(0), lane_id() >= 0), 0);
            gstore_coherent(slot, spl_i(0), splat(dAsum), lane_id() == 0);
            }
        }
    }
}

template <class T, int K, bool REV>
AUM_DEV void scanwg_bwd_prepass_pair(const AumScanBwdArgs& p, int b, int e0, int rloc, int c, int base, const float* Aptr,
                                     const float* Bt, float* xck, int64_t xck_row_stride, float* xcarry) {
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();
    vi t[K];
    vm valid[K];
    AUM_UNROLL
    for (int k = 0; k < K; ++k) { t[k] = lane * K + (base + k); valid[k] = t[k] < p.len; }
    vf dl[SCAN_R][K], dlu[SCAN_R][K], sumd[SCAN_R];
    scanwg_load_rows<T, K>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b, e0, p.dim, t,
                           valid, dl, dlu, sumd);
    for (int n = 0; n < N; ++n) {
        vf Bn[K];
        AUM_UNROLL
        for (int k = 0; k < K; ++k) Bn[k] = lds_read(Bt, lane * K + (n * S + k));
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            if (e0 + r < p.dim) {
                const float An = Aptr[(int64_t)(e0 + r) * N + n] * LOG2E;
                vf a[K], bb[K], x[K];
                AUM_UNROLL
                for (int k = 0; k < K; ++k) { a[k] = vexp2(dl[r][k] * An); bb[k] = dlu[r][k] * Bn[k]; }
                const vf cin = lds_read(xcarry, spl_i((rloc + r) * SCANWG_MAX_N + n));
                gstore_coherent(xck + (rloc + r) * xck_row_stride + (int64_t)c * N + n, spl_i(0), cin, lane == 0);
                vf xin, cout;
                affine_scan_states<K, REV>(a, bb, vexp2(sumd[r] * An), cin, x, xin, cout);
                lds_write(xcarry, spl_i((rloc + r) * SCANWG_MAX_N + n), cout);
            }
        }
    }
}

template <class T, int K, int MODE>
AUM_DEV void scanwg_bwd(const AumScanBwdArgs& p, int wg, float* lds, int rows_per_wg) {
    constexpr bool BI = MODE == 2;
    constexpr bool REV0 = MODE == 1;
    constexpr int S = WAVE * K;
    constexpr int TILE = SCANWG_MAX_N * S;
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + TILE;
    float* dBt = lds + 2 * TILE;
    float* dCt = lds + 3 * TILE;
    float* xcarry = lds + 4 * TILE;                                  // [rows][N]   (multi-chunk pre-pass)
    float* gcarry = xcarry + SCANWG_MAX_ROWS * SCANWG_MAX_N;          // [2][rows][N]
    const int nchunks = (p.len + S - 1) / S;
    const bool multi = nchunks > 1;
    const ScanWgWs L = scanwg_ws_layout(p.batch, p.dim, p.len, N, rows_per_wg, nchunks, BI);
    float* ws = (float*)p.workspace;
    const int gpb = L.gpb;
    const int b = wg / gpb;
    const int g_idx = wg % gpb;
    const int eb = g_idx * rows_per_wg;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const float ndir = BI ? 2.f : 1.f;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);
    const int64_t xck_row_stride = (int64_t)nchunks * N;
    float* xck = multi ? ws + L.xck + (int64_t)wg * rows_per_wg * xck_row_stride : nullptr;

    if (multi) {
        // pre-pass: states entering every chunk, in scan order
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            for (int i0 = w * WAVE; i0 < 3 * SCANWG_MAX_ROWS * SCANWG_MAX_N; i0 += SCANWG_NW * WAVE)
                lds_write(xcarry, lane_id() + i0, splat(0.f));
        }
        AUM_WG_BARRIER();
        for (int ci = 0; ci < nchunks; ++ci) {
            const int c = REV0 ? nchunks - 1 - ci : ci;
            AUM_FOR_EACH_WAVE(w, SCANWG_NW) { scanwg_load_tile<T, K>(Bsrc, p.B_ns, N, c * S, p.len, Bt, w); }
            AUM_WG_BARRIER();
            AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
                for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                    if (eb + 2 * pair >= p.dim) break;
                    scanwg_bwd_prepass_pair<T, K, REV0>(p, b, eb + 2 * pair, 2 * pair, c, c * S, p.A, Bt, xck, xck_row_stride,
                                                        xcarry);
                }
            }
            AUM_WG_BARRIER();
        }
    }

    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = REV0 ? ci : nchunks - 1 - ci;     // adjoint order = opposite of the scan order
        const int base = c * S;
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            scanwg_load_tile<T, K>(Bsrc, p.B_ns, N, base, p.len, Bt, w);
            scanwg_load_tile<T, K>(Csrc, p.C_ns, N, base, p.len, Ct, w);
            for (int i0 = w * WAVE; i0 < N * S; i0 += SCANWG_NW * WAVE) {
                lds_write(dBt, lane_id() + i0, splat(0.f));
                lds_write(dCt, lane_id() + i0, splat(0.f));
            }
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            const vi lane = lane_id();
            for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                const int rloc = 2 * pair;
                const int e0 = eb + rloc;
                const bool active = e0 < p.dim;      // wave-uniform; inactive waves still take every barrier below
                vi t[K];
                vm valid[K];
                AUM_UNROLL
                for (int k = 0; k < K; ++k) { t[k] = lane * K + (base + k); valid[k] = t[k] < p.len; }
                vf dl[SCAN_R][K], dlu[SCAN_R][K], dy[SCAN_R][K], G[SCAN_R][K], DA[SCAN_R][K];
                vf sumd[SCAN_R], sumd_nf[SCAN_R], sumd_nr[SCAN_R], dnf_f[SCAN_R], dnf_r[SCAN_R];
                scanwg_load_rows<T, K>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b, e0,
                                       p.dim, t, valid, dl, dlu, sumd);      // rows >= dim load nothing (masked)
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    const int e = e0 + r;
                    const bool rowok = e < p.dim;
                    const int ec = rowok ? e : p.dim - 1;
                    const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
                    const T* gp = row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds);
                    const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
                    AUM_UNROLL
                    for (int k = 0; k < K; ++k) {
                        const vm m = valid[k] && rowok;
                        vf go = gload(gp, t[k], m);
                        if (p.z) {
                            const T* zp = row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)ec * p.z_ds);
                            const T* op = row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)ec * p.out_ds);
                            const vf zz = gload(zp, t[k], m);
                            const vf yp = gload(op, t[k], m);
                            const vf sg = vsigmoid(zz);
                            const vf dzv = go * yp * sg * vfma(zz, splat(1.f) - sg, splat(1.f));
                            gstore(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)ec * p.dz_ds), t[k], dzv, m);
                            go = go * zz * sg;
                        }
                        dy[r][k] = go;
                        G[r][k] = splat(0.f);
                        DA[r][k] = splat(0.f);
                    }
                    const int tf = base + S, tr = base - 1;
                    vf df = splat(0.f), dr = splat(0.f);
                    if (multi && rowok) {
                        if (tf < p.len) {
                            df = gload(dp, spl_i(tf), lane >= 0) + bias;
                            if (softplus) df = vsoftplus(df);
                        }
                        if (tr >= 0) {
                            dr = gload(dp, spl_i(tr), lane >= 0) + bias;
                            if (softplus) dr = vsoftplus(dr);
                        }
                    }
                    dnf_f[r] = df;
                    dnf_r[r] = dr;
                    sumd_nf[r] = sumd[r] - dl[r][0] + dpp_wave_shl1(dl[r][0], df);
                    sumd_nr[r] = sumd[r] - dl[r][K - 1] + dpp_wave_shr1(dl[r][K - 1], dr);
                }
                const bool first_visit = ci == 0;
                // Rotated state order: at step j wave w works on state (j + 2w) mod 16, so no two of the 8 waves ever
                // hold the same state row of the dB/dC tiles within a step (nor in adjacent steps); with one barrier
                // per step the tile update below is a plain LDS read-add-write -- ds_add_f32 measured ~125 cycles
                // per wave-instruction here and was 57% of the kernel.
                for (int j = 0; j < SCANWG_MAX_N; ++j) {
                    const int n = (j + 2 * w) & (SCANWG_MAX_N - 1);
                    if (active && n < N) {
                    vf Bn[K], Cn[K], dBacc[K], dCacc[K];
                    AUM_UNROLL
                    for (int k = 0; k < K; ++k) {
                        Bn[k] = lds_read(Bt, lane * K + (n * S + k));
                        Cn[k] = lds_read(Ct, lane * K + (n * S + k));
                        dBacc[k] = splat(0.f);
                        dCacc[k] = splat(0.f);
                    }
                    if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                    if (MODE == 0 || BI)
                        scanwg_bwd_dir_state<T, K, false>(p, n, b, e0, rloc, p.A, ws + L.pA, Bn, Cn, dl, dlu, dy, sumd, sumd_nf,
                                                          dnf_f, xck, xck_row_stride, c, first_visit, gcarry, multi, G, DA,
                                                          dBacc, dCacc);
                    if (MODE == 1)
                        scanwg_bwd_dir_state<T, K, true>(p, n, b, e0, rloc, p.A, ws + L.pA, Bn, Cn, dl, dlu, dy, sumd, sumd_nr,
                                                         dnf_r, xck, xck_row_stride, c, first_visit, gcarry, multi, G, DA,
                                                         dBacc, dCacc);
                    if (BI)
                        scanwg_bwd_dir_state<T, K, true>(p, n, b, e0, rloc, p.A_b, ws + L.pAb, Bn, Cn, dl, dlu, dy, sumd,
                                                         sumd_nr, dnf_r, nullptr, 0, c, first_visit,
                                                         gcarry + SCANWG_MAX_ROWS * SCANWG_MAX_N, false, G, DA, dBacc, dCacc);
                    }
                    if (!(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                    AUM_UNROLL
                    for (int k = 0; k < K; ++k) {
                        const vi at = lane * K + (n * S + k);
                        lds_write(dBt, at, lds_read(dBt, at) + dBacc[k]);
                        lds_write(dCt, at, lds_read(dCt, at) + dCacc[k]);
                    }
                    }
                    }
                    AUM_WG_BARRIER_IN_PHASE();
                }
                if (active && !(p.flags & AUM_DBG_SKIP_EPILOGUE)) {
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    const int e = e0 + r;
                    if (e < p.dim) {
                        const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds);
                        const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)e * p.delta_ds);
                        const float bias = p.delta_bias ? p.delta_bias[e] : 0.f;
                        const float Dn = p.D ? ndir * p.D[e] : 0.f;
                        vf dDl = splat(0.f), dbl = splat(0.f);
                        AUM_UNROLL
                        for (int k = 0; k < K; ++k) {
                            const vf uu = gload(up, t[k], valid[k]);
                            const vf duv = vfma(dl[r][k], G[r][k], dy[r][k] * Dn);
                            vf dd = vfma(uu, G[r][k], DA[r][k]);
                            if (softplus) {
                                const vf raw = gload(dp, t[k], valid[k]) + bias;
                                dd = vsel(raw > 20.f, dd, dd * vsigmoid(raw));
                            }
                            dd = vsel(valid[k], dd, splat(0.f));
                            gstore(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), t[k], duv, valid[k]);
                            gstore(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds), t[k], dd,
                                   valid[k]);
                            dDl = vfma(dy[r][k], uu, dDl);
                            dbl = dbl + dd;
                        }
                        float sD = ndir * wave_sum(dDl), sb = wave_sum(dbl);
                        float* slotD = ws + L.pD + (int64_t)b * p.dim + e;
                        float* slotb = ws + L.pbias + (int64_t)b * p.dim + e;
                        if (!first_visit) {
                            sD += readlane(gload_coherent(slotD, spl_i(0), lane >= 0), 0);
                            sb += readlane(gload_coherent(slotb, spl_i(0), lane >= 0), 0);
                        }
                        gstore_coherent(slotD, spl_i(0), splat(sD), lane == 0);
                        gstore_coherent(slotb, spl_i(0), splat(sb), lane == 0);
                    }
                }
                }
            }
        }
        AUM_WG_BARRIER();
        // flush this chunk's dB/dC tile: one partial per workgroup, every (g,b,n,t) written exactly once
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            const vi lane = lane_id();
            float* oB = ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len;
            float* oC = ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len;
            for (int i0 = w * WAVE; i0 < N * S; i0 += SCANWG_NW * WAVE) {
                const vi idx = lane + i0;
                const vi n = idx / S;
                const vi tt = idx - n * S;
                const vi tg = tt + base;
                const vm m = tg < p.len;
                gstore(oB, n * p.len + tg, lds_read(dBt, idx), m);
                gstore(oC, n * p.len + tg, lds_read(dCt, idx), m);
            }
        }
        AUM_WG_BARRIER();
    }
}

// Second stage of the backward: dst[i] += sum_o src[o*inner + i] for up to 6 segments.
struct ScanReduceSeg { float* dst; const float* src; int64_t inner; int32_t outer; int32_t pad; };
struct ScanReduceArgs { ScanReduceSeg seg[6]; int32_t nseg; };

}  // namespace aum
