// scan_kernels.h -- selective scan forward / backward for one wavefront (= one workgroup).
//
// Replaces selective_scan_cuda.fwd / .bwd (call sites SSI:37, 62-65, 499-505, 541-552; arithmetic
// SSI:86-152 and SURVEY.md 8a').  Design (DESIGN.md "scan"):
//   * one wave owns R channel rows (same batch b, consecutive e) -> B/C registers are loaded once per
//     state n and reused by the R rows;
//   * lanes run along TIME: lane i owns the K consecutive steps [c*64K + iK, +K) of chunk c;
//   * per (row, state): a local K-step recurrence, one 64-lane associative scan of the affine maps
//     (DPP row_shr/row_shl + v_readlane, wave.h), and a second local pass that materialises x_t;
//   * the time-reversed direction (A_b) is the same data scanned with the suffix form -- no flip copies;
//     MODE 2 fuses both directions over one read of u/delta/z/B/C (single chunk, len <= 64K);
//   * backward = forward recompute + the adjoint recurrence g_t = dy_t C_t + a_{t+1} g_{t+1} as the
//     opposite-direction scan; dB/dC are accumulated in registers over the R rows and both
//     directions, then added to global memory with fp32 atomics; dA/dD/dbias are lane-reduced (DPP).
#pragma once
#include "../../include/aum_hip.h"
#include "wave.h"

namespace aum {

constexpr int SCAN_MAX_DSTATE = 256;
constexpr int SCAN_R = 2;                       // rows per wave
constexpr int SCAN_LDS_FLOATS = 2 * SCAN_R * SCAN_MAX_DSTATE;   // carries: [dir][row][state]

// x'[k] = a[k]*x + b[k] over the wave.  Order of steps: lane-major, k ascending (REV=false) or the exact
// mirror (REV=true).  carry_in/carry_out are wave-uniform (all lanes equal).  x[k] = state after step k,
// x_in = state entering this lane's first step.
// Multiplier of slot k is given by an accessor so that the adjoint pass can use "a of the scan successor" without
// materialising a shifted copy of the array.
template <int K, bool REV, class AF>
AUM_DEV void affine_scan_states_f(AF a, const vf (&b)[K], vf Ptot, vf carry_in, vf (&x)[K], vf& x_in, vf& carry_out) {
    vf s = splat(0.f);
    AUM_UNROLL
    for (int kk = 0; kk < K; ++kk) {
        const int k = REV ? K - 1 - kk : kk;
        s = vfma(a(k), s, b[k]);
    }
    const vm first = lane_id() == (REV ? WAVE - 1 : 0);
    vf S = vsel(first, vfma(Ptot, carry_in, s), s);
    vf P = Ptot;
    wave_scan_affine<REV>(P, S);
    x_in = REV ? dpp_wave_shl1(S, carry_in) : dpp_wave_shr1(S, carry_in);
    carry_out = splat(readlane(S, REV ? 0 : WAVE - 1));
    vf xx = x_in;
    AUM_UNROLL
    for (int kk = 0; kk < K; ++kk) {
        const int k = REV ? K - 1 - kk : kk;
        xx = vfma(a(k), xx, b[k]);
        x[k] = xx;
    }
}
template <int K, bool REV>
AUM_DEV void affine_scan_states(const vf (&a)[K], const vf (&b)[K], vf Ptot, vf carry_in, vf (&x)[K], vf& x_in,
                                vf& carry_out) {
    affine_scan_states_f<K, REV>([&](int k) -> const vf& { return a[k]; }, b, Ptot, carry_in, x, x_in, carry_out);
}

template <class T> AUM_DEV const T* row_ptr(const void* base, int64_t off) { return (const T*)base + off; }
template <class T> AUM_DEV T* row_ptr_w(void* base, int64_t off) { return (T*)base + off; }

// ------------------------------------------------------------------------------------------------
// Forward.  MODE 0: forward time; MODE 1: reverse time (AUM_SCAN_REVERSE); MODE 2: both, fused.
// ------------------------------------------------------------------------------------------------
template <class T, int K, bool REV>
AUM_DEV void scan_fwd_dir(const AumScanFwdArgs& p, int b, int e0, const vi (&t)[K], const vm (&valid)[K],
                          const float* Aptr, const vf (&dl)[SCAN_R][K], const vf (&dlu)[SCAN_R][K],
                          const vf (&sumd)[SCAN_R], float* lds_carry, vf (&y)[SCAN_R][K]) {
    const int N = p.dstate;
    for (int n = 0; n < N; ++n) {
        vf Bn[K], Cn[K];
        const T* Bp = row_ptr<T>(p.B, (int64_t)b * p.B_bs + (int64_t)n * p.B_ns);
        const T* Cp = row_ptr<T>(p.C, (int64_t)b * p.C_bs + (int64_t)n * p.C_ns);
        AUM_UNROLL
        for (int k = 0; k < K; ++k) {
            Bn[k] = gload(Bp, t[k], valid[k]);
            Cn[k] = gload(Cp, t[k], valid[k]);
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
                const vf cin = lds_read(lds_carry, spl_i(r * N + n));
                vf xin, cout;
                affine_scan_states<K, REV>(a, bb, Ptot, cin, x, xin, cout);
                AUM_UNROLL
                for (int k = 0; k < K; ++k) y[r][k] = vfma(Cn[k], x[k], y[r][k]);
                lds_write(lds_carry, spl_i(r * N + n), cout);
            }
        }
    }
}

template <class T, int K, int MODE>
AUM_DEV void scan_fwd_wave(const AumScanFwdArgs& p, int wg, float* lds) {
    constexpr bool BI = MODE == 2;
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    const int gpb = (p.dim + SCAN_R - 1) / SCAN_R;
    const int b = wg / gpb;
    const int e0 = (wg % gpb) * SCAN_R;
    const int nchunks = (p.len + S - 1) / S;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();

    for (int j = 0; j < SCAN_LDS_FLOATS / WAVE; ++j) lds_write(lds, lane + j * WAVE, splat(0.f));
    wave_sync();

    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = (MODE == 1) ? nchunks - 1 - ci : ci;
        const int base = c * S;
        vi t[K];
        vm valid[K];
        AUM_UNROLL
        for (int k = 0; k < K; ++k) {
            t[k] = lane * K + (base + k);
            valid[k] = t[k] < p.len;
        }
        vf dl[SCAN_R][K], dlu[SCAN_R][K], y[SCAN_R][K], sumd[SCAN_R];
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            const int e = e0 + r;
            const bool rowok = e < p.dim;
            const int ec = rowok ? e : p.dim - 1;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
            const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
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
                y[r][k] = splat(0.f);
            }
            sumd[r] = sd;
        }
        const bool last_chunk = ci == nchunks - 1;
        if (MODE == 0 || BI)
            scan_fwd_dir<T, K, false>(p, b, e0, t, valid, p.A, dl, dlu, sumd, lds, y);
        if (MODE == 1)
            scan_fwd_dir<T, K, true>(p, b, e0, t, valid, p.A, dl, dlu, sumd, lds, y);
        if (BI)
            scan_fwd_dir<T, K, true>(p, b, e0, t, valid, p.A_b, dl, dlu, sumd, lds + SCAN_R * SCAN_MAX_DSTATE, y);
        wave_sync();
        // epilogue: out = (y + ndir*D*u) * silu(z)
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
                if (last_chunk && p.last_state && !BI) {
                    // state after the final step = the carry left in LDS by the last chunk
                    for (int n0 = 0; n0 < N; n0 += WAVE) {
                        const vi n = lane + n0;
                        const vm m = n < N;
                        const vf v = lds_read(lds, vsel_i(m, n, spl_i(0)) + r * N);
                        gstore(p.last_state + ((int64_t)b * p.dim + e) * N, n, v, m);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward.
// ------------------------------------------------------------------------------------------------
// One direction, one state n, all R rows: forward recompute, adjoint scan, gradient accumulation.
//   ws_x : per-(row,dir) chunk-entry states written by the pre-pass (multi-chunk) or nullptr
template <class T, int K, bool REV>
AUM_DEV void scan_bwd_dir_state(const AumScanBwdArgs& p, int n, int e0, const float* Aptr, float* dAptr,
                                const vf (&Bn)[K], const vf (&Cn)[K], const vf (&dl)[SCAN_R][K],
                                const vf (&dlu)[SCAN_R][K], const vf (&dy)[SCAN_R][K], const vf (&sumd)[SCAN_R],
                                const vf (&sumd_next)[SCAN_R], const vf (&dnf)[SCAN_R], const float* ws_x,
                                int64_t ws_stride_row, int chunk, int nchunks, float* lds_gcarry,
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
            if (ws_x) cin = gload_coherent(ws_x + r * ws_stride_row + (int64_t)chunk * N + n, spl_i(0), lane_id() >= 0);
            vf xin, cout;
            affine_scan_states<K, REV>(a, bb, Ptot, cin, x, xin, cout);
            // adjoint: g_s = dy_s*C_s + a_{s+1} * g_{s+1}, scanned against the recurrence direction
            vf an[K], cc[K], g[K];
            const vf a_nf = vexp2(dnf[r] * An);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                cc[k] = dy[r][k] * Cn[k];
                if (!REV) an[k] = (k + 1 < K) ? a[k + 1 < K ? k + 1 : 0] : dpp_wave_shl1(a[0], a_nf);
                else      an[k] = (k > 0) ? a[k > 0 ? k - 1 : 0] : dpp_wave_shr1(a[K - 1], a_nf);
            }
            const vf Pn = vexp2(sumd_next[r] * An);
            const vf gin_c = lds_read(lds_gcarry, spl_i(r * N + n));
            vf gin, gout;
            affine_scan_states<K, !REV>(an, cc, Pn, gin_c, g, gin, gout);
            lds_write(lds_gcarry, spl_i(r * N + n), gout);
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
            const float dAsum = wave_sum(dAl);
            gatomic_add(dAptr + (int64_t)e * N + n, spl_i(0), splat(dAsum), lane_id() == 0);
        }
    }
    (void)nchunks;
}

// Pre-pass for multi-chunk rows: states entering every chunk, in scan order (one direction).
template <class T, int K, bool REV>
AUM_DEV void scan_bwd_prepass(const AumScanBwdArgs& p, int b, int e0, const float* Aptr, float* ws_x,
                              int64_t ws_stride_row, int nchunks, float* lds) {
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();
    for (int j = 0; j < SCAN_LDS_FLOATS / WAVE; ++j) lds_write(lds, lane + j * WAVE, splat(0.f));
    wave_sync();
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = REV ? nchunks - 1 - ci : ci;
        const int base = c * S;
        vi t[K];
        vm valid[K];
        AUM_UNROLL
        for (int k = 0; k < K; ++k) { t[k] = lane * K + (base + k); valid[k] = t[k] < p.len; }
        vf dl[SCAN_R][K], dlu[SCAN_R][K], sumd[SCAN_R];
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            const int e = e0 + r;
            const bool rowok = e < p.dim;
            const int ec = rowok ? e : p.dim - 1;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
            const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
            vf sd = splat(0.f);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                const vm m = valid[k] && rowok;
                const vf uu = gload(up, t[k], m);
                vf d = gload(dp, t[k], m) + bias;
                if (softplus) d = vsoftplus(d);
                d = vsel(m, d, splat(0.f));
                dl[r][k] = d; dlu[r][k] = d * uu; sd = sd + d;
            }
            sumd[r] = sd;
        }
        for (int n = 0; n < N; ++n) {
            vf Bn[K];
            const T* Bp = row_ptr<T>(p.B, (int64_t)b * p.B_bs + (int64_t)n * p.B_ns);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) Bn[k] = gload(Bp, t[k], valid[k]);
            AUM_UNROLL
            for (int r = 0; r < SCAN_R; ++r) {
                if (e0 + r < p.dim) {
                    const float An = Aptr[(int64_t)(e0 + r) * N + n] * LOG2E;
                    vf a[K], bb[K], x[K];
                    AUM_UNROLL
                    for (int k = 0; k < K; ++k) { a[k] = vexp2(dl[r][k] * An); bb[k] = dlu[r][k] * Bn[k]; }
                    const vf cin = lds_read(lds, spl_i(r * N + n));
                    gstore_coherent(ws_x + r * ws_stride_row + (int64_t)c * N + n, spl_i(0), cin, lane == 0);
                    vf xin, cout;
                    affine_scan_states<K, REV>(a, bb, vexp2(sumd[r] * An), cin, x, xin, cout);
                    lds_write(lds, spl_i(r * N + n), cout);
                }
            }
        }
        wave_sync();
    }
}

template <class T, int K, int MODE>
AUM_DEV void scan_bwd_wave(const AumScanBwdArgs& p, int wg, float* lds) {
    constexpr bool BI = MODE == 2;
    constexpr bool REV0 = MODE == 1;   // scan direction of the (first) pass
    constexpr int S = WAVE * K;
    const int N = p.dstate;
    const int gpb = (p.dim + SCAN_R - 1) / SCAN_R;
    const int b = wg / gpb;
    const int e0 = (wg % gpb) * SCAN_R;
    const int nchunks = (p.len + S - 1) / S;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();
    const float ndir = BI ? 2.f : 1.f;

    // multi-chunk (unidirectional only): chunk-entry states -> workspace [wg][row][chunk][n]
    float* ws_x = nullptr;
    const int64_t ws_stride_row = (int64_t)nchunks * N;
    if (nchunks > 1) {
        ws_x = (float*)p.workspace + (int64_t)wg * SCAN_R * ws_stride_row;
        scan_bwd_prepass<T, K, REV0>(p, b, e0, p.A, ws_x, ws_stride_row, nchunks, lds);
    }
    for (int j = 0; j < SCAN_LDS_FLOATS / WAVE; ++j) lds_write(lds, lane + j * WAVE, splat(0.f));
    wave_sync();

    for (int ci = 0; ci < nchunks; ++ci) {
        // adjoint order = opposite of the scan order
        const int c = REV0 ? ci : nchunks - 1 - ci;
        const int base = c * S;
        vi t[K];
        vm valid[K];
        AUM_UNROLL
        for (int k = 0; k < K; ++k) { t[k] = lane * K + (base + k); valid[k] = t[k] < p.len; }

        vf dl[SCAN_R][K], dlu[SCAN_R][K], dy[SCAN_R][K], G[SCAN_R][K], DA[SCAN_R][K];
        vf sumd[SCAN_R], sumd_nf[SCAN_R], sumd_nr[SCAN_R], dnf_f[SCAN_R], dnf_r[SCAN_R];
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            const int e = e0 + r;
            const bool rowok = e < p.dim;
            const int ec = rowok ? e : p.dim - 1;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
            const T* gp = row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds);
            const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
            vf sd = splat(0.f);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                const vm m = valid[k] && rowok;
                const vf uu = gload(up, t[k], m);
                vf d = gload(dp, t[k], m) + bias;
                if (softplus) d = vsoftplus(d);
                d = vsel(m, d, splat(0.f));
                dl[r][k] = d; dlu[r][k] = d * uu; sd = sd + d;
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
            sumd[r] = sd;
            // delta at the first step of the scan-order successor chunk (forward: t=base+S; reverse: t=base-1)
            const int tf = base + S, tr = base - 1;
            vf df = splat(0.f), dr = splat(0.f);
            if (tf < p.len) {
                df = gload(dp, spl_i(tf), lane >= 0) + bias;
                if (softplus) df = vsoftplus(df);
            }
            if (tr >= 0) {
                dr = gload(dp, spl_i(tr), lane >= 0) + bias;
                if (softplus) dr = vsoftplus(dr);
            }
            if (!rowok) { df = splat(0.f); dr = splat(0.f); }
            dnf_f[r] = df; dnf_r[r] = dr;
            sumd_nf[r] = sd - dl[r][0] + dpp_wave_shl1(dl[r][0], df);
            sumd_nr[r] = sd - dl[r][K - 1] + dpp_wave_shr1(dl[r][K - 1], dr);
        }

        for (int n = 0; n < N; ++n) {
            vf Bn[K], Cn[K], dBacc[K], dCacc[K];
            const T* Bp = row_ptr<T>(p.B, (int64_t)b * p.B_bs + (int64_t)n * p.B_ns);
            const T* Cp = row_ptr<T>(p.C, (int64_t)b * p.C_bs + (int64_t)n * p.C_ns);
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                Bn[k] = gload(Bp, t[k], valid[k]);
                Cn[k] = gload(Cp, t[k], valid[k]);
                dBacc[k] = splat(0.f);
                dCacc[k] = splat(0.f);
            }
            if (MODE == 0 || BI)
                scan_bwd_dir_state<T, K, false>(p, n, e0, p.A, p.dA, Bn, Cn, dl, dlu, dy, sumd, sumd_nf, dnf_f, ws_x,
                                                ws_stride_row, c, nchunks, lds, G, DA, dBacc, dCacc);
            if (MODE == 1)
                scan_bwd_dir_state<T, K, true>(p, n, e0, p.A, p.dA, Bn, Cn, dl, dlu, dy, sumd, sumd_nr, dnf_r, ws_x,
                                               ws_stride_row, c, nchunks, lds, G, DA, dBacc, dCacc);
            if (BI)
                scan_bwd_dir_state<T, K, true>(p, n, e0, p.A_b, p.dA_b, Bn, Cn, dl, dlu, dy, sumd, sumd_nr, dnf_r,
                                               nullptr, 0, c, nchunks, lds + SCAN_R * SCAN_MAX_DSTATE, G, DA, dBacc,
                                               dCacc);
            float* dBp = p.dB + (int64_t)b * p.dB_bs + (int64_t)n * p.dB_ns;
            float* dCp = p.dC + (int64_t)b * p.dC_bs + (int64_t)n * p.dC_ns;
            AUM_UNROLL
            for (int k = 0; k < K; ++k) {
                gatomic_add(dBp, t[k], dBacc[k], valid[k]);
                gatomic_add(dCp, t[k], dCacc[k], valid[k]);
            }
        }
        wave_sync();

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
                if (p.D && p.dD) gatomic_add(p.dD + e, spl_i(0), splat(ndir * wave_sum(dDl)), lane == 0);
                if (p.ddelta_bias) gatomic_add(p.ddelta_bias + e, spl_i(0), splat(wave_sum(dbl)), lane == 0);
            }
        }
    }
}

}  // namespace aum
