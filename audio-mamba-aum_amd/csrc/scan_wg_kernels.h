// scan_wg_kernels.h -- the production selective-scan kernels (dstate <= 16): 8-wave workgroups.
//
// Same per-wave algorithm as scan_kernels.h (lanes along time, a few steps per lane, DPP/readlane associative
// scan, reverse direction as the suffix form, both directions fused in MODE 2) but organised so that the
// traffic that dominated the single-wave version never leaves the CU (measured on MI355X, profiles/):
//   * one workgroup = 8 waves = up to 64 channel rows of ONE batch element.  The B/C tile of the current
//     chunk is loaded ONCE per workgroup with coalesced loads into LDS (fp32) and read from there by every
//     (row pair, state) -- instead of every wave re-reading B/C through the vector L1;
//   * backward: dB/dC are accumulated across the workgroup's rows in LDS tiles and written as ONE partial tile
//     per workgroup; a second tiny kernel sums the partials.  No global fp32 atomics (device-scope atomics
//     execute at the memory side on this chip: the atomic version spent >95% of its time there) and no LDS
//     atomics either (ds_add_f32 measured ~125 cycles per wave-instruction): the 8 waves walk the states in a
//     rotated order with one barrier per step, so every tile row has exactly one writer at a time;
//   * dA/dD/ddelta_bias leave the kernel as per-(batch,row) partials reduced by the same second kernel.
//
// Slot geometry <K, TAIL>: lane i owns the K consecutive steps [base + iK, +K) ("main" slots) and the LAST lane
// additionally owns TAIL steps [base + 64K, +TAIL).  <8,1> is the AuM shape: L = 513 = 64*8 + 1 (512 patches + the
// cls token), where the main slots of a row are one unmasked 16-byte access per lane and tensor.
#pragma once
#include "scan_kernels.h"

namespace aum {

constexpr int SCANWG_NW = 8;          // waves per workgroup
constexpr int SCANWG_MAX_N = 16;      // dstate limit of this path (LDS tile height)
constexpr int SCANWG_MAX_ROWS = 64;   // rows per workgroup (8 waves x 4 pairs x 2 rows)
constexpr int SCANWG_MAX_K = 9;       // <= 577 steps per chunk keeps the backward's 4 fp32 tiles inside 160 KB of LDS
// debug-only ablation bits (upper half of `flags`; set through AUM_ABLATE in the Python binding, never by the product)
constexpr uint32_t AUM_DBG_SKIP_STATES = 1u << 16, AUM_DBG_SKIP_LDS_ATOMICS = 1u << 17, AUM_DBG_SKIP_PARTIALS = 1u << 18,
                   AUM_DBG_SKIP_EPILOGUE = 1u << 19, AUM_DBG_NO_STEP_BARRIER = 1u << 20;

template <int K, int TAIL> struct ScanGeo {
    static constexpr int KT = K + TAIL;                    // slots per lane
    static constexpr int LK = (K % 2 == 0) ? K + 1 : K;    // LDS words per lane in a tile row: odd => conflict-free
    static constexpr int S = WAVE * K + TAIL;              // time steps per chunk
    static constexpr int SP = WAVE * LK + TAIL;            // LDS words per tile row
    static constexpr int TILE = SCANWG_MAX_N * SP;
};
// geometries with a tail slot are single-chunk only: no carry area, which keeps <8,1> at two workgroups per CU
template <int K, int TAIL> constexpr int scanwg_fwd_lds_floats() { return 2 * ScanGeo<K, TAIL>::TILE + (TAIL ? 0 : 2 * SCANWG_MAX_ROWS * SCANWG_MAX_N); }
template <int K, int TAIL> constexpr int scanwg_bwd_lds_floats() { return 4 * ScanGeo<K, TAIL>::TILE + 3 * SCANWG_MAX_ROWS * SCANWG_MAX_N; }

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

// time index, validity and LDS tile position of every slot of this lane
template <int K, int TAIL>
AUM_DEV void scan_slots(int base, int len, vi (&t)[K + TAIL], vm (&valid)[K + TAIL], vi (&pos)[K + TAIL]) {
    using G = ScanGeo<K, TAIL>;
    const vi lane = lane_id();
    AUM_UNROLL
    for (int k = 0; k < K; ++k) {
        t[k] = lane * K + (base + k);
        valid[k] = t[k] < len;
        pos[k] = lane * G::LK + k;
    }
    AUM_UNROLL
    for (int j = 0; j < TAIL; ++j) {
        t[K + j] = spl_i(base + WAVE * K + j);
        valid[K + j] = (lane == WAVE - 1) && (t[K + j] < len);
        pos[K + j] = spl_i(WAVE * G::LK + j);
    }
}

// tile position of chunk-relative time tt (0 <= tt < S)
template <int K, int TAIL> AUM_DEV vi scan_tile_pos(vi tt) {
    using G = ScanGeo<K, TAIL>;
    const vi ln = tt / K;
    const vi main_pos = ln * G::LK + (tt - ln * K);
    if (TAIL == 0) return main_pos;
    return vsel_i(tt < WAVE * K, main_pos, tt + (WAVE * G::LK - WAVE * K));
}

// Cooperative load of one [N][S] tile of B (or C) into LDS as fp32; t outside [0,len) -> 0.
template <class T, int K, int TAIL>
AUM_DEV void scanwg_load_tile(const T* src, int64_t n_stride, int N, int base, int len, float* tile, int w) {
    using G = ScanGeo<K, TAIL>;
    const vi lane = lane_id();
    for (int i0 = w * WAVE; i0 < N * G::S; i0 += SCANWG_NW * WAVE) {
        const vi idx = lane + i0;
        const vm in = idx < N * G::S;
        const vi n = vmin_i(idx / G::S, N - 1);
        const vi tt = vmin_i(idx - (idx / G::S) * G::S, G::S - 1);
        const vi t = tt + base;
        const vf v = vsel(t < len, gload_u(src, n * (int)n_stride + vmin_i(t, len - 1)), splat(0.f));
        lds_write_m(tile, n * G::SP + scan_tile_pos<K, TAIL>(tt), v, in);
    }
}

// One row, all slots -> fp32 registers.  K == 8 with the main part fully inside the row: one (2-byte types) or two
// (fp32) 16-byte accesses per lane; otherwise clamped unconditional scalar loads (no exec-mask branches).
template <class T, int K, int TAIL>
AUM_DEV void scan_row_read(const T* rp, int base, int len, const vi (&t)[K + TAIL], const vm (&valid)[K + TAIL],
                           vf (&o)[K + TAIL]) {
    bool vec = false;
    if constexpr (K == 8) {
        if (base + WAVE * K <= len) {
            vec = true;
            vf m8[8];
            gload8(rp, lane_id() * 8 + base, lane_id() >= 0, m8);
            AUM_UNROLL
            for (int k = 0; k < 8; ++k) o[k] = m8[k];
            AUM_UNROLL
            for (int j = 0; j < TAIL; ++j) o[K + j] = vsel(valid[K + j], gload_u(rp, vmin_i(t[K + j], len - 1)), splat(0.f));
        }
    }
    if (!vec) {
        AUM_UNROLL
        for (int k = 0; k < K + TAIL; ++k) o[k] = vsel(valid[k], gload_u(rp, vmin_i(t[k], len - 1)), splat(0.f));
    }
}
template <class T, int K, int TAIL>
AUM_DEV void scan_row_write(T* rp, int base, int len, const vi (&t)[K + TAIL], const vm (&valid)[K + TAIL],
                            const vf (&v)[K + TAIL]) {
    bool vec = false;
    if constexpr (K == 8) {
        if (base + WAVE * K <= len) {
            vec = true;
            vf m8[8];
            AUM_UNROLL
            for (int k = 0; k < 8; ++k) m8[k] = v[k];
            gstore8(rp, lane_id() * 8 + base, m8, lane_id() >= 0);
            AUM_UNROLL
            for (int j = 0; j < TAIL; ++j) gstore(rp, t[K + j], v[K + j], valid[K + j]);
        }
    }
    if (!vec) {
        AUM_UNROLL
        for (int k = 0; k < K + TAIL; ++k) gstore(rp, t[k], v[k], valid[k]);
    }
}

// delta = softplus(delta + bias) (masked to 0 outside the row), delta*u, and the lane sum of delta, for the pair's rows
template <class T, int K, int TAIL>
AUM_DEV void scanwg_load_rows(const void* u, int64_t u_bs, int64_t u_ds, const void* delta, int64_t d_bs, int64_t d_ds,
                              const float* delta_bias, bool softplus, int b, int e0, int dim, int base, int len,
                              const vi (&t)[K + TAIL], const vm (&valid)[K + TAIL], vf (&dl)[SCAN_R][K + TAIL],
                              vf (&dlu)[SCAN_R][K + TAIL], vf (&sumd)[SCAN_R]) {
    AUM_UNROLL
    for (int r = 0; r < SCAN_R; ++r) {
        const int e = e0 + r;
        const bool rowok = e < dim;
        const int ec = rowok ? e : dim - 1;
        const T* up = row_ptr<T>(u, (int64_t)b * u_bs + (int64_t)ec * u_ds);
        const T* dp = row_ptr<T>(delta, (int64_t)b * d_bs + (int64_t)ec * d_ds);
        const float bias = delta_bias ? delta_bias[ec] : 0.f;
        vf uu[K + TAIL], dd[K + TAIL];
        scan_row_read<T, K, TAIL>(up, base, len, t, valid, uu);
        scan_row_read<T, K, TAIL>(dp, base, len, t, valid, dd);
        vf sd = splat(0.f);
        AUM_UNROLL
        for (int k = 0; k < K + TAIL; ++k) {
            vf d = dd[k] + bias;
            if (softplus) d = vsoftplus(d);
            d = vsel(valid[k] && rowok, d, splat(0.f));
            dl[r][k] = d;
            dlu[r][k] = d * uu[k];
            sd = sd + d;
        }
        sumd[r] = sd;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_fwd_dir(const AumScanFwdArgs& p, int b, int e0, int rloc, const float* Aptr, const float* Bt,
                            const float* Ct, const vi (&pos)[K + TAIL], const vf (&dl)[SCAN_R][K + TAIL],
                            const vf (&dlu)[SCAN_R][K + TAIL], const vf (&sumd)[SCAN_R], float* carry, bool multi,
                            bool write_last, vf (&y)[SCAN_R][K + TAIL]) {
    using G = ScanGeo<K, TAIL>;
    constexpr int KT = G::KT;
    const int N = p.dstate;
    const vi lane = lane_id();
    for (int n = 0; n < N; ++n) {
        vf Bn[KT], Cn[KT];
        AUM_UNROLL
        for (int k = 0; k < KT; ++k) {
            Bn[k] = lds_read(Bt, pos[k] + n * G::SP);
            Cn[k] = lds_read(Ct, pos[k] + n * G::SP);
        }
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            const int e = e0 + r;
            if (e < p.dim) {
                const float An = Aptr[(int64_t)e * N + n] * LOG2E;
                vf a[KT], bb[KT], x[KT];
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) {
                    a[k] = vexp2(dl[r][k] * An);
                    bb[k] = dlu[r][k] * Bn[k];
                }
                const vf Ptot = vexp2(sumd[r] * An);
                vf cin = splat(0.f);
                if (multi) cin = lds_read(carry, spl_i((rloc + r) * SCANWG_MAX_N + n));
                vf xin, cout;
                affine_scan_states<KT, REV>(a, bb, Ptot, cin, x, xin, cout);
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) y[r][k] = vfma(Cn[k], x[k], y[r][k]);
                if (multi) lds_write(carry, spl_i((rloc + r) * SCANWG_MAX_N + n), cout);
                if (write_last) gstore(p.last_state + ((int64_t)b * p.dim + e) * N + n, spl_i(0), cout, lane == 0);
            }
        }
    }
}

template <class T, int K, int TAIL, int MODE>
AUM_DEV void scanwg_fwd(const AumScanFwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using G = ScanGeo<K, TAIL>;
    constexpr bool BI = MODE == 2;
    constexpr int KT = G::KT;
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + G::TILE;
    float* carry = lds + 2 * G::TILE;     // [2][SCANWG_MAX_ROWS][SCANWG_MAX_N]
    const int gpb = (p.dim + rows_per_wg - 1) / rows_per_wg;
    const int b = wg / gpb;
    const int eb = (wg % gpb) * rows_per_wg;
    const int nchunks = (p.len + G::S - 1) / G::S;
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
        const int base = c * G::S;
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            scanwg_load_tile<T, K, TAIL>(Bsrc, p.B_ns, N, base, p.len, Bt, w);
            scanwg_load_tile<T, K, TAIL>(Csrc, p.C_ns, N, base, p.len, Ct, w);
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                const int rloc = 2 * pair;
                const int e0 = eb + rloc;
                if (e0 >= p.dim) break;
                vi t[KT], pos[KT];
                vm valid[KT];
                scan_slots<K, TAIL>(base, p.len, t, valid, pos);
                vf dl[SCAN_R][KT], dlu[SCAN_R][KT], y[SCAN_R][KT], sumd[SCAN_R];
                scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                             e0, p.dim, base, p.len, t, valid, dl, dlu, sumd);
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    AUM_UNROLL
                    for (int k = 0; k < KT; ++k) y[r][k] = splat(0.f);
                }
                const bool wl = (ci == nchunks - 1) && p.last_state != nullptr;
                if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                    if (MODE == 0 || BI)
                        scanwg_fwd_dir<T, K, TAIL, false>(p, b, e0, rloc, p.A, Bt, Ct, pos, dl, dlu, sumd, carry, multi,
                                                          wl && !BI, y);
                    if (MODE == 1)
                        scanwg_fwd_dir<T, K, TAIL, true>(p, b, e0, rloc, p.A, Bt, Ct, pos, dl, dlu, sumd, carry, multi, wl, y);
                    if (BI)
                        scanwg_fwd_dir<T, K, TAIL, true>(p, b, e0, rloc, p.A_b, Bt, Ct, pos, dl, dlu, sumd,
                                                         carry + SCANWG_MAX_ROWS * SCANWG_MAX_N, multi, false, y);
                }
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    const int e = e0 + r;
                    if (e < p.dim) {
                        const float Dn = p.D ? (BI ? 2.f : 1.f) * p.D[e] : 0.f;
                        const int64_t ooff = (int64_t)b * p.out_bs + (int64_t)e * p.out_ds;
                        vf o[KT];
                        if (p.D) {
                            vf uu[KT];
                            scan_row_read<T, K, TAIL>(row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds), base, p.len, t,
                                                      valid, uu);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = vfma(uu[k], splat(Dn), y[r][k]);
                        } else {
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = y[r][k];
                        }
                        if (p.out_pre) scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.out_pre, ooff), base, p.len, t, valid, o);
                        if (p.z) {
                            vf zz[KT];
                            scan_row_read<T, K, TAIL>(row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)e * p.z_ds), base, p.len, t,
                                                      valid, zz);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = o[k] * (zz[k] * vsigmoid(zz[k]));
                        }
                        scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.out, ooff), base, p.len, t, valid, o);
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
template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_bwd_dir_state(const AumScanBwdArgs& p, int n, int b, int e0, int rloc, const float* Aptr,
                                  const vf (&Bn)[K + TAIL], const vf (&Cn)[K + TAIL],
                                  const vf (&dl)[SCAN_R][K + TAIL], const vf (&dlu)[SCAN_R][K + TAIL],
                                  const vf (&dy)[SCAN_R][K + TAIL], const vf (&sumd)[SCAN_R],
                                  const vf (&sumd_next)[SCAN_R], const vf (&dnf)[SCAN_R], const float* xck,
                                  int64_t xck_row_stride, int chunk, float* gcarry, bool multi,
                                  vf (&G)[SCAN_R][K + TAIL], vf (&DA)[SCAN_R][K + TAIL], vf (&dBacc)[K + TAIL],
                                  vf (&dCacc)[K + TAIL], vf (&dAv)[SCAN_R]) {
    constexpr int KT = K + TAIL;
    const int N = p.dstate;
    AUM_UNROLL
    for (int r = 0; r < SCAN_R; ++r) {
        const int e = e0 + r;
        if (e < p.dim) {
            const float Araw = Aptr[(int64_t)e * N + n];
            const float An = Araw * LOG2E;
            vf a[KT], bb[KT], x[KT];
            AUM_UNROLL
            for (int k = 0; k < KT; ++k) {
                a[k] = vexp2(dl[r][k] * An);
                bb[k] = dlu[r][k] * Bn[k];
            }
            const vf Ptot = vexp2(sumd[r] * An);
            vf cin = splat(0.f);
            if (multi) cin = gload_coherent(xck + (rloc + r) * xck_row_stride + (int64_t)chunk * N + n, spl_i(0), lane_id() >= 0);
            vf xin, cout;
            affine_scan_states<KT, REV>(a, bb, Ptot, cin, x, xin, cout);
            vf an[KT], cc[KT], g[KT];
            const vf a_nf = vexp2(dnf[r] * An);
            AUM_UNROLL
            for (int k = 0; k < KT; ++k) {
                cc[k] = dy[r][k] * Cn[k];
                if (!REV) an[k] = (k + 1 < KT) ? a[k + 1 < KT ? k + 1 : 0] : dpp_wave_shl1(a[0], a_nf);
                else      an[k] = (k > 0) ? a[k > 0 ? k - 1 : 0] : dpp_wave_shr1(a[KT - 1], a_nf);
            }
            const vf Pn = vexp2(sumd_next[r] * An);
            vf gin_c = splat(0.f);
            if (multi) gin_c = lds_read(gcarry, spl_i((rloc + r) * SCANWG_MAX_N + n));
            vf gin, gout;
            affine_scan_states<KT, !REV>(an, cc, Pn, gin_c, g, gin, gout);
            if (multi) lds_write(gcarry, spl_i((rloc + r) * SCANWG_MAX_N + n), gout);
            vf dAl = splat(0.f);
            AUM_UNROLL
            for (int k = 0; k < KT; ++k) {
                const vf xprev = REV ? ((k == KT - 1) ? xin : x[k + 1 < KT ? k + 1 : 0]) : ((k == 0) ? xin : x[k > 0 ? k - 1 : 0]);
                const vf h = g[k] * a[k] * xprev;
                G[r][k] = vfma(g[k], Bn[k], G[r][k]);
                DA[r][k] = vfma(splat(Araw), h, DA[r][k]);
                dBacc[k] = vfma(g[k], dlu[r][k], dBacc[k]);
                dCacc[k] = vfma(dy[r][k], x[k], dCacc[k]);
                dAl = vfma(dl[r][k], h, dAl);
            }
            // dA[e][n] partial of this (batch,row): parked in lane n, written once per row after the state loop
            if (!(p.flags & AUM_DBG_SKIP_PARTIALS)) dAv[r] = vsel(lane_id() == n, splat(wave_sum(dAl)), dAv[r]);
        }
    }
}

template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_bwd_prepass_pair(const AumScanBwdArgs& p, int b, int e0, int rloc, int c, int base, const float* Aptr,
                                     const float* Bt, float* xck, int64_t xck_row_stride, float* xcarry) {
    using G = ScanGeo<K, TAIL>;
    constexpr int KT = G::KT;
    const int N = p.dstate;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();
    vi t[KT], pos[KT];
    vm valid[KT];
    scan_slots<K, TAIL>(base, p.len, t, valid, pos);
    vf dl[SCAN_R][KT], dlu[SCAN_R][KT], sumd[SCAN_R];
    scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b, e0, p.dim,
                                 base, p.len, t, valid, dl, dlu, sumd);
    for (int n = 0; n < N; ++n) {
        vf Bn[KT];
        AUM_UNROLL
        for (int k = 0; k < KT; ++k) Bn[k] = lds_read(Bt, pos[k] + n * G::SP);
        AUM_UNROLL
        for (int r = 0; r < SCAN_R; ++r) {
            if (e0 + r < p.dim) {
                const float An = Aptr[(int64_t)(e0 + r) * N + n] * LOG2E;
                vf a[KT], bb[KT], x[KT];
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) { a[k] = vexp2(dl[r][k] * An); bb[k] = dlu[r][k] * Bn[k]; }
                const vf cin = lds_read(xcarry, spl_i((rloc + r) * SCANWG_MAX_N + n));
                gstore_coherent(xck + (rloc + r) * xck_row_stride + (int64_t)c * N + n, spl_i(0), cin, lane == 0);
                vf xin, cout;
                affine_scan_states<KT, REV>(a, bb, vexp2(sumd[r] * An), cin, x, xin, cout);
                lds_write(xcarry, spl_i((rloc + r) * SCANWG_MAX_N + n), cout);
            }
        }
    }
}

template <class T, int K, int TAIL, int MODE>
AUM_DEV void scanwg_bwd(const AumScanBwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using GE = ScanGeo<K, TAIL>;
    constexpr bool BI = MODE == 2;
    constexpr bool REV0 = MODE == 1;
    constexpr int KT = GE::KT;
    constexpr int S = GE::S;
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + GE::TILE;
    float* dBt = lds + 2 * GE::TILE;
    float* dCt = lds + 3 * GE::TILE;
    float* xcarry = lds + 4 * GE::TILE;                                // [rows][N]   (multi-chunk pre-pass)
    float* gcarry = xcarry + SCANWG_MAX_ROWS * SCANWG_MAX_N;           // [2][rows][N]
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
            AUM_FOR_EACH_WAVE(w, SCANWG_NW) { scanwg_load_tile<T, K, TAIL>(Bsrc, p.B_ns, N, c * S, p.len, Bt, w); }
            AUM_WG_BARRIER();
            AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
                for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                    if (eb + 2 * pair >= p.dim) break;
                    scanwg_bwd_prepass_pair<T, K, TAIL, REV0>(p, b, eb + 2 * pair, 2 * pair, c, c * S, p.A, Bt, xck,
                                                              xck_row_stride, xcarry);
                }
            }
            AUM_WG_BARRIER();
        }
    }

    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = REV0 ? ci : nchunks - 1 - ci;     // adjoint order = opposite of the scan order
        const int base = c * S;
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            scanwg_load_tile<T, K, TAIL>(Bsrc, p.B_ns, N, base, p.len, Bt, w);
            scanwg_load_tile<T, K, TAIL>(Csrc, p.C_ns, N, base, p.len, Ct, w);
            for (int i0 = w * WAVE; i0 < GE::TILE; i0 += SCANWG_NW * WAVE) {
                const vi idx = lane_id() + i0;
                lds_write_m(dBt, idx, splat(0.f), idx < GE::TILE);
                lds_write_m(dCt, idx, splat(0.f), idx < GE::TILE);
            }
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            const vi lane = lane_id();
            for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                const int rloc = 2 * pair;
                const int e0 = eb + rloc;
                const bool active = e0 < p.dim;      // wave-uniform; inactive waves still take every barrier below
                vi t[KT], pos[KT];
                vm valid[KT];
                scan_slots<K, TAIL>(base, p.len, t, valid, pos);
                vf dl[SCAN_R][KT], dlu[SCAN_R][KT], dy[SCAN_R][KT], G[SCAN_R][KT], DA[SCAN_R][KT];
                vf sumd[SCAN_R], sumd_nf[SCAN_R], sumd_nr[SCAN_R], dnf_f[SCAN_R], dnf_r[SCAN_R];
                scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                             e0, p.dim, base, p.len, t, valid, dl, dlu, sumd);   // rows >= dim give zeros
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) {
                    const int e = e0 + r;
                    const bool rowok = e < p.dim;
                    const int ec = rowok ? e : p.dim - 1;
                    const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
                    const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
                    vf go[KT];
                    scan_row_read<T, K, TAIL>(row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds), base, p.len,
                                              t, valid, go);
                    if (p.z) {
                        vf zz[KT], yp[KT], dzv[KT];
                        scan_row_read<T, K, TAIL>(row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)ec * p.z_ds), base, p.len, t,
                                                  valid, zz);
                        scan_row_read<T, K, TAIL>(row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)ec * p.out_ds), base,
                                                  p.len, t, valid, yp);
                        AUM_UNROLL
                        for (int k = 0; k < KT; ++k) {
                            const vf sg = vsigmoid(zz[k]);
                            dzv[k] = go[k] * yp[k] * sg * vfma(zz[k], splat(1.f) - sg, splat(1.f));
                            go[k] = go[k] * zz[k] * sg;
                        }
                        if (rowok)
                            scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)ec * p.dz_ds), base,
                                                       p.len, t, valid, dzv);
                    }
                    AUM_UNROLL
                    for (int k = 0; k < KT; ++k) {
                        dy[r][k] = rowok ? go[k] : splat(0.f);
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
                    sumd_nr[r] = sumd[r] - dl[r][KT - 1] + dpp_wave_shr1(dl[r][KT - 1], dr);
                }
                const bool first_visit = ci == 0;
                vf dAv0[SCAN_R], dAv1[SCAN_R];      // lane n <- dA (dA_b) partial of state n, per row
                AUM_UNROLL
                for (int r = 0; r < SCAN_R; ++r) { dAv0[r] = splat(0.f); dAv1[r] = splat(0.f); }
                // Rotated state order: at step j wave w works on state (j + 2w) mod 16, so no two of the 8 waves ever
                // hold the same state row of the dB/dC tiles within a step, nor in adjacent steps (j+1+2w = j+2w' has no
                // solution); a barrier after every SECOND step therefore keeps any two waves at most one step apart and
                // the tile update below can be a plain LDS read-add-write.
                for (int j = 0; j < SCANWG_MAX_N; ++j) {
                    const int n = (j + 2 * w) & (SCANWG_MAX_N - 1);
                    if (active && n < N) {
                        vf Bn[KT], Cn[KT], dBacc[KT], dCacc[KT];
                        AUM_UNROLL
                        for (int k = 0; k < KT; ++k) {
                            Bn[k] = lds_read(Bt, pos[k] + n * GE::SP);
                            Cn[k] = lds_read(Ct, pos[k] + n * GE::SP);
                            dBacc[k] = splat(0.f);
                            dCacc[k] = splat(0.f);
                        }
                        if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                            if (MODE == 0 || BI)
                                scanwg_bwd_dir_state<T, K, TAIL, false>(p, n, b, e0, rloc, p.A, Bn, Cn, dl, dlu, dy, sumd, sumd_nf,
                                                                        dnf_f, xck, xck_row_stride, c, gcarry, multi, G, DA,
                                                                        dBacc, dCacc, dAv0);
                            if (MODE == 1)
                                scanwg_bwd_dir_state<T, K, TAIL, true>(p, n, b, e0, rloc, p.A, Bn, Cn, dl, dlu, dy, sumd, sumd_nr,
                                                                       dnf_r, xck, xck_row_stride, c, gcarry, multi, G, DA,
                                                                       dBacc, dCacc, dAv0);
                            if (BI)
                                scanwg_bwd_dir_state<T, K, TAIL, true>(p, n, b, e0, rloc, p.A_b, Bn, Cn, dl, dlu, dy, sumd, sumd_nr,
                                                                       dnf_r, nullptr, 0, c,
                                                                       gcarry + SCANWG_MAX_ROWS * SCANWG_MAX_N, false, G, DA,
                                                                       dBacc, dCacc, dAv1);
                        }
                        if (!(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) {
                                const vi at = pos[k] + n * GE::SP;
                                // tail slots are shared by all lanes but owned by the last one
                                const vm own = k < K ? (lane >= 0) : (lane == WAVE - 1);
                                lds_write_m(dBt, at, lds_read(dBt, at) + dBacc[k], own);
                                lds_write_m(dCt, at, lds_read(dCt, at) + dCacc[k], own);
                            }
                        }
                    }
                    if ((j & 1) && !(p.flags & AUM_DBG_NO_STEP_BARRIER)) AUM_WG_BARRIER_IN_PHASE();
                }
                if (active && !(p.flags & AUM_DBG_SKIP_PARTIALS)) {
                    AUM_UNROLL
                    for (int r = 0; r < SCAN_R; ++r) {
                        const int e = e0 + r;
                        if (e < p.dim) {
                            float* sA = ws + L.pA + ((int64_t)b * p.dim + e) * N;
                            float* sAb = ws + L.pAb + ((int64_t)b * p.dim + e) * N;
                            const vm mn = lane < N;
                            const vi ln = vmin_i(lane, N - 1);
                            if (!first_visit) {     // later chunks of a multi-chunk row accumulate (same wave wrote it)
                                dAv0[r] = dAv0[r] + gload_coherent(sA, ln, mn);
                                gstore_coherent(sA, ln, dAv0[r], mn);
                            } else if (multi) {
                                gstore_coherent(sA, ln, dAv0[r], mn);
                            } else {
                                gstore(sA, ln, dAv0[r], mn);
                                if (BI) gstore(sAb, ln, dAv1[r], mn);
                            }
                        }
                    }
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
                            vf uu[KT], raw[KT], duv[KT], ddv[KT];
                            scan_row_read<T, K, TAIL>(up, base, p.len, t, valid, uu);
                            if (softplus) scan_row_read<T, K, TAIL>(dp, base, p.len, t, valid, raw);
                            vf dDl = splat(0.f), dbl = splat(0.f);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) {
                                duv[k] = vfma(dl[r][k], G[r][k], dy[r][k] * Dn);
                                vf dd = vfma(uu[k], G[r][k], DA[r][k]);
                                if (softplus) {
                                    const vf rw = raw[k] + bias;
                                    dd = vsel(rw > 20.f, dd, dd * vsigmoid(rw));
                                }
                                dd = vsel(valid[k], dd, splat(0.f));
                                ddv[k] = dd;
                                dDl = vfma(dy[r][k], uu[k], dDl);
                                dbl = dbl + dd;
                            }
                            scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), base, p.len,
                                                       t, valid, duv);
                            scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds),
                                                       base, p.len, t, valid, ddv);
                            float sD = ndir * wave_sum(dDl), sb = wave_sum(dbl);
                            float* slotD = ws + L.pD + (int64_t)b * p.dim + e;
                            float* slotb = ws + L.pbias + (int64_t)b * p.dim + e;
                            if (!first_visit) {
                                sD += readlane(gload_coherent(slotD, spl_i(0), lane >= 0), 0);
                                sb += readlane(gload_coherent(slotb, spl_i(0), lane >= 0), 0);
                            }
                            if (multi) {
                                gstore_coherent(slotD, spl_i(0), splat(sD), lane == 0);
                                gstore_coherent(slotb, spl_i(0), splat(sb), lane == 0);
                            } else {
                                gstore(slotD, spl_i(0), splat(sD), lane == 0);
                                gstore(slotb, spl_i(0), splat(sb), lane == 0);
                            }
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
                const vi n = vmin_i(idx / S, N - 1);
                const vi tt = vmin_i(idx - (idx / S) * S, S - 1);
                const vi tg = tt + base;
                const vm m = (idx < N * S) && (tg < p.len);
                const vi at = n * GE::SP + scan_tile_pos<K, TAIL>(tt);
                gstore(oB, n * p.len + tg, lds_read(dBt, at), m);
                gstore(oC, n * p.len + tg, lds_read(dCt, at), m);
            }
        }
        AUM_WG_BARRIER();
    }
}

// Second stage of the backward: dst[i] += sum_o src[o*inner + i] for up to 6 segments.
struct ScanReduceSeg { float* dst; const float* src; int64_t inner; int32_t outer; int32_t pad; };
struct ScanReduceArgs { ScanReduceSeg seg[6]; int32_t nseg; };

}  // namespace aum
