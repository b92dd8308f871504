// scan_half_kernels.h -- backward selective scan for the AuM row shape (512 main steps + at most one tail step in a single
// chunk, dstate <= 16), ONE row per wave.
//
// scan_wg_kernels.h packs the two rows of a row PAIR into the halves of a vf2 so that every VALU instruction is a packed
// (2 x fp32) one.  For the backward that doubles the live state of a lane -- five persistent and seven per-state 9-slot
// arrays of vf2 -- past the 256-register budget of an 8-wave workgroup: 288 bytes of scratch per lane, ~1 GB of spill
// traffic per launch.  Here the two halves of a vf2 are the two HALVES OF ONE LANE'S 8 STEPS (slots i and 4+i), i.e. the
// same packed-math rate with half the registers:
//   * element-wise work: 4 packed + 1 tail instruction per row and array;
//   * the in-lane recurrence becomes a two-level scan: both 4-step halves run as one packed chain from a zero entry,
//     the halves are composed with the products of their multipliers, the lane totals go through the same DPP wave scan,
//     and the second pass enters the upper half with P_lo * x_in + S_lo;
//   * lane totals are products of the step multipliers, so the exp(sum delta) side computations of the pair kernel and
//     their chunk-edge bookkeeping disappear (single chunk: no carries).
// Everything outside the state loop (B/C tiles in LDS, rotated state order with plain read-add-write of the dB/dC tiles,
// per-workgroup partials reduced by k_scan_reduce, workspace layout) is that of scanwg_bwd and shares its helpers.
// Reference: SSI:62-65 / 541-561 (selective_scan_cuda.bwd call sites), SSI:86-152 (selective_scan_ref) for the math.
#pragma once
#include "scan_wg_kernels.h"

namespace aum {

// Waves per workgroup / rows per workgroup by direction mode (measured on MI355X, B = 64, bf16): the fused bidirectional
// kernel needs 165 VGPRs -> 12 waves (3 per SIMD) 1.60 ms vs 8 waves 1.78 ms vs 16 waves 1.93 ms (156 B/lane of scratch);
// the one-direction kernels fit 128 VGPRs -> 16 waves 0.97 ms vs 12 waves 1.05 ms vs 8 waves 1.18 ms.
AUM_HOSTDEV constexpr int scanh_nw(int mode) { return mode == 2 ? 12 : 16; }
AUM_HOSTDEV constexpr int scanh_rows(int mode) { return mode == 2 ? 96 : 64; }      // a multiple of the wave count
// state rotation of the dB/dC tile updates (see scanwg_bwd): 8 waves would use offsets 2 apart and a barrier every second
// step; denser wave counts use adjacent offsets and a barrier after every step, which keeps the plain read-add-write race-free
template <int TAIL> constexpr int scanh_bwd_lds_floats() { return 4 * ScanGeo<8, TAIL>::TILE; }
AUM_HOSTDEV bool scanh_shape_ok(int len, int dstate, int nchunks) { return nchunks == 1 && len >= 512 && len <= 513 && dstate <= SCANWG_MAX_N; }
// the one-row kernel replaces scanwg_bwd<8,1> for these launches (same test in the dispatcher and the launcher)
AUM_HOSTDEV bool scanh_selected(int K, int tail, int nchunks, int len, int dstate, uint32_t flags) {
    return K == 8 && tail == 1 && !(flags & AUM_SCAN_ROWPAIR) && scanh_shape_ok(len, dstate, nchunks);
}

// One row -> half-packed slots: m[i] = (step 8*lane + i, step 8*lane + 4 + i), tl = the tail step (last lane only).
template <class T> AUM_DEV void scanh_row_read(const T* rp, int len, vf2 (&m)[4], vf& tl) {
    const vi lane = lane_id();
    vf v[8];
    gload8(rp, lane * 8, lane >= 0, v);
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) m[i] = mk2(v[i], v[4 + i]);
    const vm tv = (lane == WAVE - 1) && (spl_i(len) > 512);
    tl = vsel(tv, gload_u(rp, spl_i(len - 1)), splat(0.f));
}
template <class T> AUM_DEV void scanh_row_write(T* rp, int len, const vf2 (&m)[4], vf tl) {
    const vi lane = lane_id();
    vf v[8];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) { v[i] = lo2(m[i]); v[4 + i] = hi2(m[i]); }
    gstore8(rp, lane * 8, v, lane >= 0);
    gstore(rp, spl_i(len - 1), tl, (lane == WAVE - 1) && (spl_i(len) > 512));
}

// x' = m_k * x + b_k over the lane's 9 slots (8 half-packed + tail) and the 64 lanes; zero state before the first step.
// REV = false: steps in order 0..7, tail, lanes 0 -> 63.  REV = true: tail, 7..0, lanes 63 -> 0.
// Outputs: x[i] = state AFTER slots (i, 4+i), x8 after the tail slot, x_in = state entering the lane.
template <bool REV> AUM_DEV void scanh_affine(const vf2 (&m)[4], vf m8, const vf2 (&b)[4], vf b8, vf2 (&x)[4], vf& x8, vf& x_in) {
    vf2 s = spl2(splat(0.f));
    vf2 Pp = m[0] * m[1];
    Pp = Pp * m[2];
    Pp = Pp * m[3];
    AUM_UNROLL
    for (int ii = 0; ii < 4; ++ii) {
        const int i = REV ? 3 - ii : ii;
        s = vfma2(m[i], s, b[i]);
    }
    const vf Plo = lo2(Pp), Phi = hi2(Pp);
    vf S, P = Plo * Phi * m8;
    if (!REV) {
        S = vfma(Phi, lo2(s), hi2(s));
        S = vfma(m8, S, b8);
    } else {
        S = vfma(Phi, b8, hi2(s));
        S = vfma(Plo, S, lo2(s));
    }
    wave_scan_affine<REV>(P, S);
    x_in = REV ? dpp_wave_shl1(S, splat(0.f)) : dpp_wave_shr1(S, splat(0.f));
    vf2 xx;
    if (!REV) {
        xx = mk2(x_in, vfma(Plo, x_in, lo2(s)));
    } else {
        x8 = vfma(m8, x_in, b8);
        xx = mk2(vfma(Phi, x8, hi2(s)), x8);
    }
    AUM_UNROLL
    for (int ii = 0; ii < 4; ++ii) {
        const int i = REV ? 3 - ii : ii;
        xx = vfma2(m[i], xx, b[i]);
        x[i] = xx;
    }
    if (!REV) x8 = vfma(m8, hi2(x[3]), b8);
}

// one (state n, direction) of one row: forward states, adjoint, and the five accumulations
template <bool REV>
AUM_DEV void scanh_bwd_dir_state(float Araw, int n, const vf2 (&Bn)[4], vf Bn8, const vf2 (&Cn)[4], vf Cn8, const vf2 (&dl)[4], vf dl8,
                                 const vf2 (&dlu)[4], vf dlu8, const vf2 (&dy)[4], vf dy8, vf2 (&G)[4], vf& G8, vf2 (&DA)[4], vf& DA8,
                                 vf2 (&dBacc)[4], vf& dB8, vf2 (&dCacc)[4], vf& dC8, vf& dAv, bool want_dA) {
    const float An = Araw * LOG2E;
    vf2 a[4], bb[4], x[4], cc[4], m[4], g[4];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        a[i] = vexp2_2(dl[i] * spl2(splat(An)));
        bb[i] = dlu[i] * Bn[i];
        cc[i] = dy[i] * Cn[i];
    }
    const vf a8 = vexp2(dl8 * An), bb8 = dlu8 * Bn8, cc8 = dy8 * Cn8;
    vf x8, xin, g8, gin;
    scanh_affine<REV>(a, a8, bb, bb8, x, x8, xin);
    // adjoint g_k = dy_k C_k + a_succ(k) * g_succ(k), scanned against the recurrence; the multiplier of a slot is the `a` of its
    // scan successor (the neighbour lane's first slot at the lane edge; 1 past the end of the row)
    vf m8;
    if (!REV) {
        m[0] = a[1]; m[1] = a[2]; m[2] = a[3];
        m[3] = mk2(hi2(a[0]), a8);
        m8 = dpp_wave_shl1(lo2(a[0]), splat(1.f));
    } else {
        m[0] = mk2(dpp_wave_shr1(a8, splat(1.f)), lo2(a[3]));
        m[1] = a[0]; m[2] = a[1]; m[3] = a[2];
        m8 = hi2(a[3]);
    }
    scanh_affine<!REV>(m, m8, cc, cc8, g, g8, gin);
    (void)gin;
    vf2 dAl = spl2(splat(0.f));
    const vf2 Ar = spl2(splat(Araw));
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        vf2 xprev;
        if (!REV) xprev = i == 0 ? mk2(xin, lo2(x[3])) : x[i > 0 ? i - 1 : 0];
        else xprev = i == 3 ? mk2(hi2(x[0]), x8) : x[i < 3 ? i + 1 : 3];
        const vf2 h = g[i] * a[i] * xprev;
        G[i] = vfma2(g[i], Bn[i], G[i]);
        DA[i] = vfma2(Ar, h, DA[i]);
        dBacc[i] = vfma2(g[i], dlu[i], dBacc[i]);
        dCacc[i] = vfma2(dy[i], x[i], dCacc[i]);
        dAl = vfma2(dl[i], h, dAl);
    }
    {
        const vf xprev8 = REV ? xin : hi2(x[3]);
        const vf h8 = g8 * a8 * xprev8;
        G8 = vfma(g8, Bn8, G8);
        DA8 = vfma(splat(Araw), h8, DA8);
        dB8 = vfma(g8, dlu8, dB8);
        dC8 = vfma(dy8, x8, dC8);
        // dA[e][n] = sum over the row: reduced inside each 16-lane row now (lane 16q + n keeps row q's share), the four
        // shares are added once per row after the state loop (sum_rows4) -- no v_readlane chain per state
        if (want_dA) dAv = vsel((lane_id() & 15) == n, row_sum16(vfma(dl8, h8, lo2(dAl) + hi2(dAl))), dAv);
    }
}

template <class T, int TAIL, int MODE>
AUM_DEV void scanh_bwd(const AumScanBwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using GE = ScanGeo<8, TAIL>;
    constexpr bool BI = MODE == 2;
    constexpr int SCANH_NW = scanh_nw(MODE);
    constexpr int SCANH_ROT = SCANWG_MAX_N / SCANH_NW >= 2 ? SCANWG_MAX_N / SCANH_NW : 1;
    constexpr int SCANH_BARRIER_MASK = SCANH_ROT >= 2 ? 1 : 0;   // barrier after step j when (j & mask) == mask
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + GE::TILE;
    float* dBt = lds + 2 * GE::TILE;
    float* dCt = lds + 3 * GE::TILE;
    const ScanWgWs L = scanwg_ws_layout(p.batch, p.dim, p.len, N, rows_per_wg, 1, BI);
    float* ws = (float*)p.workspace;
    const int b = wg / L.gpb, g_idx = wg % L.gpb;
    const int eb = g_idx * rows_per_wg;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const float ndir = BI ? 2.f : 1.f;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);

    AUM_FOR_EACH_WAVE(w, SCANH_NW) {
        scanwg_load_tile<T, 8, TAIL, SCANH_NW>(Bsrc, p.B_ns, N, 0, p.len, Bt, w);
        scanwg_load_tile<T, 8, TAIL, SCANH_NW>(Csrc, p.C_ns, N, 0, p.len, Ct, w);
        for (int i0 = w * WAVE; i0 < GE::TILE; i0 += SCANH_NW * WAVE) {
            const vi idx = lane_id() + i0;
            lds_write_m(dBt, idx, splat(0.f), idx < GE::TILE);
            lds_write_m(dCt, idx, splat(0.f), idx < GE::TILE);
        }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, SCANH_NW) {
        const vi lane = lane_id();
        const vm tail_lane = (lane == WAVE - 1) && (spl_i(p.len) > 512);
        vi pos[4], pos4[4];
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { pos[i] = lane * GE::LK + i; pos4[i] = lane * GE::LK + 4 + i; }
        const vi pos8 = spl_i(WAVE * GE::LK);
        const int niter = (rows_per_wg + SCANH_NW - 1) / SCANH_NW;
        for (int it = 0; it < niter; ++it) {
            const int rloc = w + it * SCANH_NW;
            const int e = eb + rloc;
            const bool active = rloc < rows_per_wg && e < p.dim;   // wave-uniform; inactive waves still take every barrier below
            const int ec = active ? e : p.dim - 1;
            const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
            vf2 dl[4], dlu[4], dy[4], G[4], DA[4];
            vf dl8, dlu8, dy8, G8 = splat(0.f), DA8 = splat(0.f);
            {   // delta = softplus(delta + bias), delta * u
                vf2 uu[4], dd[4];
                vf u8, d8;
                scanh_row_read<T>(up, p.len, uu, u8);
                scanh_row_read<T>(dp, p.len, dd, d8);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const vf2 dr = dd[i] + spl2(splat(bias));
                    dl[i] = softplus ? vsoftplus2(dr) : dr;
                    dlu[i] = dl[i] * uu[i];
                    G[i] = spl2(splat(0.f));
                    DA[i] = spl2(splat(0.f));
                }
                vf d = d8 + bias;
                if (softplus) d = vsoftplus(d);
                dl8 = vsel(tail_lane, d, splat(0.f));       // identity step (a = 1, b = 0) outside the tail lane
                dlu8 = dl8 * u8;
            }
            {   // dout (and the gate): dy = dout * silu(z), dz = dout * out_pre * silu'(z)
                vf2 go[4];
                vf go8;
                scanh_row_read<T>(row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds), p.len, go, go8);
                if (p.z) {
                    vf2 zz[4], yp[4], dzv[4];
                    vf z8, yp8, dz8;
                    scanh_row_read<T>(row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)ec * p.z_ds), p.len, zz, z8);
                    scanh_row_read<T>(row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)ec * p.out_ds), p.len, yp, yp8);
                    const vf2 one2 = spl2(splat(1.f));
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        const vf2 sg = vsigmoid2(zz[i]);
                        dzv[i] = go[i] * yp[i] * sg * vfma2(zz[i], one2 - sg, one2);
                        go[i] = go[i] * zz[i] * sg;
                    }
                    const vf sg = vsigmoid(z8);
                    dz8 = go8 * yp8 * sg * vfma(z8, splat(1.f) - sg, splat(1.f));
                    go8 = go8 * z8 * sg;
                    if (active) scanh_row_write<T>(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)ec * p.dz_ds), p.len, dzv, dz8);
                }
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) dy[i] = go[i];
                dy8 = vsel(tail_lane, go8, splat(0.f));
            }
            vf dAv0 = splat(0.f), dAv1 = splat(0.f);          // lane n <- dA (dA_b) partial of state n
            const bool want_dA = !(p.flags & AUM_DBG_SKIP_PARTIALS);
            // rotated state order: see scanwg_bwd -- no two waves hold the same dB/dC tile row in the same or adjacent steps
            for (int j = 0; j < SCANWG_MAX_N; ++j) {
                const int n = (j + SCANH_ROT * w) & (SCANWG_MAX_N - 1);
                if (active && n < N) {
                    vf2 Bn[4], Cn[4], dBacc[4], dCacc[4];
                    vf dB8 = splat(0.f), dC8 = splat(0.f);
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        Bn[i] = mk2(lds_read(Bt, pos[i] + n * GE::SP), lds_read(Bt, pos4[i] + n * GE::SP));
                        Cn[i] = mk2(lds_read(Ct, pos[i] + n * GE::SP), lds_read(Ct, pos4[i] + n * GE::SP));
                        dBacc[i] = spl2(splat(0.f));
                        dCacc[i] = spl2(splat(0.f));
                    }
                    const vf Bn8 = lds_read(Bt, pos8 + n * GE::SP), Cn8 = lds_read(Ct, pos8 + n * GE::SP);
                    if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                        if (MODE == 0 || BI)
                            scanh_bwd_dir_state<false>(p.A[(int64_t)ec * N + n], n, Bn, Bn8, Cn, Cn8, dl, dl8, dlu, dlu8, dy, dy8, G, G8,
                                                       DA, DA8, dBacc, dB8, dCacc, dC8, dAv0, want_dA);
                        if (MODE == 1)
                            scanh_bwd_dir_state<true>(p.A[(int64_t)ec * N + n], n, Bn, Bn8, Cn, Cn8, dl, dl8, dlu, dlu8, dy, dy8, G, G8,
                                                      DA, DA8, dBacc, dB8, dCacc, dC8, dAv0, want_dA);
                        if (BI)
                            scanh_bwd_dir_state<true>(p.A_b[(int64_t)ec * N + n], n, Bn, Bn8, Cn, Cn8, dl, dl8, dlu, dlu8, dy, dy8, G,
                                                      G8, DA, DA8, dBacc, dB8, dCacc, dC8, dAv1, want_dA);
                    }
                    if (!(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            lds_write(dBt, a0, lds_read(dBt, a0) + lo2(dBacc[i]));
                            lds_write(dBt, a1, lds_read(dBt, a1) + hi2(dBacc[i]));
                            lds_write(dCt, a0, lds_read(dCt, a0) + lo2(dCacc[i]));
                            lds_write(dCt, a1, lds_read(dCt, a1) + hi2(dCacc[i]));
                        }
                        if (TAIL > 0) {
                            const vi a8 = pos8 + n * GE::SP;          // shared slot: owned by the last lane
                            lds_write_m(dBt, a8, lds_read(dBt, a8) + dB8, lane == WAVE - 1);
                            lds_write_m(dCt, a8, lds_read(dCt, a8) + dC8, lane == WAVE - 1);
                        }
                    }
                }
                if ((j & SCANH_BARRIER_MASK) == SCANH_BARRIER_MASK && !(p.flags & AUM_DBG_NO_STEP_BARRIER)) AUM_WG_BARRIER_IN_PHASE();
            }
            if (active && want_dA) {
                const vm mn = lane < N;
                const vi ln = vmin_i(lane, N - 1);
                gstore(ws + L.pA + ((int64_t)b * p.dim + e) * N, ln, sum_rows4(dAv0), mn);
                if (BI) gstore(ws + L.pAb + ((int64_t)b * p.dim + e) * N, ln, sum_rows4(dAv1), mn);
            }
            if (active && !(p.flags & AUM_DBG_SKIP_EPILOGUE)) {
                const float Dn = p.D ? ndir * p.D[e] : 0.f;
                vf2 uu[4], raw[4], duv[4], ddv[4];
                vf u8, raw8 = splat(0.f);
                scanh_row_read<T>(up, p.len, uu, u8);
                if (softplus) scanh_row_read<T>(dp, p.len, raw, raw8);
                vf2 dDl = spl2(splat(0.f)), dbl = spl2(splat(0.f));
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) {
                    duv[i] = vfma2(dl[i], G[i], dy[i] * spl2(splat(Dn)));
                    vf2 dd = vfma2(uu[i], G[i], DA[i]);
                    if (softplus) {
                        const vf2 rw = raw[i] + spl2(splat(bias));
                        const vf2 ds = dd * vsigmoid2(rw);
                        dd = mk2(vsel(lo2(rw) > 20.f, lo2(dd), lo2(ds)), vsel(hi2(rw) > 20.f, hi2(dd), hi2(ds)));
                    }
                    ddv[i] = dd;
                    dDl = vfma2(dy[i], uu[i], dDl);
                    dbl = dbl + dd;
                }
                vf du8 = vfma(dl8, G8, dy8 * Dn);
                vf dd8 = vfma(u8, G8, DA8);
                if (softplus) {
                    const vf r8 = raw8 + bias;
                    dd8 = vsel(r8 > 20.f, dd8, dd8 * vsigmoid(r8));
                }
                dd8 = vsel(tail_lane, dd8, splat(0.f));
                du8 = vsel(tail_lane, du8, splat(0.f));
                scanh_row_write<T>(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), p.len, duv, du8);
                scanh_row_write<T>(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds), p.len, ddv, dd8);
                const float sD = ndir * wave_sum(vfma(dy8, u8, lo2(dDl) + hi2(dDl)));
                const float sb = wave_sum(lo2(dbl) + hi2(dbl) + dd8);
                gstore(ws + L.pD + (int64_t)b * p.dim + e, spl_i(0), splat(sD), lane == 0);
                gstore(ws + L.pbias + (int64_t)b * p.dim + e, spl_i(0), splat(sb), lane == 0);
            }
        }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, SCANH_NW) {
        scanwg_store_tile<8, TAIL, SCANH_NW>(dBt, ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
        scanwg_store_tile<8, TAIL, SCANH_NW>(dCt, ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
    }
}

}  // namespace aum
