// scan_row_kernels.h -- the selective scan for the AuM row shape L = 513 (512 patches + the cls token), dstate <= 16: forward
// and backward, ONE row per wave, second generation of scan_half_kernels.h.
//
// What changed against scanwg_fwd<8,1> / scanh_bwd (measured on MI355X, B = 64, bf16: 0.56 / 1.44 ms, both VALU-bound at ~75 % vector
// ALU occupancy, ~2400 / ~6100 vector instructions per row):
//   * NO TAIL SLOT.  The 513th step used to be a ninth, un-packed slot of every lane (an identity step on 63 of them): one extra
//     instruction per four packed ones in every per-(state, direction) block.  Here the 512 main steps are the only per-lane
//     slots; the tail step of ALL (direction, state) pairs of a row is one set of vector instructions with lane 16*d + n holding
//     (direction slot d, state n): exp, multiply and accumulate once per row instead of once per state.  The tail couples to the
//     main steps through wave-uniform carries (v_readlane / v_writelane): the time-reversed direction starts from it (carry into
//     lane 63), the forward direction ends in it.
//   * LANE-ENTRY CHECKPOINT (`x_lane`, ABI 3).  The forward stores, per (row, direction, state), the state entering every lane's
//     8-step block (64 fp32 = one 256-byte line).  The backward reads it and runs the recurrence once, serially over the lane's
//     8 steps -- no first local pass, no wave scan for the states (the adjoint still needs both).  HBM is the resource this
//     kernel has to spare (7 % of peak before): +8 KB per row and direction pair, +0.8 GB per launch at B = 64.
//   * b_t = delta_t u_t B_t and c_t = dy_t C_t do not depend on the direction: computed once per state, used by both.
// Everything else is the design of scan_half_kernels.h: the two HALVES of a lane's 8 steps (i, 4+i) in the halves of a vf2
// (packed fp32 math at half the registers of a row pair), B/C tiles in LDS, dB/dC accumulated in LDS tiles by plain
// read-add-write under a rotated state order, per-workgroup partials reduced by k_scan_reduce, no atomics anywhere.
// Reference: SSI:37, 62-65, 499-507, 541-561 (selective_scan_cuda.fwd/.bwd call sites), SSI:86-152 (selective_scan_ref).
#pragma once
#include "scan_half_kernels.h"

namespace aum {

constexpr int SCANR_LEN = 513;                         // 64 lanes x 8 steps + the tail step
#ifndef AUM_SCANR_FWD_NW
#define AUM_SCANR_FWD_NW 8
#endif
#ifndef AUM_SCANR_FWD_MINW
#define AUM_SCANR_FWD_MINW 4
#endif
constexpr int SCANR_FWD_NW = AUM_SCANR_FWD_NW;         // forward: 8 waves, 2 tiles (74 KB) -> two workgroups per CU at <= 128 VGPRs
#ifndef AUM_SCANR_BWD_NW2
#define AUM_SCANR_BWD_NW2 12                           // waves per workgroup of the fused bidirectional backward
#endif
#ifndef AUM_SCANR_NOLOAD
#define AUM_SCANR_NOLOAD 0                             // timing experiments only: the backward does not read the checkpoint
#endif
AUM_HOSTDEV constexpr int scanr_bwd_nw(int mode) { return mode == 2 ? AUM_SCANR_BWD_NW2 : 16; }
AUM_HOSTDEV constexpr int scanr_bwd_rows(int mode) { return 8 * scanr_bwd_nw(mode); }
constexpr int SCANR_TACC = 128;                        // floats per wave of the tail dB/dC hand-off area
constexpr int scanr_fwd_lds_floats() { return 2 * ScanGeo<8, 1>::TILE; }
// B/C tile layout of the forward: steps i and 4 + i adjacent (scan_tile_slot<true>), so the two halves of a vf2 are one
// ds_read2_b32 -- the layout the one-row backward uses (AUM_SCANH_PAIRED).  Same-box A/B, three alternating runs: fused
// bidirectional (training) 0.539 -> 0.522 ms, one direction 0.388 -> 0.373 ms (profiles/r02_ab_fwd_paired.txt).
#ifndef AUM_SCANR_FWD_PAIRED
#define AUM_SCANR_FWD_PAIRED 1
#endif
constexpr bool SCANR_FWD_PAIRED = AUM_SCANR_FWD_PAIRED != 0;
// Rows per forward workgroup.  Unlike the one-row backward (scanh_rows_for: one workgroup per CU, so fewer and longer is better) the
// forward has two workgroups per CU and gains from MORE, shorter ones: one workgroup's tile load and row prologues overlap the
// other's state loops.  Sweep at B = 64, E = 1536 (fused bidirectional, training): 16 / 32 / 64 / 96 / 192 rows = 0.540 / 0.540 /
// 0.556 / 0.568 / 0.600 ms.  32 rows while that still gives >= 1024 workgroups, fewer for small batches.
// -DAUM_SCANR_FWD_ROWS=n overrides the count for sweeps.
#ifndef AUM_SCANR_FWD_ROWS
#define AUM_SCANR_FWD_ROWS 0
#endif
AUM_HOSTDEV int scanr_fwd_rows_for(int batch, int dim) {
    if (AUM_SCANR_FWD_ROWS) return AUM_SCANR_FWD_ROWS;
    int rows = 32;
    while (rows > SCANR_FWD_NW && (int64_t)batch * ((dim + rows - 1) / rows) < 1024) rows /= 2;
    return rows;
}
constexpr int scanr_bwd_lds_floats() { return 4 * ScanGeo<8, 1>::TILE + 16 * SCANR_TACC; }
AUM_HOSTDEV bool scanr_selected(int len, int dstate, uint32_t flags) {
    return len == SCANR_LEN && dstate <= SCANWG_MAX_N && !(flags & (AUM_SCAN_ROWPAIR | AUM_SCAN_GENERIC));
}
// floats of the lane-entry checkpoint: [batch][dim][directions][dstate][64]
AUM_HOSTDEV int64_t scanr_ckpt_floats(int batch, int dim, int dstate, bool bidir) {
    return (int64_t)batch * dim * (bidir ? 2 : 1) * dstate * WAVE;
}

// the 512 main steps of one row -> half-packed slots m[i] = (step 8*lane + i, step 8*lane + 4 + i)
template <class T> AUM_DEV void scanr_row_read(const T* rp, vf2 (&m)[4]) {
    const vi lane = lane_id();
    vf v[8];
    gload8(rp, lane * 8, lane >= 0, v);
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) m[i] = mk2(v[i], v[4 + i]);
}
template <class T> AUM_DEV void scanr_row_write(T* rp, const vf2 (&m)[4], float tail) {
    const vi lane = lane_id();
    vf v[8];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) { v[i] = lo2(m[i]); v[4 + i] = hi2(m[i]); }
    gstore8(rp, lane * 8, v, lane >= 0);
    gstore(rp, spl_i(SCANR_LEN - 1), splat(tail), lane == 0);
}

// x' = m_k x + b_k over the lane's 8 half-packed steps and the 64 lanes.  REV = false: steps 0..7, lanes 0 -> 63, zero state
// before lane 0.  REV = true: steps 7..0, lanes 63 -> 0, the wave-uniform `cin` before lane 63's first step (HAS_CIN).
// x[i] = state AFTER slots (i, 4+i); x_in = state entering the lane; S = state after the lane's last step.
// `cin` is a plain float or a vf holding the same value in every lane.
template <bool REV, bool HAS_CIN, class CIN>
AUM_DEV void scanr_affine(const vf2 (&m)[4], const vf2 (&b)[4], CIN cin, vf2 (&x)[4], vf& x_in, vf& S) {
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
    vf P = Plo * Phi;
    S = REV ? vfma(Plo, hi2(s), lo2(s)) : vfma(Phi, lo2(s), hi2(s));
    if (REV && HAS_CIN) S = vsel(lane_id() == WAVE - 1, vfma(P, splat(cin), S), S);
    wave_scan_affine<REV>(P, S);
    x_in = REV ? dpp_wave_shl1(S, HAS_CIN ? splat(cin) : splat(0.f)) : dpp_wave_shr1(S, splat(0.f));
    vf2 xx = REV ? mk2(vfma(Phi, x_in, hi2(s)), x_in) : mk2(x_in, vfma(Plo, x_in, lo2(s)));
    AUM_UNROLL
    for (int ii = 0; ii < 4; ++ii) {
        const int i = REV ? 3 - ii : ii;
        xx = vfma2(m[i], xx, b[i]);
        x[i] = xx;
    }
}

// the states of a lane's 8 steps from the checkpointed entry state: one serial chain, no scan
template <bool REV> AUM_DEV void scanr_states_from_entry(const vf2 (&a)[4], const vf2 (&b)[4], vf x_in, vf2 (&x)[4]) {
    vf lo[4], hi[4];
    vf c = x_in;
    if (!REV) {
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { c = vfma(lo2(a[i]), c, lo2(b[i])); lo[i] = c; }
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { c = vfma(hi2(a[i]), c, hi2(b[i])); hi[i] = c; }
    } else {
        AUM_UNROLL
        for (int i = 3; i >= 0; --i) { c = vfma(hi2(a[i]), c, hi2(b[i])); hi[i] = c; }
        AUM_UNROLL
        for (int i = 3; i >= 0; --i) { c = vfma(lo2(a[i]), c, lo2(b[i])); lo[i] = c; }
    }
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) x[i] = mk2(lo[i], hi[i]);
}

// direction slot d of a MODE: slot 0 = A, slot 1 = A_b (MODE 2 only); is the slot's recurrence time-reversed?
template <int MODE> AUM_HOSTDEV constexpr bool scanr_slot_rev(int d) { return MODE == 1 || (MODE == 2 && d == 1); }

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <class T, int MODE>
AUM_DEV void scanr_fwd(const AumScanFwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using GE = ScanGeo<8, 1>;
    constexpr int NW = SCANR_FWD_NW;
    constexpr int ND = MODE == 2 ? 2 : 1;
    constexpr int TCOL = WAVE * GE::LK;                     // tile word of the tail step
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + GE::TILE;
    const int gpb = (p.dim + rows_per_wg - 1) / rows_per_wg;
    const int b = wg / gpb;
    const int eb = (wg % gpb) * rows_per_wg;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);

    AUM_FOR_EACH_WAVE(w, NW) {
        scanwg_load_tile<T, 8, 1, NW, SCANR_FWD_PAIRED>(Bsrc, p.B_ns, N, 0, p.len, Bt, w);
        scanwg_load_tile<T, 8, 1, NW, SCANR_FWD_PAIRED>(Csrc, p.C_ns, N, 0, p.len, Ct, w);
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, NW) {
        const vi lane = lane_id();
        vi pos[4], pos4[4];
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) {         // steps i and 4 + i of the lane: adjacent tile words in the paired layout
            pos[i] = lane * GE::LK + scan_tile_slot<SCANR_FWD_PAIRED>(i);
            pos4[i] = lane * GE::LK + scan_tile_slot<SCANR_FWD_PAIRED>(4 + i);
        }
        // tail lanes: lane 16*d + n <-> (direction slot d, state n)
        const vi tn = vmin_i(lane & 15, N - 1);
        const vm tvalid = (lane < 16 * ND) && ((lane & 15) < N);
        const vm tfwd = MODE == 0 ? (lane >= 0) : MODE == 1 ? (lane < 0) : (lane < 16);
        const vf B_t = lds_read(Bt, tn * GE::SP + TCOL), C_t = lds_read(Ct, tn * GE::SP + TCOL);
        for (int rloc = w; rloc < rows_per_wg; rloc += NW) {
            const int e = eb + rloc;
            if (e >= p.dim) break;
            const float bias = p.delta_bias ? p.delta_bias[e] : 0.f;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)e * p.delta_ds);
            vf2 uu[4], dl[4], dlu[4], y[4];
            float u_t, dl_t;
            {
                vf2 dd[4];
                scanr_row_read<T>(up, uu);
                scanr_row_read<T>(dp, dd);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const vf2 dr = dd[i] + spl2(splat(bias));
                    dl[i] = softplus ? vsoftplus2(dr) : dr;
                    dlu[i] = dl[i] * uu[i];
                    y[i] = spl2(splat(0.f));
                }
                u_t = gload_s(up, SCANR_LEN - 1);
                dl_t = gload_s(dp, SCANR_LEN - 1) + bias;
                if (softplus) dl_t = vsoftplus(dl_t);
            }
            // tail step of every (direction, state): a_t = exp(delta_512 A), b_t = delta_512 u_512 B_512
            vf A_t = gload(p.A + (int64_t)e * N, tn, tvalid && (lane < 16));
            if (MODE == 2) A_t = A_t + gload(p.A_b + (int64_t)e * N, tn, tvalid && (lane >= 16));
            const vf a_t = vexp2(A_t * splat(dl_t * LOG2E));
            const vf b_t = B_t * splat(dl_t * u_t);
            vf xm = splat(0.f);          // lane n <- state n after the 512 main steps (forward-time slot)
            vf lastv = splat(0.f);       // lane n <- state n after the last step of a time-reversed single-direction call
            float* ck = p.x_lane ? p.x_lane + ((int64_t)b * p.dim + e) * ND * N * WAVE : nullptr;
            if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                for (int n = 0; n < N; ++n) {
                    vf2 Cn[4], bb[4];
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        bb[i] = dlu[i] * mk2(lds_read(Bt, pos[i] + n * GE::SP), lds_read(Bt, pos4[i] + n * GE::SP));
                        Cn[i] = mk2(lds_read(Ct, pos[i] + n * GE::SP), lds_read(Ct, pos4[i] + n * GE::SP));
                    }
                    AUM_UNROLL
                    for (int d = 0; d < ND; ++d) {
                        constexpr bool REV0 = scanr_slot_rev<MODE>(0);
                        const float An = (d == 0 ? p.A : p.A_b)[(int64_t)e * N + n] * LOG2E;
                        vf2 a[4], x[4];
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) a[i] = vexp2_2(dl[i] * spl2(splat(An)));
                        vf x_in, S;
                        if (d == 0 && !REV0) {
                            scanr_affine<false, false>(a, bb, 0.f, x, x_in, S);
                            xm = writelane(xm, readlane(S, WAVE - 1), n);
                        } else {     // time-reversed: the tail step comes first, x_512 = b_512
                            scanr_affine<true, true>(a, bb, readlane(b_t, 16 * d + n), x, x_in, S);
                            if (p.last_state) lastv = writelane(lastv, readlane(S, 0), n);
                        }
                        if (ck) gstore(ck + ((int64_t)d * N + n) * WAVE, lane, x_in, lane >= 0);
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) y[i] = vfma2(Cn[i], x[i], y[i]);
                    }
                }
            }
            // tail: x_512 = a_512 x_511 + b_512 (forward-time slot) or b_512 (time-reversed slot); y_512 = sum C_512 x_512
            const vf x_t = vsel(tfwd, vfma(a_t, xm, b_t), b_t);
            const float y_t = wave_sum(vsel(tvalid, C_t * x_t, splat(0.f)));
            if (p.last_state) {
                float* lp = p.last_state + ((int64_t)b * p.dim + e) * N;
                gstore(lp, tn, MODE == 0 ? x_t : lastv, (lane < 16) && ((lane & 15) < N));
            }
            const float Dn = p.D ? (float)ND * p.D[e] : 0.f;
            const int64_t ooff = (int64_t)b * p.out_bs + (int64_t)e * p.out_ds;
            vf2 o[4];
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) o[i] = vfma2(uu[i], spl2(splat(Dn)), y[i]);
            float o_t = vfma(u_t, Dn, y_t);
            if (p.out_pre) scanr_row_write<T>(row_ptr_w<T>(p.out_pre, ooff), o, o_t);
            if (p.z) {
                const T* zp = row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)e * p.z_ds);
                vf2 zz[4];
                scanr_row_read<T>(zp, zz);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) o[i] = o[i] * (zz[i] * vsigmoid2(zz[i]));
                const float z_t = gload_s(zp, SCANR_LEN - 1);
                o_t = o_t * (z_t * vsigmoid(z_t));
            }
            scanr_row_write<T>(row_ptr_w<T>(p.out, ooff), o, o_t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------
// one (state n, direction slot) of one row: states from the checkpoint, adjoint scan, the five accumulations.
//   REV = false (forward-time recurrence): the adjoint runs lanes 63 -> 0; it enters lane 63 as g_512 (`gcin`) through the tail
//         step's multiplier `a_edge` = a_512.  Returns x_511 (`x_last`).
//   REV = true: the adjoint runs lanes 0 -> 63 from zero; returns a_511 g_511 (`ga_last`), the tail step's adjoint input.
template <bool REV>
AUM_DEV void scanr_bwd_dir_state(float Araw, int n, vf x_in, const vf2 (&Bn)[4], const vf2 (&bb)[4], const vf2 (&cc)[4], const vf2 (&dl)[4],
                                 const vf2 (&dlu)[4], const vf2 (&dy)[4], float a_edge, float gcin, vf2 (&G)[4], vf2 (&DA)[4],
                                 vf2 (&dBacc)[4], vf2 (&dCacc)[4], vf& dAv, float& x_last, float& ga_last) {
    const float An = Araw * LOG2E;
    vf2 a[4], x[4], m[4], g[4];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) a[i] = vexp2_2(dl[i] * spl2(splat(An)));
    scanr_states_from_entry<REV>(a, bb, x_in, x);
    // adjoint g_k = dy_k C_k + a_succ(k) g_succ(k): the multiplier of a slot is the `a` of its scan successor
    vf gin, gS;
    if (!REV) {
        m[0] = a[1]; m[1] = a[2]; m[2] = a[3];
        m[3] = mk2(hi2(a[0]), dpp_wave_shl1(lo2(a[0]), splat(a_edge)));
        scanr_affine<true, true>(m, cc, gcin, g, gin, gS);
        x_last = readlane(hi2(x[3]), WAVE - 1);
    } else {
        m[0] = mk2(dpp_wave_shr1(hi2(a[3]), splat(1.f)), lo2(a[3]));
        m[1] = a[0]; m[2] = a[1]; m[3] = a[2];
        scanr_affine<false, false>(m, cc, 0.f, g, gin, gS);
    }
    (void)gin; (void)gS;
    vf2 dAl = spl2(splat(0.f));
    const vf2 Ar = spl2(splat(Araw));
    vf2 ga3 = spl2(splat(0.f));
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        vf2 xprev;
        if (!REV) xprev = i == 0 ? mk2(x_in, lo2(x[3])) : x[i > 0 ? i - 1 : 0];
        else xprev = i == 3 ? mk2(hi2(x[0]), x_in) : x[i < 3 ? i + 1 : 3];
        const vf2 ga = g[i] * a[i];
        if (i == 3) ga3 = ga;
        const vf2 h = ga * xprev;
        G[i] = vfma2(g[i], Bn[i], G[i]);
        DA[i] = vfma2(Ar, h, DA[i]);
        dBacc[i] = vfma2(g[i], dlu[i], dBacc[i]);
        dCacc[i] = vfma2(dy[i], x[i], dCacc[i]);
        dAl = vfma2(dl[i], h, dAl);
    }
    if (REV) ga_last = readlane(hi2(ga3), WAVE - 1);
    // dA[e][n] = sum over the row: reduced inside each 16-lane row now (lane 16q + n keeps row q's share), the four shares are
    // added once per row after the state loop (sum_rows4)
    dAv = vsel((lane_id() & 15) == n, row_sum16(lo2(dAl) + hi2(dAl)), dAv);
}

template <class T, int MODE>
AUM_DEV void scanr_bwd(const AumScanBwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using GE = ScanGeo<8, 1>;
    constexpr bool BI = MODE == 2;
    constexpr int ND = BI ? 2 : 1;
    constexpr int NW = scanr_bwd_nw(MODE);
    constexpr int ROT = SCANWG_MAX_N / NW >= 2 ? SCANWG_MAX_N / NW : 1;
    constexpr int BARRIER_MASK = ROT >= 2 ? 1 : 0;      // barrier after step j when (j & mask) == mask
    constexpr int TCOL = WAVE * GE::LK;
    constexpr bool REV0 = scanr_slot_rev<MODE>(0);
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + GE::TILE;
    float* dBt = lds + 2 * GE::TILE;
    float* dCt = lds + 3 * GE::TILE;
    float* tacc = lds + 4 * GE::TILE;                   // [NW][SCANR_TACC]: per-wave tail dB (lanes) | tail dC (64 + lanes)
    const ScanWgWs L = scanwg_ws_layout(p.batch, p.dim, p.len, N, rows_per_wg, 1, BI);
    float* ws = (float*)p.workspace;
    const int b = wg / L.gpb, g_idx = wg % L.gpb;
    const int eb = g_idx * rows_per_wg;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const float ndir = BI ? 2.f : 1.f;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);

    AUM_FOR_EACH_WAVE(w, NW) {
        scanwg_load_tile<T, 8, 1, NW, false>(Bsrc, p.B_ns, N, 0, p.len, Bt, w);
        scanwg_load_tile<T, 8, 1, NW, false>(Csrc, p.C_ns, N, 0, p.len, Ct, w);
        for (int i0 = w * WAVE; i0 < GE::TILE; i0 += NW * WAVE) {
            const vi idx = lane_id() + i0;
            lds_write_m(dBt, idx, splat(0.f), idx < GE::TILE);
            lds_write_m(dCt, idx, splat(0.f), idx < GE::TILE);
        }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, NW) {
        const vi lane = lane_id();
        vi pos[4], pos4[4];
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { pos[i] = lane * GE::LK + i; pos4[i] = lane * GE::LK + 4 + i; }
        const vi tn = vmin_i(lane & 15, N - 1);
        const vm tvalid = (lane < 16 * ND) && ((lane & 15) < N);
        const vm tfwd = MODE == 0 ? (lane >= 0) : MODE == 1 ? (lane < 0) : (lane < 16);
        const vf B_t = lds_read(Bt, tn * GE::SP + TCOL), C_t = lds_read(Ct, tn * GE::SP + TCOL);
        vf tdB = splat(0.f), tdC = splat(0.f);          // tail dB / dC of this wave's rows, lane 16*d + n
        const int niter = (rows_per_wg + NW - 1) / NW;
        for (int it = 0; it < niter; ++it) {
            const int rloc = w + it * NW;
            const int e = eb + rloc;
            const bool active = rloc < rows_per_wg && e < p.dim;   // wave-uniform; inactive waves still take every barrier below
            const int ec = active ? e : p.dim - 1;
            const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
            const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds);
            const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds);
            vf2 dl[4], dlu[4], dy[4], G[4], DA[4];
            float u_t, raw_t, dl_t, dy_t;
            {   // delta = softplus(delta + bias), delta * u
                vf2 uu[4], dd[4];
                scanr_row_read<T>(up, uu);
                scanr_row_read<T>(dp, dd);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const vf2 dr = dd[i] + spl2(splat(bias));
                    dl[i] = softplus ? vsoftplus2(dr) : dr;
                    dlu[i] = dl[i] * uu[i];
                    G[i] = spl2(splat(0.f));
                    DA[i] = spl2(splat(0.f));
                }
                u_t = gload_s(up, SCANR_LEN - 1);
                raw_t = gload_s(dp, SCANR_LEN - 1) + bias;
                dl_t = softplus ? vsoftplus(raw_t) : raw_t;
            }
            {   // dout (and the gate): dy = dout * silu(z), dz = dout * out_pre * silu'(z)
                const T* gop = row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds);
                vf2 go[4];
                scanr_row_read<T>(gop, go);
                float go_t = gload_s(gop, SCANR_LEN - 1);
                if (p.z) {
                    const T* zp = row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)ec * p.z_ds);
                    const T* ypp = row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)ec * p.out_ds);
                    vf2 zz[4], yp[4], dzv[4];
                    scanr_row_read<T>(zp, zz);
                    scanr_row_read<T>(ypp, yp);
                    const vf2 one2 = spl2(splat(1.f));
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        const vf2 sg = vsigmoid2(zz[i]);
                        dzv[i] = go[i] * yp[i] * sg * vfma2(zz[i], one2 - sg, one2);
                        go[i] = go[i] * zz[i] * sg;
                    }
                    const float z_t = gload_s(zp, SCANR_LEN - 1), yp_t = gload_s(ypp, SCANR_LEN - 1);
                    const float sg = vsigmoid(z_t);
                    const float dz_t = go_t * yp_t * sg * vfma(z_t, 1.f - sg, 1.f);
                    go_t = go_t * z_t * sg;
                    if (active) scanr_row_write<T>(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)ec * p.dz_ds), dzv, dz_t);
                }
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) dy[i] = go[i];
                dy_t = go_t;
            }
            // tail step of every (direction slot, state), lane 16*d + n
            vf A_t = gload(p.A + (int64_t)ec * N, tn, tvalid && (lane < 16));
            if (BI) A_t = A_t + gload(p.A_b + (int64_t)ec * N, tn, tvalid && (lane >= 16));
            const vf a_t = vexp2(A_t * splat(dl_t * LOG2E));
            const vf b_t = B_t * splat(dl_t * u_t);
            const vf cc_t = C_t * splat(dy_t);
            vf xm = splat(0.f);          // lane 16*d + n <- x_511 of a forward-time slot
            vf gm = splat(0.f);          // lane 16*d + n <- a_511 g_511 of a time-reversed slot
            vf dAv0 = splat(0.f), dAv1 = splat(0.f);          // lane n <- dA (dA_b) partial of state n
            const float* ck = p.x_lane + ((int64_t)b * p.dim + ec) * ND * N * WAVE;
            // rotated state order: see scanwg_bwd -- no two waves hold the same dB/dC tile row in the same or adjacent steps
            vf xin_next[ND];
            {
                const int n0 = (ROT * w) & (SCANWG_MAX_N - 1);
                AUM_UNROLL
                for (int d = 0; d < ND; ++d) xin_next[d] = gload(ck + ((int64_t)d * N + (n0 < N ? n0 : 0)) * WAVE, lane, lane >= 0);
            }
            for (int j = 0; j < SCANWG_MAX_N; ++j) {
                const int n = (j + ROT * w) & (SCANWG_MAX_N - 1);
                const int nn = (j + 1 + ROT * w) & (SCANWG_MAX_N - 1);
                vf xin_cur[ND];
                AUM_UNROLL
                for (int d = 0; d < ND; ++d) {
                    xin_cur[d] = xin_next[d];
                    if (!AUM_SCANR_NOLOAD) xin_next[d] = gload(ck + ((int64_t)d * N + (nn < N ? nn : 0)) * WAVE, lane, lane >= 0);    // one state ahead
                }
                if (active && n < N && !(p.flags & AUM_DBG_SKIP_STATES)) {
                    vf2 Bn[4], bb[4], cc[4], dBacc[4], dCacc[4];
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        Bn[i] = mk2(lds_read(Bt, pos[i] + n * GE::SP), lds_read(Bt, pos4[i] + n * GE::SP));
                        cc[i] = dy[i] * mk2(lds_read(Ct, pos[i] + n * GE::SP), lds_read(Ct, pos4[i] + n * GE::SP));
                        bb[i] = dlu[i] * Bn[i];
                        dBacc[i] = spl2(splat(0.f));
                        dCacc[i] = spl2(splat(0.f));
                    }
                    float xl = 0.f, gal = 0.f;
                    if (!REV0) {
                        scanr_bwd_dir_state<false>(p.A[(int64_t)ec * N + n], n, xin_cur[0], Bn, bb, cc, dl, dlu, dy, readlane(a_t, n),
                                                   readlane(cc_t, n), G, DA, dBacc, dCacc, dAv0, xl, gal);
                        xm = writelane(xm, xl, n);
                    } else {
                        scanr_bwd_dir_state<true>(p.A[(int64_t)ec * N + n], n, xin_cur[0], Bn, bb, cc, dl, dlu, dy, 1.f, 0.f, G, DA, dBacc,
                                                  dCacc, dAv0, xl, gal);
                        gm = writelane(gm, gal, n);
                    }
                    if (BI) {
                        scanr_bwd_dir_state<true>(p.A_b[(int64_t)ec * N + n], n, xin_cur[ND - 1], Bn, bb, cc, dl, dlu, dy, 1.f, 0.f, G, DA,
                                                  dBacc, dCacc, dAv1, xl, gal);
                        gm = writelane(gm, gal, 16 + n);
                    }
                    if (!(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                        // dB/dC tile rows of state n: all reads, then the adds, then the writes (one LDS round trip in the chain)
                        vf rb[8], rc[8];
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            rb[i] = lds_read(dBt, a0);
                            rb[4 + i] = lds_read(dBt, a1);
                            rc[i] = lds_read(dCt, a0);
                            rc[4 + i] = lds_read(dCt, a1);
                        }
                        AUM_SCHED_FENCE();
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            rb[i] = rb[i] + lo2(dBacc[i]);
                            rb[4 + i] = rb[4 + i] + hi2(dBacc[i]);
                            rc[i] = rc[i] + lo2(dCacc[i]);
                            rc[4 + i] = rc[4 + i] + hi2(dCacc[i]);
                        }
                        AUM_SCHED_FENCE();
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            lds_write(dBt, a0, rb[i]);
                            lds_write(dBt, a1, rb[4 + i]);
                            lds_write(dCt, a0, rc[i]);
                            lds_write(dCt, a1, rc[4 + i]);
                        }
                    }
                }
                // LDS-only barrier: the tiles are all the waves share, and __syncthreads() would drain vmcnt -- wait at every step
                // for the checkpoint prefetch issued a few instructions earlier
                if ((j & BARRIER_MASK) == BARRIER_MASK && !(p.flags & AUM_DBG_NO_STEP_BARRIER)) AUM_WG_BARRIER_LDS();
            }
            // ---- tail step: adjoint, gradients of u_512 / delta_512 / z_512, tail dB / dC / dA ----
            const vf x_t = vsel(tfwd, vfma(a_t, xm, b_t), b_t);
            const vf g_t = vsel(tfwd, cc_t, cc_t + gm);
            const vf h_t = vsel(tfwd && tvalid, g_t * a_t * xm, splat(0.f));
            const float G_t = wave_sum(vsel(tvalid, g_t * B_t, splat(0.f)));
            const float DA_t = wave_sum(A_t * h_t);
            if (active) {
                tdB = tdB + vsel(tvalid, g_t * splat(dl_t * u_t), splat(0.f));
                tdC = tdC + vsel(tvalid, x_t * splat(dy_t), splat(0.f));
            }
            if (!REV0) dAv0 = dAv0 + vsel(lane < 16, h_t * splat(dl_t), splat(0.f));
            if (active && !(p.flags & AUM_DBG_SKIP_PARTIALS)) {
                const vm mn = lane < N;
                const vi ln = vmin_i(lane, N - 1);
                gstore(ws + L.pA + ((int64_t)b * p.dim + e) * N, ln, sum_rows4(dAv0), mn);
                if (BI) gstore(ws + L.pAb + ((int64_t)b * p.dim + e) * N, ln, sum_rows4(dAv1), mn);
            }
            if (active && !(p.flags & AUM_DBG_SKIP_EPILOGUE)) {
                const float Dn = p.D ? ndir * p.D[e] : 0.f;
                vf2 uu[4], raw[4], duv[4], ddv[4];
                scanr_row_read<T>(up, uu);
                if (softplus) scanr_row_read<T>(dp, raw);
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
                const float du_t = vfma(dl_t, G_t, dy_t * Dn);
                float dd_t = vfma(u_t, G_t, DA_t);
                if (softplus && !(raw_t > 20.f)) dd_t = dd_t * vsigmoid(raw_t);
                scanr_row_write<T>(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), duv, du_t);
                scanr_row_write<T>(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds), ddv, dd_t);
                const float sD = ndir * (wave_sum(lo2(dDl) + hi2(dDl)) + dy_t * u_t);
                const float sb = wave_sum(lo2(dbl) + hi2(dbl)) + dd_t;
                gstore(ws + L.pD + (int64_t)b * p.dim + e, spl_i(0), splat(sD), lane == 0);
                gstore(ws + L.pbias + (int64_t)b * p.dim + e, spl_i(0), splat(sb), lane == 0);
            }
        }
        lds_write(tacc, lane + w * SCANR_TACC, tdB);
        lds_write(tacc, lane + (w * SCANR_TACC + WAVE), tdC);
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, NW) {
        if (w == 0) {      // tail column of the dB/dC tiles: sum over the waves and the direction slots
            const vi lane = lane_id();
            vf sB = splat(0.f), sC = splat(0.f);
            for (int ww = 0; ww < NW; ++ww) {
                sB = sB + lds_read(tacc, lane + ww * SCANR_TACC);
                sC = sC + lds_read(tacc, lane + (ww * SCANR_TACC + WAVE));
            }
            if (BI) {
                sB = sB + lane_gather(sB, (lane + 16) & (WAVE - 1));
                sC = sC + lane_gather(sC, (lane + 16) & (WAVE - 1));
            }
            const vi tn = vmin_i(lane, N - 1);
            lds_write_m(dBt, tn * GE::SP + TCOL, sB, lane < N);
            lds_write_m(dCt, tn * GE::SP + TCOL, sC, lane < N);
        }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, NW) {
        scanwg_store_tile<8, 1, NW, false>(dBt, ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
        scanwg_store_tile<8, 1, NW, false>(dCt, ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
    }
}

}  // namespace aum
