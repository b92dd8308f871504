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
// Rows per workgroup sized to the launch.  Only one workgroup fits a CU (four LDS tiles), each pays for the B/C tile load and the
// dB/dC tile flush of its batch entry, and every workgroup leaves one dB/dC partial for the reduce kernel: so as few, as long
// workgroups as still cover the chip -- one per CU when the launch is large enough (B = 64, E = 1536: 384 rows, 256 workgroups, 4
// partials per batch entry instead of 16), down to one row per wave for small batches (B = 8: 48 rows, 256 workgroups, where
// the fixed 96 left half of the CUs idle).  -DAUM_SCANH_ROWS_FIXED=1 restores the fixed count for A/B runs, =n (> 1) forces n rows.
#ifndef AUM_SCANH_ROWS_FIXED
#define AUM_SCANH_ROWS_FIXED 0
#endif
constexpr int SCANH_NUM_CU = 256;
constexpr int SCANH_MAX_ROWS = 384;
AUM_HOSTDEV int scanh_rows_for(int batch, int dim, int mode) {
    if (AUM_SCANH_ROWS_FIXED) return AUM_SCANH_ROWS_FIXED > 1 ? AUM_SCANH_ROWS_FIXED : scanh_rows(mode);
    const int nw = scanh_nw(mode);
    int64_t r = (int64_t)batch * dim / SCANH_NUM_CU / nw * nw;
    if (r < nw) r = nw;
    if (r > SCANH_MAX_ROWS) r = SCANH_MAX_ROWS;
    return (int)r;
}
// Tile layout: a lane's steps i and 4+i adjacent in the LDS tiles, so the B/C reads and the dB/dC read-add-write move whole vf2
// values (ds_read2_b32 / ds_write2_b32) and the 12 v_mov_b32 per state that re-pair the (i, i+1) reads of the plain layout
// disappear.  Round 1 measured it 1.5 % slower (before the batched tile update); on top of the batched update it is 0.5 % (fused
// bidirectional) / 1.2 % (one direction) faster in three alternating same-box runs, gradients bitwise equal
// (profiles/r02_ab_paired_batched.txt) -> on.  -DAUM_SCANH_PAIRED=0 restores the plain layout.
#ifndef AUM_SCANH_PAIRED
#define AUM_SCANH_PAIRED 1
#endif
constexpr bool SCANH_PAIRED = AUM_SCANH_PAIRED != 0;
// Opt-in (-DAUM_SCANH_RMW_BATCH=1): the dB/dC tile update of the fused bidirectional kernel issues all 18 LDS reads, then adds,
// then writes -- one LDS round trip in the latency chain between the arithmetic and the step barrier instead of several.
#ifndef AUM_SCANH_RMW_BATCH
#define AUM_SCANH_RMW_BATCH 1
#endif
constexpr bool SCANH_RMW_BATCH = AUM_SCANH_RMW_BATCH != 0;
// tile words of a lane's packed slot i: .x = step i, .y = step 4+i
AUM_HOSTDEV constexpr int scanh_slot_lo(int i) { return SCANH_PAIRED ? 2 * i : i; }
AUM_HOSTDEV constexpr int scanh_slot_hi(int i) { return SCANH_PAIRED ? 2 * i + 1 : 4 + i; }
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
// CARRY: `cin` is the (wave-uniform) state before the first step of the chunk, `cout` the state after its last one.
template <bool REV, bool CARRY>
AUM_DEV void scanh_affine_c(const vf2 (&m)[4], vf m8, const vf2 (&b)[4], vf b8, vf cin, vf2 (&x)[4], vf& x8, vf& x_in, vf& cout) {
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
    if (CARRY) S = vsel(lane_id() == (REV ? WAVE - 1 : 0), vfma(P, cin, S), S);
    wave_scan_affine<REV>(P, S);
    const vf fill = CARRY ? cin : splat(0.f);
    x_in = REV ? dpp_wave_shl1(S, fill) : dpp_wave_shr1(S, fill);
    if (CARRY) cout = splat(readlane(S, REV ? 0 : WAVE - 1));
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
template <bool REV> AUM_DEV void scanh_affine(const vf2 (&m)[4], vf m8, const vf2 (&b)[4], vf b8, vf2 (&x)[4], vf& x8, vf& x_in) {
    vf unused;
    scanh_affine_c<REV, false>(m, m8, b, b8, splat(0.f), x, x8, x_in, unused);
}

// one (state n, direction) of one row: forward states, adjoint, and the five accumulations
// CARRY (chunked rows): xcin = state entering the chunk in scan order, gcin = adjoint entering it in adjoint order, a_edge =
// multiplier of the step that follows the chunk in time (forward direction only; the reverse direction's carry already
// holds it), gcout = adjoint leaving the chunk.
template <bool REV, bool CARRY>
AUM_DEV void scanh_bwd_dir_state_c(float Araw, int n, const vf2 (&Bn)[4], vf Bn8, const vf2 (&Cn)[4], vf Cn8, const vf2 (&dl)[4], vf dl8,
                                   const vf2 (&dlu)[4], vf dlu8, const vf2 (&dy)[4], vf dy8, vf2 (&G)[4], vf& G8, vf2 (&DA)[4], vf& DA8,
                                   vf2 (&dBacc)[4], vf& dB8, vf2 (&dCacc)[4], vf& dC8, vf& dAv, bool want_dA, vf xcin, vf gcin,
                                   vf a_edge, vf& gcout) {
    const float An = Araw * LOG2E;
    vf2 a[4], bb[4], x[4], cc[4], m[4], g[4];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        a[i] = vexp2_2(dl[i] * spl2(splat(An)));
        bb[i] = dlu[i] * Bn[i];
        cc[i] = dy[i] * Cn[i];
    }
    const vf a8 = vexp2(dl8 * An), bb8 = dlu8 * Bn8, cc8 = dy8 * Cn8;
    vf x8, xin, g8, gin, xcout;
    scanh_affine_c<REV, CARRY>(a, a8, bb, bb8, xcin, x, x8, xin, xcout);
    (void)xcout;
    // adjoint g_k = dy_k C_k + a_succ(k) * g_succ(k), scanned against the recurrence; the multiplier of a slot is the `a` of its
    // scan successor (the neighbour lane's first slot at the lane edge; 1 past the end of the row)
    vf m8;
    if (!REV) {
        m[0] = a[1]; m[1] = a[2]; m[2] = a[3];
        m[3] = mk2(hi2(a[0]), a8);
        m8 = dpp_wave_shl1(lo2(a[0]), CARRY ? a_edge : splat(1.f));
    } else {
        m[0] = mk2(dpp_wave_shr1(a8, splat(1.f)), lo2(a[3]));
        m[1] = a[0]; m[2] = a[1]; m[3] = a[2];
        m8 = hi2(a[3]);
    }
    scanh_affine_c<!REV, CARRY>(m, m8, cc, cc8, gcin, g, g8, gin, gcout);
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

template <bool REV>
AUM_DEV void scanh_bwd_dir_state(float Araw, int n, const vf2 (&Bn)[4], vf Bn8, const vf2 (&Cn)[4], vf Cn8, const vf2 (&dl)[4], vf dl8,
                                 const vf2 (&dlu)[4], vf dlu8, const vf2 (&dy)[4], vf dy8, vf2 (&G)[4], vf& G8, vf2 (&DA)[4], vf& DA8,
                                 vf2 (&dBacc)[4], vf& dB8, vf2 (&dCacc)[4], vf& dC8, vf& dAv, bool want_dA) {
    vf unused;
    scanh_bwd_dir_state_c<REV, false>(Araw, n, Bn, Bn8, Cn, Cn8, dl, dl8, dlu, dlu8, dy, dy8, G, G8, DA, DA8, dBacc, dB8, dCacc, dC8, dAv,
                                      want_dA, splat(0.f), splat(0.f), splat(1.f), unused);
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
        scanwg_load_tile<T, 8, TAIL, SCANH_NW, SCANH_PAIRED>(Bsrc, p.B_ns, N, 0, p.len, Bt, w);
        scanwg_load_tile<T, 8, TAIL, SCANH_NW, SCANH_PAIRED>(Csrc, p.C_ns, N, 0, p.len, Ct, w);
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
        for (int i = 0; i < 4; ++i) { pos[i] = lane * GE::LK + scanh_slot_lo(i); pos4[i] = lane * GE::LK + scanh_slot_hi(i); }
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
                    if (SCANH_RMW_BATCH && !(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                        vf rb[9], rc[9];
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            rb[i] = lds_read(dBt, a0);
                            rb[4 + i] = lds_read(dBt, a1);
                            rc[i] = lds_read(dCt, a0);
                            rc[4 + i] = lds_read(dCt, a1);
                        }
                        rb[8] = lds_read(dBt, pos8 + n * GE::SP);
                        rc[8] = lds_read(dCt, pos8 + n * GE::SP);
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
                        if (TAIL > 0) {
                            const vi a8 = pos8 + n * GE::SP;
                            lds_write_m(dBt, a8, rb[8] + dB8, lane == WAVE - 1);
                            lds_write_m(dCt, a8, rc[8] + dC8, lane == WAVE - 1);
                        }
                    } else if (!(p.flags & AUM_DBG_SKIP_LDS_ATOMICS)) {
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            if constexpr (SCANH_PAIRED) {      // adjacent words: one packed add per pair
                                const vf2 tb = mk2(lds_read(dBt, a0), lds_read(dBt, a1)) + dBacc[i];
                                const vf2 tc = mk2(lds_read(dCt, a0), lds_read(dCt, a1)) + dCacc[i];
                                lds_write(dBt, a0, lo2(tb));
                                lds_write(dBt, a1, hi2(tb));
                                lds_write(dCt, a0, lo2(tc));
                                lds_write(dCt, a1, hi2(tc));
                            } else {
                                lds_write(dBt, a0, lds_read(dBt, a0) + lo2(dBacc[i]));
                                lds_write(dBt, a1, lds_read(dBt, a1) + hi2(dBacc[i]));
                                lds_write(dCt, a0, lds_read(dCt, a0) + lo2(dCacc[i]));
                                lds_write(dCt, a1, lds_read(dCt, a1) + hi2(dCacc[i]));
                            }
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
        scanwg_store_tile<8, TAIL, SCANH_NW, SCANH_PAIRED>(dBt, ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
        scanwg_store_tile<8, TAIL, SCANH_NW, SCANH_PAIRED>(dCt, ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len, N, 0, p.len, w);
    }
}

// ------------------------------------------------------------------------------------------------
// Chunked rows (long-form clips: 512*m patches + the cls token, L = 1025, 2049, 4097, ...), ONE direction per launch.
// Same one-row lane layout, 512 main steps per chunk; the single tail step belongs to the last chunk (an identity slot in
// the others).  Opposite carries make direction fusion impossible here: the states enter a chunk from the scan side, the
// adjoint from the other side, so the host issues one launch per direction (as it does for the row-pair kernels).
//   pre-pass   chunks in scan order: lane totals only -> state entering every chunk (workspace `xck`, carry in LDS);
//              skipped when the caller hands in the forward's checkpoint (`x_ck`, written by scanwg_fwd_ct)
//   main pass  chunks in adjoint order: states + adjoint with carries, per-chunk du / ddelta / dz, dB/dC tile per chunk
// dA / dD / ddelta_bias partials of a row accumulate in their workspace slot across chunks (same wave every time).
// 128 VGPRs -> 16 waves, 64 rows per workgroup, rotation offset 1 with a barrier after every state step.
// ------------------------------------------------------------------------------------------------
constexpr int SCANH_CH_NW = 16, SCANH_CH_ROWS = 64;
template <int TAIL> constexpr int scanh_chunked_lds_floats() { return 4 * ScanGeo<8, TAIL>::TILE + 2 * SCANH_CH_ROWS * SCANWG_MAX_N; }
// rows per workgroup (a multiple of the 16 waves, at most SCANH_CH_ROWS): long-form batches are small, so the grid is sized
// to the 256 CUs -- B = 8, E = 1536: 64 rows would leave a quarter of the chip idle (192 workgroups), 48 rows give 256
#ifndef AUM_SCANH_CH_ROWS_FORCE
#define AUM_SCANH_CH_ROWS_FORCE 0      // > 0: forced row count (sweeps)
#endif
AUM_HOSTDEV int scanh_chunked_rows(int batch, int dim) {
    if (AUM_SCANH_CH_ROWS_FORCE) return AUM_SCANH_CH_ROWS_FORCE;
    int best = SCANH_CH_ROWS;
    long best_cost = -1;
    for (int rows = SCANH_CH_ROWS; rows >= SCANH_CH_NW; rows -= SCANH_CH_NW) {
        const long grid = (long)batch * ((dim + rows - 1) / rows);
        const long cost = ((grid + 255) / 256) * rows;          // rounds over the chip x rows a wave walks per round
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rows; }
    }
    return best;
}
AUM_HOSTDEV bool scanh_chunked_selected(int len, int dstate, int mode, uint32_t flags) {
    return mode != 2 && !(flags & AUM_SCAN_ROWPAIR) && len >= 1024 && (len & 511) <= 1 && dstate <= SCANWG_MAX_N;
}

template <class T> AUM_DEV void scanh_chunk_read(const T* rp, bool has_tail, vf2 (&m)[4], vf& tl) {
    const vi lane = lane_id();
    vf v[8];
    gload8(rp, lane * 8, lane >= 0, v);
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) m[i] = mk2(v[i], v[4 + i]);
    tl = splat(0.f);
    if (has_tail) tl = vsel(lane == WAVE - 1, gload_u(rp, spl_i(512)), splat(0.f));
}
template <class T> AUM_DEV void scanh_chunk_write(T* rp, bool has_tail, const vf2 (&m)[4], vf tl) {
    const vi lane = lane_id();
    vf v[8];
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) { v[i] = lo2(m[i]); v[4 + i] = hi2(m[i]); }
    gstore8(rp, lane * 8, v, lane >= 0);
    if (has_tail) gstore(rp, spl_i(512), tl, lane == WAVE - 1);
}

// delta = softplus(delta + bias), delta * u of one chunk of one row (tail slot: identity unless the chunk owns the tail)
template <class T>
AUM_DEV void scanh_chunk_delta(const T* up, const T* dp, float bias, bool softplus, bool has_tail, vf2 (&dl)[4], vf& dl8, vf2 (&dlu)[4],
                               vf& dlu8) {
    vf2 uu[4], dd[4];
    vf u8, d8;
    scanh_chunk_read<T>(up, has_tail, uu, u8);
    scanh_chunk_read<T>(dp, has_tail, dd, d8);
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        const vf2 dr = dd[i] + spl2(splat(bias));
        dl[i] = softplus ? vsoftplus2(dr) : dr;
        dlu[i] = dl[i] * uu[i];
    }
    vf d = d8 + bias;
    if (softplus) d = vsoftplus(d);
    const vm tail_lane = (lane_id() == WAVE - 1) && (spl_i(has_tail ? 1 : 0) > 0);
    dl8 = vsel(tail_lane, d, splat(0.f));
    dlu8 = dl8 * u8;
}

template <class T, int TAIL, int MODE>
AUM_DEV void scanh_bwd_chunked(const AumScanBwdArgs& p, int wg, float* lds, int rows_per_wg) {
    using GE = ScanGeo<8, TAIL>;
    static_assert(MODE == 0 || MODE == 1, "one direction per launch");
    constexpr bool REV0 = MODE == 1;
    constexpr int NW = SCANH_CH_NW;
    constexpr int CS = 512;                                            // main steps per chunk
    const int N = p.dstate;
    float* Bt = lds;
    float* Ct = lds + GE::TILE;
    float* dBt = lds + 2 * GE::TILE;
    float* dCt = lds + 3 * GE::TILE;
    float* xcarry = lds + 4 * GE::TILE;                                // [rows][16]   pre-pass
    float* gcarry = xcarry + SCANH_CH_ROWS * SCANWG_MAX_N;             // [rows][16]   main pass
    const int nchunks = p.len / CS;
    const ScanWgWs L = scanwg_ws_layout(p.batch, p.dim, p.len, N, rows_per_wg, nchunks, false);
    float* ws = (float*)p.workspace;
    const int b = wg / L.gpb, g_idx = wg % L.gpb;
    const int eb = g_idx * rows_per_wg;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const bool accumulate = (p.flags & AUM_SCAN_ACCUMULATE) != 0;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);
    const int64_t xck_row_stride = (int64_t)nchunks * N;
    float* xck = ws + L.xck + (int64_t)wg * rows_per_wg * xck_row_stride;
    const int niter = (rows_per_wg + NW - 1) / NW;

    AUM_FOR_EACH_WAVE(w, NW) {
        for (int i0 = w * WAVE; i0 < 2 * SCANH_CH_ROWS * SCANWG_MAX_N; i0 += NW * WAVE) lds_write(xcarry, lane_id() + i0, splat(0.f));
    }
    AUM_WG_BARRIER();
    // ---- pre-pass: state entering every chunk, chunks in scan order (skipped when the forward's checkpoint is handed in) ----
    const float* ckpt = (TAIL > 0) ? p.x_ck : nullptr;
    for (int ci = 0; ci < (ckpt ? 0 : nchunks); ++ci) {
        const int c = REV0 ? nchunks - 1 - ci : ci;
        const int base = c * CS;
        const bool has_tail = TAIL > 0 && c == nchunks - 1;
        const int len_eff = has_tail ? p.len : base + CS;
        AUM_FOR_EACH_WAVE(w, NW) { scanwg_load_tile<T, 8, TAIL, NW, SCANH_PAIRED>(Bsrc, p.B_ns, N, base, len_eff, Bt, w); }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, NW) {
            const vi lane = lane_id();
            for (int it = 0; it < niter; ++it) {
                const int rloc = w + it * NW;
                const int e = eb + rloc;
                if (rloc >= rows_per_wg || e >= p.dim) continue;
                const float bias = p.delta_bias ? p.delta_bias[e] : 0.f;
                vf2 dl[4], dlu[4];
                vf dl8, dlu8;
                scanh_chunk_delta<T>(row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds) + base,
                                     row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)e * p.delta_ds) + base, bias, softplus, has_tail,
                                     dl, dl8, dlu, dlu8);
                for (int n = 0; n < N; ++n) {
                    const float An = p.A[(int64_t)e * N + n] * LOG2E;
                    vf2 a[4], bb[4], x[4];
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        a[i] = vexp2_2(dl[i] * spl2(splat(An)));
                        bb[i] = dlu[i] * mk2(lds_read(Bt, lane * GE::LK + scanh_slot_lo(i) + n * GE::SP), lds_read(Bt, lane * GE::LK + scanh_slot_hi(i) + n * GE::SP));
                    }
                    const vf a8 = vexp2(dl8 * An), bb8 = dlu8 * lds_read(Bt, spl_i(WAVE * GE::LK + n * GE::SP));
                    const vf cin = lds_read(xcarry, spl_i(rloc * SCANWG_MAX_N + n));
                    gstore_coherent(xck + rloc * xck_row_stride + (int64_t)c * N + n, spl_i(0), cin, lane == 0);
                    vf x8, xin, cout;
                    scanh_affine_c<REV0, true>(a, a8, bb, bb8, cin, x, x8, xin, cout);    // only the chunk total is used
                    lds_write(xcarry, spl_i(rloc * SCANWG_MAX_N + n), cout);
                }
            }
        }
        AUM_WG_BARRIER();
    }
    // ---- main pass: chunks in adjoint order ----
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = REV0 ? ci : nchunks - 1 - ci;
        const int base = c * CS;
        const bool has_tail = TAIL > 0 && c == nchunks - 1;
        const int len_eff = has_tail ? p.len : base + CS;
        const bool first_visit = ci == 0;
        AUM_FOR_EACH_WAVE(w, NW) {
            scanwg_load_tile<T, 8, TAIL, NW, SCANH_PAIRED>(Bsrc, p.B_ns, N, base, len_eff, Bt, w);
            scanwg_load_tile<T, 8, TAIL, NW, SCANH_PAIRED>(Csrc, p.C_ns, N, base, len_eff, Ct, w);
            for (int i0 = w * WAVE; i0 < GE::TILE; i0 += NW * WAVE) {
                const vi idx = lane_id() + i0;
                lds_write_m(dBt, idx, splat(0.f), idx < GE::TILE);
                lds_write_m(dCt, idx, splat(0.f), idx < GE::TILE);
            }
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, NW) {
            const vi lane = lane_id();
            const vm tail_lane = (lane == WAVE - 1) && (spl_i(has_tail ? 1 : 0) > 0);
            vi pos[4], pos4[4];
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) { pos[i] = lane * GE::LK + scanh_slot_lo(i); pos4[i] = lane * GE::LK + scanh_slot_hi(i); }
            const vi pos8 = spl_i(WAVE * GE::LK);
            for (int it = 0; it < niter; ++it) {
                const int rloc = w + it * NW;
                const int e = eb + rloc;
                const bool active = rloc < rows_per_wg && e < p.dim;   // wave-uniform; inactive waves still take every barrier below
                const int ec = active ? e : p.dim - 1;
                const int rl = active ? rloc : 0;
                const float bias = p.delta_bias ? p.delta_bias[ec] : 0.f;
                const T* up = row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)ec * p.u_ds) + base;
                const T* dp = row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)ec * p.delta_ds) + base;
                vf2 dl[4], dlu[4], dy[4], G[4], DA[4];
                vf dl8, dlu8, dy8, G8 = splat(0.f), DA8 = splat(0.f);
                scanh_chunk_delta<T>(up, dp, bias, softplus, has_tail, dl, dl8, dlu, dlu8);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) { G[i] = spl2(splat(0.f)); DA[i] = spl2(splat(0.f)); }
                {   // dout (and the gate): dy = dout * silu(z), dz = dout * out_pre * silu'(z)
                    vf2 go[4];
                    vf go8;
                    scanh_chunk_read<T>(row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)ec * p.dout_ds) + base, has_tail, go, go8);
                    if (p.z) {
                        vf2 zz[4], yp[4], dzv[4];
                        vf z8, yp8, dz8;
                        scanh_chunk_read<T>(row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)ec * p.z_ds) + base, has_tail, zz, z8);
                        scanh_chunk_read<T>(row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)ec * p.out_ds) + base, has_tail, yp, yp8);
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
                        if (active) {
                            T* dzp = row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)ec * p.dz_ds) + base;
                            if (accumulate) {
                                vf2 pv[4];
                                vf pv8;
                                scanh_chunk_read<T>(dzp, has_tail, pv, pv8);
                                AUM_UNROLL
                                for (int i = 0; i < 4; ++i) dzv[i] = dzv[i] + pv[i];
                                dz8 = dz8 + pv8;
                            }
                            scanh_chunk_write<T>(dzp, has_tail, dzv, dz8);
                        }
                    }
                    AUM_UNROLL
                    for (int i = 0; i < 4; ++i) dy[i] = go[i];
                    dy8 = vsel(tail_lane, go8, splat(0.f));
                }
                // forward direction: the adjoint enters from the chunk that follows in time and is multiplied by that chunk's
                // first step, a = exp(delta_next * A_n); nothing follows the last chunk (multiplier 1, carry 0)
                vf dnf = splat(0.f);
                if (!REV0 && base + CS + (has_tail ? 1 : 0) < p.len) {
                    dnf = gload_u(dp, spl_i(CS)) + bias;
                    if (softplus) dnf = vsoftplus(dnf);
                }
                vf dAv0 = splat(0.f);
                const bool want_dA = !(p.flags & AUM_DBG_SKIP_PARTIALS);
                for (int j = 0; j < SCANWG_MAX_N; ++j) {
                    const int n = (j + w) & (SCANWG_MAX_N - 1);
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
                        const float Araw = p.A[(int64_t)ec * N + n];
                        const vf xcin = ckpt ? gload_coherent(ckpt + (((int64_t)b * p.dim + ec) * nchunks + c) * N + n, spl_i(0), lane >= 0)
                                             : gload_coherent(xck + rl * xck_row_stride + (int64_t)c * N + n, spl_i(0), lane >= 0);
                        const vf gcin = lds_read(gcarry, spl_i(rl * SCANWG_MAX_N + n));
                        const vf a_edge = vexp2(dnf * (Araw * LOG2E));
                        vf gcout;
                        scanh_bwd_dir_state_c<REV0, true>(Araw, n, Bn, Bn8, Cn, Cn8, dl, dl8, dlu, dlu8, dy, dy8, G, G8, DA, DA8, dBacc, dB8,
                                                          dCacc, dC8, dAv0, want_dA, xcin, gcin, a_edge, gcout);
                        lds_write(gcarry, spl_i(rl * SCANWG_MAX_N + n), gcout);
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const vi a0 = pos[i] + n * GE::SP, a1 = pos4[i] + n * GE::SP;
                            if constexpr (SCANH_PAIRED) {      // adjacent words: one packed add per pair
                                const vf2 tb = mk2(lds_read(dBt, a0), lds_read(dBt, a1)) + dBacc[i];
                                const vf2 tc = mk2(lds_read(dCt, a0), lds_read(dCt, a1)) + dCacc[i];
                                lds_write(dBt, a0, lo2(tb));
                                lds_write(dBt, a1, hi2(tb));
                                lds_write(dCt, a0, lo2(tc));
                                lds_write(dCt, a1, hi2(tc));
                            } else {
                                lds_write(dBt, a0, lds_read(dBt, a0) + lo2(dBacc[i]));
                                lds_write(dBt, a1, lds_read(dBt, a1) + hi2(dBacc[i]));
                                lds_write(dCt, a0, lds_read(dCt, a0) + lo2(dCacc[i]));
                                lds_write(dCt, a1, lds_read(dCt, a1) + hi2(dCacc[i]));
                            }
                        }
                        if (TAIL > 0) {
                            const vi a8 = pos8 + n * GE::SP;          // shared slot: owned by the last lane
                            lds_write_m(dBt, a8, lds_read(dBt, a8) + dB8, lane == WAVE - 1);
                            lds_write_m(dCt, a8, lds_read(dCt, a8) + dC8, lane == WAVE - 1);
                        }
                    }
                    AUM_WG_BARRIER_IN_PHASE();
                }
                if (active && want_dA) {
                    const vm mn = lane < N;
                    const vi ln = vmin_i(lane, N - 1);
                    float* sA = ws + L.pA + ((int64_t)b * p.dim + e) * N;
                    vf v0 = sum_rows4(dAv0);
                    if (!first_visit) v0 = v0 + gload_coherent(sA, ln, mn);
                    gstore_coherent(sA, ln, v0, mn);
                }
                if (active) {
                    const float Dn = p.D ? p.D[e] : 0.f;
                    vf2 uu[4], raw[4], duv[4], ddv[4];
                    vf u8, raw8 = splat(0.f);
                    scanh_chunk_read<T>(up, has_tail, uu, u8);
                    if (softplus) scanh_chunk_read<T>(dp, has_tail, raw, raw8);
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
                    T* dup = row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds) + base;
                    T* ddp = row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds) + base;
                    vf du8s = du8, dd8s = dd8;      // stored values; dd8 itself still feeds the ddelta_bias partial below
                    if (accumulate) {       // second direction: lands on the first one's gradients
                        vf2 pv[4];
                        vf pv8;
                        scanh_chunk_read<T>(dup, has_tail, pv, pv8);
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) duv[i] = duv[i] + pv[i];
                        du8s = du8s + pv8;
                        scanh_chunk_read<T>(ddp, has_tail, pv, pv8);
                        AUM_UNROLL
                        for (int i = 0; i < 4; ++i) ddv[i] = ddv[i] + pv[i];
                        dd8s = dd8s + pv8;
                    }
                    scanh_chunk_write<T>(dup, has_tail, duv, du8s);
                    scanh_chunk_write<T>(ddp, has_tail, ddv, dd8s);
                    float sD = wave_sum(vfma(dy8, u8, lo2(dDl) + hi2(dDl)));
                    float sb = wave_sum(lo2(dbl) + hi2(dbl) + dd8);
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
        AUM_WG_BARRIER();
        // flush this chunk's dB/dC tile: one partial per workgroup, every (g,b,n,t) written exactly once
        AUM_FOR_EACH_WAVE(w, NW) {
            scanwg_store_tile<8, TAIL, NW, SCANH_PAIRED>(dBt, ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len, N, base, len_eff, w, p.len);
            scanwg_store_tile<8, TAIL, NW, SCANH_PAIRED>(dCt, ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len, N, base, len_eff, w, p.len);
        }
        AUM_WG_BARRIER();
    }
}

}  // namespace aum
