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

#ifndef AUM_SCANWG_NW
#define AUM_SCANWG_NW 8
#endif
constexpr int SCANWG_NW = AUM_SCANWG_NW;          // waves per workgroup
constexpr int SCANWG_MAX_N = 16;      // dstate limit of this path (LDS tile height)
constexpr int SCANWG_MAX_ROWS = 64;   // rows per workgroup (8 waves x 4 pairs x 2 rows)
constexpr int SCANWG_MAX_K = 9;       // <= 577 steps per chunk keeps the backward's 4 fp32 tiles inside 160 KB of LDS
// Ablation bits for timing experiments (upper half of `flags`, tools/kbench.py --only ablate): they exist only in -DAUM_ABLATE builds
// (tools/build_variant.sh).  In the production library the masks are zero, every `flags & AUM_DBG_SKIP_*` test is a constant and its
// branch is not in the kernels' hot loops.
#ifdef AUM_ABLATE
constexpr uint32_t AUM_DBG_SKIP_STATES = 1u << 16, AUM_DBG_SKIP_LDS_ATOMICS = 1u << 17, AUM_DBG_SKIP_PARTIALS = 1u << 18,
                   AUM_DBG_SKIP_EPILOGUE = 1u << 19, AUM_DBG_NO_STEP_BARRIER = 1u << 20;
#else
constexpr uint32_t AUM_DBG_SKIP_STATES = 0, AUM_DBG_SKIP_LDS_ATOMICS = 0, AUM_DBG_SKIP_PARTIALS = 0, AUM_DBG_SKIP_EPILOGUE = 0,
                   AUM_DBG_NO_STEP_BARRIER = 0;
#endif
// kernel selection for tests and A/B runs (host-side dispatch, not in any loop): 64-row workgroups in the chunked one-row backward
// whatever the grid
constexpr uint32_t AUM_DBG_CHUNKED_MAX_ROWS = 1u << 21;

template <int K, int TAIL> struct ScanGeo {
    static constexpr int KT = K + TAIL;                    // slots per lane
    static constexpr int LK = (K % 2 == 0) ? K + 1 : K;    // LDS words per lane in a tile row: odd => conflict-free
    static constexpr int S = WAVE * K + TAIL;              // time steps per chunk
    static constexpr int SP = WAVE * LK + TAIL;            // LDS words per tile row
    static constexpr int TILE = SCANWG_MAX_N * SP;
};
// geometries with a tail slot are single-chunk only: no carry area, which keeps <8,1> at two workgroups per CU
// CT ("chunked tail"): rows of 64K*m + 1 steps walked in 64K-step chunks, the tail slot owned by the last chunk; one direction
// per launch, so one carry area
template <int K, int TAIL, bool CT = false> constexpr int scanwg_fwd_lds_floats() {
    return 2 * ScanGeo<K, TAIL>::TILE + (CT ? SCANWG_MAX_ROWS * SCANWG_MAX_N : TAIL ? 0 : 2 * SCANWG_MAX_ROWS * SCANWG_MAX_N);
}
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

// Word of main step k (0..7) inside a lane's 8-step group of a tile row.  PAIRED (the half-packed one-row kernels, opt-in):
// steps i and 4+i sit next to each other, so the two halves of a vf2 are one ds_read2_b32 / ds_write2_b32 of adjacent words.
template <bool PAIRED> AUM_DEV constexpr int scan_tile_slot(int k) { return PAIRED ? ((k & 3) * 2 + (k >> 2)) : k; }

// Cooperative load of one [N][S] tile of B (or C) into LDS as fp32; t outside [0,len) -> 0.
// K == 8 with the main part inside the row: 16-byte loads of 8 consecutive steps (= one lane's main slots), so a tile is
// 2 load instructions per thread instead of 36 dependent 2-byte gathers; otherwise element-wise.
template <class T, int K, int TAIL, int NW = SCANWG_NW, bool PAIRED = false>
AUM_DEV void scanwg_load_tile(const T* src, int64_t n_stride, int N, int base, int len, float* tile, int w) {
    using G = ScanGeo<K, TAIL>;
    const vi lane = lane_id();
    if constexpr (K == 8) {
        if (base + WAVE * K <= len) {
            for (int i0 = w * WAVE; i0 < N * WAVE; i0 += NW * WAVE) {      // (state n, lane-block j) pairs
                const vi idx = lane + i0;
                const vm in = idx < N * WAVE;
                const vi n = vmin_i(idx >> 6, N - 1);
                const vi j = idx & (WAVE - 1);
                vf v[8];
                gload8(src, n * (int)n_stride + j * 8 + base, in, v);
                AUM_UNROLL
                for (int k = 0; k < 8; ++k) lds_write_m(tile, n * G::SP + j * G::LK + scan_tile_slot<PAIRED>(k), v[k], in);
            }
            if (TAIL > 0) {       // tail columns: N x TAIL scalars
                const vi n = vmin_i(lane / (TAIL > 0 ? TAIL : 1), N - 1);
                const vi jt = lane - (lane / (TAIL > 0 ? TAIL : 1)) * (TAIL > 0 ? TAIL : 1);
                const vm in = (lane < N * TAIL) && (spl_i(w) == 0);
                const vi t = jt + (base + WAVE * K);
                const vf v = vsel(t < len, gload_u(src, n * (int)n_stride + vmin_i(t, len - 1)), splat(0.f));
                lds_write_m(tile, n * G::SP + WAVE * G::LK + jt, v, in);
            }
            return;
        }
    }
    static_assert(!PAIRED || K == 8, "paired tiles: K == 8 only");
    for (int i0 = w * WAVE; i0 < N * G::S; i0 += NW * WAVE) {     // (PAIRED callers never get here: their rows hold >= 512 steps)
        const vi idx = lane + i0;
        const vm in = idx < N * G::S;
        const vi n = vmin_i(idx / G::S, N - 1);
        const vi tt = vmin_i(idx - (idx / G::S) * G::S, G::S - 1);
        const vi t = tt + base;
        const vf v = vsel(t < len, gload_u(src, n * (int)n_stride + vmin_i(t, len - 1)), splat(0.f));
        lds_write_m(tile, n * G::SP + scan_tile_pos<K, TAIL>(tt), v, in);
    }
}

// Write one [N][S] fp32 LDS tile to a dense (N, len) fp32 global partial (t >= len skipped); vectorised like the load.
// `pitch` (default: len) is the row pitch of dst when `len` is only the validity bound (chunked rows whose tail column
// belongs to the last chunk).
template <int K, int TAIL, int NW = SCANWG_NW, bool PAIRED = false>
AUM_DEV void scanwg_store_tile(const float* tile, float* dst, int N, int base, int len, int w, int pitch = -1) {
    using G = ScanGeo<K, TAIL>;
    const vi lane = lane_id();
    if (pitch < 0) pitch = len;
    if constexpr (K == 8) {
        if (base + WAVE * K <= len) {
            for (int i0 = w * WAVE; i0 < N * WAVE; i0 += NW * WAVE) {
                const vi idx = lane + i0;
                const vm in = idx < N * WAVE;
                const vi n = vmin_i(idx >> 6, N - 1);
                const vi j = idx & (WAVE - 1);
                vf v[8];
                AUM_UNROLL
                for (int k = 0; k < 8; ++k) v[k] = lds_read(tile, n * G::SP + j * G::LK + scan_tile_slot<PAIRED>(k));
                gstore8(dst, n * pitch + j * 8 + base, v, in);
            }
            if (TAIL > 0) {
                const vi n = vmin_i(lane / (TAIL > 0 ? TAIL : 1), N - 1);
                const vi jt = lane - (lane / (TAIL > 0 ? TAIL : 1)) * (TAIL > 0 ? TAIL : 1);
                const vi t = jt + (base + WAVE * K);
                const vm in = (lane < N * TAIL) && (spl_i(w) == 0) && (t < len);
                gstore(dst, n * pitch + t, lds_read(tile, n * G::SP + WAVE * G::LK + jt), in);
            }
            return;
        }
    }
    for (int i0 = w * WAVE; i0 < N * G::S; i0 += NW * WAVE) {
        const vi idx = lane + i0;
        const vi n = vmin_i(idx / G::S, N - 1);
        const vi tt = vmin_i(idx - (idx / G::S) * G::S, G::S - 1);
        const vi tg = tt + base;
        const vm m = (idx < N * G::S) && (tg < len);
        gstore(dst, n * pitch + tg, lds_read(tile, n * G::SP + scan_tile_pos<K, TAIL>(tt)), m);
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

// Two-row affine scan on packed pairs: x'[k] = a[k]*x + b[k] for both rows of the wave at once (see
// affine_scan_states_f in scan_kernels.h for the single-row form and the meaning of the arguments).
template <int K, bool REV, class AF>
AUM_DEV void affine_scan_states2_f(AF a, const vf2 (&b)[K], vf2 Ptot, vf2 carry_in, vf2 (&x)[K], vf2& x_in, vf2& carry_out) {
    vf2 s = spl2(splat(0.f));
    AUM_UNROLL
    for (int kk = 0; kk < K; ++kk) {
        const int k = REV ? K - 1 - kk : kk;
        s = vfma2(a(k), s, b[k]);
    }
    const vm first = lane_id() == (REV ? WAVE - 1 : 0);
    vf2 S = vsel2(first, vfma2(Ptot, carry_in, s), s);
    vf2 P = Ptot;
    wave_scan_affine2<REV>(P, S);
    x_in = REV ? mk2(dpp_wave_shl1(lo2(S), lo2(carry_in)), dpp_wave_shl1(hi2(S), hi2(carry_in)))
               : mk2(dpp_wave_shr1(lo2(S), lo2(carry_in)), dpp_wave_shr1(hi2(S), hi2(carry_in)));
    carry_out = mk2(splat(readlane(lo2(S), REV ? 0 : WAVE - 1)), splat(readlane(hi2(S), REV ? 0 : WAVE - 1)));
    vf2 xx = x_in;
    AUM_UNROLL
    for (int kk = 0; kk < K; ++kk) {
        const int k = REV ? K - 1 - kk : kk;
        xx = vfma2(a(k), xx, b[k]);
        x[k] = xx;
    }
}

// delta = softplus(delta + bias) (masked to 0 outside the row), delta*u, and the lane sum of delta, for the pair's
// two rows, packed (row e0 in .x, row e0+1 in .y)
template <class T, int K, int TAIL>
AUM_DEV void scanwg_load_rows(const void* u, int64_t u_bs, int64_t u_ds, const void* delta, int64_t d_bs, int64_t d_ds,
                              const float* delta_bias, bool softplus, int b, int e0, int dim, int base, int len,
                              const vi (&t)[K + TAIL], const vm (&valid)[K + TAIL], vf2 (&dl)[K + TAIL],
                              vf2 (&dlu)[K + TAIL], vf2& sumd) {
    vf uus[SCAN_R][K + TAIL], dds[SCAN_R][K + TAIL];
    float biases[SCAN_R];
    bool rowoks[SCAN_R];
    AUM_UNROLL
    for (int r = 0; r < SCAN_R; ++r) {
        const int e = e0 + r;
        rowoks[r] = e < dim;
        const int ec = rowoks[r] ? e : dim - 1;
        const T* up = row_ptr<T>(u, (int64_t)b * u_bs + (int64_t)ec * u_ds);
        const T* dp = row_ptr<T>(delta, (int64_t)b * d_bs + (int64_t)ec * d_ds);
        biases[r] = delta_bias ? delta_bias[ec] : 0.f;
        scan_row_read<T, K, TAIL>(up, base, len, t, valid, uus[r]);
        scan_row_read<T, K, TAIL>(dp, base, len, t, valid, dds[r]);
    }
    // the two rows are packed BEFORE the softplus so that its adds / multiplies / residual step run as v_pk_*
    const vf2 bias2 = mk2(splat(biases[0]), splat(biases[1]));
    sumd = spl2(splat(0.f));
    AUM_UNROLL
    for (int k = 0; k < K + TAIL; ++k) {
        vf2 d = mk2(dds[0][k], dds[1][k]) + bias2;
        if (softplus) d = vsoftplus2(d);
        d = mk2(vsel(valid[k] && rowoks[0], lo2(d), splat(0.f)), vsel(valid[k] && rowoks[1], hi2(d), splat(0.f)));
        dl[k] = d;
        dlu[k] = d * mk2(uus[0][k], uus[1][k]);
        sumd = sumd + d;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_fwd_dir(const AumScanFwdArgs& p, int b, int e0, int rloc, const float* Aptr, const float* Bt,
                            const float* Ct, const vi (&pos)[K + TAIL], const vf2 (&dl)[K + TAIL],
                            const vf2 (&dlu)[K + TAIL], vf2 sumd, float* carry, bool multi, bool write_last,
                            vf2 (&y)[K + TAIL]) {
    using G = ScanGeo<K, TAIL>;
    constexpr int KT = G::KT;
    const int N = p.dstate;
    const vi lane = lane_id();
    const int e1 = e0 + 1 < p.dim ? e0 + 1 : e0;           // odd dim: the second row mirrors the first, outputs masked
    for (int n = 0; n < N; ++n) {
        vf2 a[KT], bb[KT], x[KT];
        const vf2 An = mk2(splat(Aptr[(int64_t)e0 * N + n] * LOG2E), splat(Aptr[(int64_t)e1 * N + n] * LOG2E));
        AUM_UNROLL
        for (int k = 0; k < KT; ++k) {
            a[k] = vexp2_2(dl[k] * An);
            bb[k] = dlu[k] * spl2(lds_read(Bt, pos[k] + n * G::SP));
        }
        const vf2 Ptot = vexp2_2(sumd * An);
        vf2 cin = spl2(splat(0.f));
        if (multi) cin = mk2(lds_read(carry, spl_i(rloc * SCANWG_MAX_N + n)), lds_read(carry, spl_i((rloc + 1) * SCANWG_MAX_N + n)));
        vf2 xin, cout;
        affine_scan_states2_f<KT, REV>([&](int k) -> const vf2& { return a[k]; }, bb, Ptot, cin, x, xin, cout);
        AUM_UNROLL
        for (int k = 0; k < KT; ++k) y[k] = vfma2(spl2(lds_read(Ct, pos[k] + n * G::SP)), x[k], y[k]);
        if (multi) {
            lds_write(carry, spl_i(rloc * SCANWG_MAX_N + n), lo2(cout));
            lds_write(carry, spl_i((rloc + 1) * SCANWG_MAX_N + n), hi2(cout));
        }
        if (write_last) {
            gstore(p.last_state + ((int64_t)b * p.dim + e0) * N + n, spl_i(0), lo2(cout), lane == 0);
            if (e0 + 1 < p.dim) gstore(p.last_state + ((int64_t)b * p.dim + e0 + 1) * N + n, spl_i(0), hi2(cout), lane == 0);
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
                vf2 dl[KT], dlu[KT], y[KT], sumd;
                scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                             e0, p.dim, base, p.len, t, valid, dl, dlu, sumd);
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) y[k] = spl2(splat(0.f));
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
                            for (int k = 0; k < KT; ++k) o[k] = vfma(uu[k], splat(Dn), r == 0 ? lo2(y[k]) : hi2(y[k]));
                        } else {
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = r == 0 ? lo2(y[k]) : hi2(y[k]);
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

// scanwg_fwd for rows of 64K*m + 1 steps walked in 64K-step chunks (CT, "chunked tail"): the tail slot is owned by the last
// chunk and hidden from the others through `len_eff`.  Kept as a separate function so that the single-chunk kernels above
// compile exactly as before.
template <class T, int K, int TAIL, int MODE, bool CT = true>
AUM_DEV void scanwg_fwd_ct(const AumScanFwdArgs& p, int wg, float* lds, int rows_per_wg) {
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
    static_assert(!CT || (TAIL == 1 && MODE != 2), "chunked-tail rows: tail geometry, one direction");
    constexpr int CSTEP = CT ? WAVE * K : G::S;                       // time steps between chunk starts
    const int nchunks = CT ? p.len / CSTEP : (p.len + G::S - 1) / G::S;
    const bool multi = nchunks > 1;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const T* Bsrc = row_ptr<T>(p.B, (int64_t)b * p.B_bs);
    const T* Csrc = row_ptr<T>(p.C, (int64_t)b * p.C_bs);

    if (multi) {
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            for (int i0 = w * WAVE; i0 < (CT ? 1 : 2) * SCANWG_MAX_ROWS * SCANWG_MAX_N; i0 += SCANWG_NW * WAVE)
                lds_write(carry, lane_id() + i0, splat(0.f));
        }
        AUM_WG_BARRIER();
    }
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = (MODE == 1) ? nchunks - 1 - ci : ci;
        const int base = c * CSTEP;
        const int len_eff = (CT && c != nchunks - 1) ? base + CSTEP : p.len;      // hides the tail slot from all but the last chunk
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            scanwg_load_tile<T, K, TAIL>(Bsrc, p.B_ns, N, base, len_eff, Bt, w);
            scanwg_load_tile<T, K, TAIL>(Csrc, p.C_ns, N, base, len_eff, Ct, w);
        }
        AUM_WG_BARRIER();
        AUM_FOR_EACH_WAVE(w, SCANWG_NW) {
            for (int pair = w; 2 * pair < rows_per_wg; pair += SCANWG_NW) {
                const int rloc = 2 * pair;
                const int e0 = eb + rloc;
                if (e0 >= p.dim) break;
                vi t[KT], pos[KT];
                vm valid[KT];
                scan_slots<K, TAIL>(base, len_eff, t, valid, pos);
                vf2 dl[KT], dlu[KT], y[KT], sumd;
                scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                             e0, p.dim, base, len_eff, t, valid, dl, dlu, sumd);
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) y[k] = spl2(splat(0.f));
                if (CT && p.x_ck) {      // checkpoint: the state entering this chunk, lanes 0..N-1 <- states of the pair's two rows
                    const vi ln = lane_id() & (SCANWG_MAX_N - 1);
                    const vm mn = (lane_id() < SCANWG_MAX_N) && (ln < N);
                    AUM_UNROLL
                    for (int r = 0; r < SCAN_R; ++r)
                        if (e0 + r < p.dim)
                            gstore(p.x_ck + (((int64_t)b * p.dim + e0 + r) * nchunks + c) * N, vmin_i(ln, N - 1),
                                   lds_read(carry, ln + (rloc + r) * SCANWG_MAX_N), mn);
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
                            scan_row_read<T, K, TAIL>(row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)e * p.u_ds), base, len_eff, t,
                                                      valid, uu);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = vfma(uu[k], splat(Dn), r == 0 ? lo2(y[k]) : hi2(y[k]));
                        } else {
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = r == 0 ? lo2(y[k]) : hi2(y[k]);
                        }
                        if (p.out_pre) scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.out_pre, ooff), base, len_eff, t, valid, o);
                        if (p.z) {
                            vf zz[KT];
                            scan_row_read<T, K, TAIL>(row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)e * p.z_ds), base, len_eff, t,
                                                      valid, zz);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = o[k] * (zz[k] * vsigmoid(zz[k]));
                        }
                        if (p.flags & AUM_SCAN_ACCUMULATE) {     // second direction: lands on the first one's output
                            vf prev[KT];
                            scan_row_read<T, K, TAIL>(row_ptr<T>(p.out, ooff), base, len_eff, t, valid, prev);
                            AUM_UNROLL
                            for (int k = 0; k < KT; ++k) o[k] = o[k] + prev[k];
                        }
                        scan_row_write<T, K, TAIL>(row_ptr_w<T>(p.out, ooff), base, len_eff, t, valid, o);
                    }
                }
            }
        }
        AUM_WG_BARRIER();
    }
}

// ------------------------------------------------------------------------------------------------
// Backward (two rows per wave, packed)
// ------------------------------------------------------------------------------------------------
template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_bwd_dir_state(const AumScanBwdArgs& p, int n, int e0, int rloc, const float* Aptr,
                                  const vf (&Bn)[K + TAIL], const vf (&Cn)[K + TAIL], const vf2 (&dl)[K + TAIL],
                                  const vf2 (&dlu)[K + TAIL], const vf2 (&dy)[K + TAIL], vf2 sumd, vf2 sumd_next, vf2 dnf,
                                  const float* xck, int64_t xck_row_stride, int chunk, float* gcarry, bool multi,
                                  vf2 (&G)[K + TAIL], vf2 (&DA)[K + TAIL], vf2 (&dBacc)[K + TAIL], vf2 (&dCacc)[K + TAIL],
                                  vf2& dAv) {
    constexpr int KT = K + TAIL;
    const int N = p.dstate;
    const int e1 = e0 + 1 < p.dim ? e0 + 1 : e0;
    const vf2 Araw = mk2(splat(Aptr[(int64_t)e0 * N + n]), splat(Aptr[(int64_t)e1 * N + n]));
    const vf2 An = Araw * spl2(splat(LOG2E));
    vf2 a[KT], bb[KT], x[KT];
    AUM_UNROLL
    for (int k = 0; k < KT; ++k) {
        a[k] = vexp2_2(dl[k] * An);
        bb[k] = dlu[k] * spl2(Bn[k]);
    }
    const vf2 Ptot = vexp2_2(sumd * An);
    vf2 cin = spl2(splat(0.f));
    if (multi)
        cin = mk2(gload_coherent(xck + rloc * xck_row_stride + (int64_t)chunk * N + n, spl_i(0), lane_id() >= 0),
                  gload_coherent(xck + (rloc + 1) * xck_row_stride + (int64_t)chunk * N + n, spl_i(0), lane_id() >= 0));
    vf2 xin, cout;
    affine_scan_states2_f<KT, REV>([&](int k) -> const vf2& { return a[k]; }, bb, Ptot, cin, x, xin, cout);
    // adjoint: g_s = dy_s*C_s + a_{s+1} * g_{s+1}, scanned against the recurrence direction; the multiplier of slot k
    // is a of the slot's scan successor (neighbour lane at the lane edge, next chunk's first step at the chunk edge)
    vf2 cc[KT], g[KT];
    const vf2 a_nf = vexp2_2(dnf * An);
    AUM_UNROLL
    for (int k = 0; k < KT; ++k) cc[k] = dy[k] * spl2(Cn[k]);
    const vf2 a_edge = REV ? mk2(dpp_wave_shr1(lo2(a[KT - 1]), lo2(a_nf)), dpp_wave_shr1(hi2(a[KT - 1]), hi2(a_nf)))
                           : mk2(dpp_wave_shl1(lo2(a[0]), lo2(a_nf)), dpp_wave_shl1(hi2(a[0]), hi2(a_nf)));
    auto a_succ = [&](int k) -> const vf2& {
        if (!REV) return (k + 1 < KT) ? a[k + 1 < KT ? k + 1 : 0] : a_edge;
        return (k > 0) ? a[k > 0 ? k - 1 : 0] : a_edge;
    };
    const vf2 Pn = vexp2_2(sumd_next * An);
    vf2 gin_c = spl2(splat(0.f));
    if (multi) gin_c = mk2(lds_read(gcarry, spl_i(rloc * SCANWG_MAX_N + n)), lds_read(gcarry, spl_i((rloc + 1) * SCANWG_MAX_N + n)));
    vf2 gin, gout;
    affine_scan_states2_f<KT, !REV>(a_succ, cc, Pn, gin_c, g, gin, gout);
    if (multi) {
        lds_write(gcarry, spl_i(rloc * SCANWG_MAX_N + n), lo2(gout));
        lds_write(gcarry, spl_i((rloc + 1) * SCANWG_MAX_N + n), hi2(gout));
    }
    vf2 dAl = spl2(splat(0.f));
    AUM_UNROLL
    for (int k = 0; k < KT; ++k) {
        const vf2 xprev = REV ? ((k == KT - 1) ? xin : x[k + 1 < KT ? k + 1 : 0]) : ((k == 0) ? xin : x[k > 0 ? k - 1 : 0]);
        const vf2 h = g[k] * a[k] * xprev;
        G[k] = vfma2(g[k], spl2(Bn[k]), G[k]);
        DA[k] = vfma2(Araw, h, DA[k]);
        dBacc[k] = vfma2(g[k], dlu[k], dBacc[k]);
        dCacc[k] = vfma2(dy[k], x[k], dCacc[k]);
        dAl = vfma2(dl[k], h, dAl);
    }
    // dA[e][n] partials of the two rows: parked in lane n, written once per row after the state loop
    if (!(p.flags & AUM_DBG_SKIP_PARTIALS))
        dAv = vsel2(lane_id() == n, mk2(splat(wave_sum(lo2(dAl))), splat(wave_sum(hi2(dAl)))), dAv);
}

template <class T, int K, int TAIL, bool REV>
AUM_DEV void scanwg_bwd_prepass_pair(const AumScanBwdArgs& p, int b, int e0, int rloc, int c, int base, const float* Aptr,
                                     const float* Bt, float* xck, int64_t xck_row_stride, float* xcarry) {
    using G = ScanGeo<K, TAIL>;
    constexpr int KT = G::KT;
    const int N = p.dstate;
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const vi lane = lane_id();
    const int e1 = e0 + 1 < p.dim ? e0 + 1 : e0;
    vi t[KT], pos[KT];
    vm valid[KT];
    scan_slots<K, TAIL>(base, p.len, t, valid, pos);
    vf2 dl[KT], dlu[KT], sumd;
    scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b, e0, p.dim,
                                 base, p.len, t, valid, dl, dlu, sumd);
    for (int n = 0; n < N; ++n) {
        const vf2 An = mk2(splat(Aptr[(int64_t)e0 * N + n] * LOG2E), splat(Aptr[(int64_t)e1 * N + n] * LOG2E));
        vf2 a[KT], bb[KT], x[KT];
        AUM_UNROLL
        for (int k = 0; k < KT; ++k) {
            a[k] = vexp2_2(dl[k] * An);
            bb[k] = dlu[k] * spl2(lds_read(Bt, pos[k] + n * G::SP));
        }
        const vf2 cin = mk2(lds_read(xcarry, spl_i(rloc * SCANWG_MAX_N + n)), lds_read(xcarry, spl_i((rloc + 1) * SCANWG_MAX_N + n)));
        gstore_coherent(xck + rloc * xck_row_stride + (int64_t)c * N + n, spl_i(0), lo2(cin), lane == 0);
        gstore_coherent(xck + (rloc + 1) * xck_row_stride + (int64_t)c * N + n, spl_i(0), hi2(cin), lane == 0);
        vf2 xin, cout;
        affine_scan_states2_f<KT, REV>([&](int k) -> const vf2& { return a[k]; }, bb, vexp2_2(sumd * An), cin, x, xin, cout);
        lds_write(xcarry, spl_i(rloc * SCANWG_MAX_N + n), lo2(cout));
        lds_write(xcarry, spl_i((rloc + 1) * SCANWG_MAX_N + n), hi2(cout));
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
                const int e0c = active ? e0 : p.dim - 1;
                vi t[KT], pos[KT];
                vm valid[KT];
                scan_slots<K, TAIL>(base, p.len, t, valid, pos);
                vf2 dl[KT], dlu[KT], dy[KT], G[KT], DA[KT], sumd;
                scanwg_load_rows<T, K, TAIL>(p.u, p.u_bs, p.u_ds, p.delta, p.delta_bs, p.delta_ds, p.delta_bias, softplus, b,
                                             e0, p.dim, base, p.len, t, valid, dl, dlu, sumd);   // rows >= dim give zeros
                vf dys[SCAN_R][KT], dfs[SCAN_R], drs[SCAN_R];
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
                    for (int k = 0; k < KT; ++k) dys[r][k] = rowok ? go[k] : splat(0.f);
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
                    dfs[r] = df;
                    drs[r] = dr;
                }
                AUM_UNROLL
                for (int k = 0; k < KT; ++k) {
                    dy[k] = mk2(dys[0][k], dys[1][k]);
                    G[k] = spl2(splat(0.f));
                    DA[k] = spl2(splat(0.f));
                }
                const vf2 dnf_f = mk2(dfs[0], dfs[1]), dnf_r = mk2(drs[0], drs[1]);
                const vf2 sumd_nf = sumd - dl[0] + mk2(dpp_wave_shl1(lo2(dl[0]), dfs[0]), dpp_wave_shl1(hi2(dl[0]), dfs[1]));
                const vf2 sumd_nr = sumd - dl[KT - 1] +
                                    mk2(dpp_wave_shr1(lo2(dl[KT - 1]), drs[0]), dpp_wave_shr1(hi2(dl[KT - 1]), drs[1]));
                const bool first_visit = ci == 0;
                vf2 dAv0 = spl2(splat(0.f)), dAv1 = spl2(splat(0.f));      // lane n <- dA (dA_b) partial of state n
                // Rotated state order: at step j wave w works on state (j + 2w) mod 16, so no two of the 8 waves ever
                // hold the same state row of the dB/dC tiles within a step, nor in adjacent steps (j+1+2w = j+2w' has no
                // solution); a barrier after every SECOND step therefore keeps any two waves at most one step apart and
                // the tile update below can be a plain LDS read-add-write.
                for (int j = 0; j < SCANWG_MAX_N; ++j) {
                    const int n = (j + (SCANWG_MAX_N / SCANWG_NW) * w) & (SCANWG_MAX_N - 1);
                    if (active && n < N) {
                        vf Bn[KT], Cn[KT];
                        vf2 dBacc[KT], dCacc[KT];
                        AUM_UNROLL
                        for (int k = 0; k < KT; ++k) {
                            Bn[k] = lds_read(Bt, pos[k] + n * GE::SP);
                            Cn[k] = lds_read(Ct, pos[k] + n * GE::SP);
                            dBacc[k] = spl2(splat(0.f));
                            dCacc[k] = spl2(splat(0.f));
                        }
                        if (!(p.flags & AUM_DBG_SKIP_STATES)) {
                            if (MODE == 0 || BI)
                                scanwg_bwd_dir_state<T, K, TAIL, false>(p, n, e0c, rloc, p.A, Bn, Cn, dl, dlu, dy, sumd, sumd_nf,
                                                                        dnf_f, xck, xck_row_stride, c, gcarry, multi, G, DA,
                                                                        dBacc, dCacc, dAv0);
                            if (MODE == 1)
                                scanwg_bwd_dir_state<T, K, TAIL, true>(p, n, e0c, rloc, p.A, Bn, Cn, dl, dlu, dy, sumd, sumd_nr,
                                                                       dnf_r, xck, xck_row_stride, c, gcarry, multi, G, DA,
                                                                       dBacc, dCacc, dAv0);
                            if (BI)
                                scanwg_bwd_dir_state<T, K, TAIL, true>(p, n, e0c, rloc, p.A_b, Bn, Cn, dl, dlu, dy, sumd, sumd_nr,
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
                                lds_write_m(dBt, at, lds_read(dBt, at) + (lo2(dBacc[k]) + hi2(dBacc[k])), own);
                                lds_write_m(dCt, at, lds_read(dCt, at) + (lo2(dCacc[k]) + hi2(dCacc[k])), own);
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
                            vf v0 = r == 0 ? lo2(dAv0) : hi2(dAv0);
                            const vf v1 = r == 0 ? lo2(dAv1) : hi2(dAv1);
                            if (!first_visit) {     // later chunks of a multi-chunk row accumulate (same wave wrote it)
                                v0 = v0 + gload_coherent(sA, ln, mn);
                                gstore_coherent(sA, ln, v0, mn);
                            } else if (multi) {
                                gstore_coherent(sA, ln, v0, mn);
                            } else {
                                gstore(sA, ln, v0, mn);
                                if (BI) gstore(sAb, ln, v1, mn);
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
                                const vf dlk = r == 0 ? lo2(dl[k]) : hi2(dl[k]);
                                const vf Gk = r == 0 ? lo2(G[k]) : hi2(G[k]);
                                const vf DAk = r == 0 ? lo2(DA[k]) : hi2(DA[k]);
                                const vf dyk = r == 0 ? lo2(dy[k]) : hi2(dy[k]);
                                duv[k] = vfma(dlk, Gk, dyk * Dn);
                                vf dd = vfma(uu[k], Gk, DAk);
                                if (softplus) {
                                    const vf rw = raw[k] + bias;
                                    dd = vsel(rw > 20.f, dd, dd * vsigmoid(rw));
                                }
                                dd = vsel(valid[k], dd, splat(0.f));
                                ddv[k] = dd;
                                dDl = vfma(dyk, uu[k], dDl);
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
            scanwg_store_tile<K, TAIL>(dBt, ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len, N, base, p.len, w);
            scanwg_store_tile<K, TAIL>(dCt, ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len, N, base, p.len, w);
        }
        AUM_WG_BARRIER();
    }
}

// Second stage of the backward: dst[i] += sum_o src[o*inner + i] for up to 6 segments.
struct ScanReduceSeg { float* dst; const float* src; int64_t inner; int32_t outer; int32_t pad; };
struct ScanReduceArgs { ScanReduceSeg seg[6]; int32_t nseg; };

}  // namespace aum
