// conv_tm_kernels.h -- width <= 4 depthwise causal conv1d (+ SiLU) on TOKEN-MAJOR activations (batch, len, dim): the channel is the
// fastest axis, so a lane owns 16 (or, in the backward, 8) bytes of consecutive channels, a wave 1 KB (512 bytes) of a token row, and the causal
// window is four REGISTER rows that slide along time -- no halo exchange between lanes, every global access 16 (8) bytes per lane.
// x may be the first half of the in_proj output (row stride 2 * dim), dx the first half of its gradient.
//   forward   (MS:272 / causal_conv1d_fn):  y[t] = silu(bias + sum_k w[k] x[t - (W-1) + k])
//   backward  (SSI:594-596 call site):      dpre = dy silu'(pre);  dx[t] = sum_k w[k] dpre[t + (W-1) - k];
//                                           dw[k] = sum_{b,t} dpre[t] x[t - (W-1) + k];  dbias = sum_{b,t} dpre[t]
// AUM_CONV_REVERSE: the same on the time-reversed sequence (flip(conv(flip(x))) without the copies) -- the wave walks time the other way.
// A wave takes convt_tc(len) steps of one batch entry; the backward walks them from the last to the first (dx[t] needs dpre[t .. t+3], known
// by then) and leaves ONE partial row of dw / dbias per wave: [part][dim][k] and [part][dim] fp32, summed by aum_sum_rows in a fixed
// order (no atomics: bitwise repeatable).
#pragma once
#include "wave.h"

namespace aum {

constexpr int CONVT_W = 4;
// Time steps per wave (at most) and bytes of a token row per lane, separately for the two kernels (round 5, profiles/r05_ab_conv_tc.txt): the
// forward keeps 16 bytes per lane and 64-step chunks (L = 513: nine ranges of 57 steps; 3 x 64 x 9 = 1 728 waves at 140 registers, all resident);
// the backward holds two tensors and a twice-as-wide window -- 251 registers at 16 bytes per lane, 1.69 waves per SIMD of which no SIMD can
// take a third -- and runs with 8 bytes per lane (four channels of a 16-bit type, 136 registers) and 65-step chunks: 6 x 64 x 8 = 3 072 waves,
// exactly three per SIMD (2.16 -> 1.92 ms per step of the bench, same box; the forward in that form 1.05 -> 1.08: not taken).
#ifndef AUM_CONVT_TC_FWD
#define AUM_CONVT_TC_FWD 64
#endif
#ifndef AUM_CONVT_TC_BWD
#define AUM_CONVT_TC_BWD 65
#endif
#ifndef AUM_CONVT_NB16_FWD
#define AUM_CONVT_NB16_FWD 16
#endif
#ifndef AUM_CONVT_BWD_BLOCKS
#define AUM_CONVT_BWD_BLOCKS 1     // 8-step blocks of loads the backward keeps in flight (2: opt-in A/B build)
#endif
#ifndef AUM_CONVT_NB16_BWD
#define AUM_CONVT_NB16_BWD 8
#endif
static_assert((AUM_CONVT_NB16_FWD == 16 || AUM_CONVT_NB16_FWD == 8) && (AUM_CONVT_NB16_BWD == 16 || AUM_CONVT_NB16_BWD == 8), "16 or 8 bytes per lane");
constexpr int CONVT_UB = 8;        // steps fetched together (raw fragments, widened when used); the forward keeps two such blocks in flight
template <bool BWD> AUM_HOSTDEV constexpr int convt_tcmax() { return BWD ? AUM_CONVT_TC_BWD : AUM_CONVT_TC_FWD; }

template <bool BWD> AUM_HOSTDEV int convt_chunks(int len) { return (len + convt_tcmax<BWD>() - 1) / convt_tcmax<BWD>(); }
// steps per wave: the row cut into convt_chunks(len) EQUAL ranges -- L = 513 is nine ranges of 57 steps, not eight of 64 and a ninth wave
// with one step (same box: forward 1.17 -> 1.12 ms, backward 2.12 -> 2.02 ms per step of the bench)
template <bool BWD> AUM_HOSTDEV int convt_tc(int len) {
    const int nch = convt_chunks<BWD>(len), t = (len + nch - 1) / nch;
    return (nch - 1) * t < len ? t : convt_tcmax<BWD>();
}
template <class T, bool BWD> AUM_HOSTDEV constexpr int convt_nb() { return sizeof(T) == 4 ? 16 : (BWD ? AUM_CONVT_NB16_BWD : AUM_CONVT_NB16_FWD); }
template <class T, bool BWD> AUM_HOSTDEV constexpr int convt_vec() { return convt_nb<T, BWD>() / (int)sizeof(T); }
template <int NB> struct ConvtRawSel { typedef vq type; };
template <> struct ConvtRawSel<8> { typedef vh type; };
template <class T, bool BWD> using convt_raw = typename ConvtRawSel<convt_nb<T, BWD>()>::type;
template <class T, bool BWD> AUM_DEV convt_raw<T, BWD> convt_load(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    if constexpr (convt_nb<T, BWD>() == 16) return gbuf_load16(b, voff_bytes, soff_bytes);
    else return gbuf_load8(b, voff_bytes, soff_bytes);
}
template <class T, bool BWD> AUM_DEV void convt_unpack(const convt_raw<T, BWD>& q, vf (&o)[convt_vec<T, BWD>()]) {
    if constexpr (convt_nb<T, BWD>() == 16) vq_unpack<T>(q, o);
    else vh_unpack<T>(q, o);
}
template <class T, bool BWD> AUM_DEV void convt_store_m(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vf (&v)[convt_vec<T, BWD>()], vm m) {
    if constexpr (convt_nb<T, BWD>() == 16) gbuf_store16_m(b, voff_bytes, soff_bytes, vq_pack<T>(v), m);
    else gbuf_store8_m(b, voff_bytes, soff_bytes, vh_pack<T>(v), m);
}
template <class T, bool BWD> AUM_DEV void convt_store(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vf (&v)[convt_vec<T, BWD>()]) {
    if constexpr (convt_nb<T, BWD>() == 16) gbuf_store16(b, voff_bytes, soff_bytes, vq_pack<T>(v));
    else gbuf_store8(b, voff_bytes, soff_bytes, vh_pack<T>(v));
}
template <bool V> struct ConvtTag { static constexpr bool value = V; };
template <class T, bool BWD> AUM_HOSTDEV int convt_cblocks(int dim) { return (dim + WAVE * convt_vec<T, BWD>() - 1) / (WAVE * convt_vec<T, BWD>()); }
AUM_HOSTDEV inline int convt_nparts(int batch, int len) { return batch * convt_chunks<true>(len); }      // the backward's partial rows: one per (batch entry, chunk)

AUM_DEV vf convt_silu(vf a) { return a * vsigmoid(a); }

template <class T, bool BWD> struct ConvtLane {
    static constexpr int V = convt_vec<T, BWD>();
    vf w[CONVT_W][V];      // w[k][v]: tap k (right-aligned: taps 4 - width .. 3 are real) of channel c0 + v
    vf bias[V];
    vi c0;                 // first channel of the lane (clamped into the tensor for lanes past the end)
    vm live;
};
template <class T, bool BWD> AUM_DEV void convt_lane_setup(const AumConvTmArgs& a, int cb, ConvtLane<T, BWD>& ln) {
    constexpr int V = convt_vec<T, BWD>();
    const vi c = (lane_id() + cb * WAVE) * V;
    ln.live = c < a.dim;
    ln.c0 = vsel_i(ln.live, c, c * 0);
    if (a.width == CONVT_W) {
        // the four taps of a channel are 16 contiguous bytes: one access per channel instead of four
        const gbuf<float> wb = make_gbuf(a.weight);
        AUM_UNROLL
        for (int v = 0; v < V; ++v) {
            vf t[4];
            vq_unpack<float>(gbuf_load16(wb, (ln.c0 + v) * (CONVT_W * 4), 0), t);
            AUM_UNROLL
            for (int k = 0; k < CONVT_W; ++k) ln.w[k][v] = t[k];
        }
    } else {
        AUM_UNROLL
        for (int v = 0; v < V; ++v) {
            AUM_UNROLL
            for (int k = 0; k < CONVT_W; ++k) {
                const int kk = k - (CONVT_W - a.width);
                ln.w[k][v] = kk >= 0 ? gload_u(a.weight, (ln.c0 + v) * a.width + kk) : splat(0.f);
            }
        }
    }
    if (a.bias) {
        const gbuf<float> bb = make_gbuf(a.bias);
        AUM_UNROLL
        for (int i = 0; i < V / 4; ++i) {
            vf t[4];
            vq_unpack<float>(gbuf_load16(bb, ln.c0 * 4 + 16 * i, 0), t);
            AUM_UNROLL
            for (int k = 0; k < 4; ++k) ln.bias[4 * i + k] = t[k];
        }
    } else {
        AUM_UNROLL
        for (int v = 0; v < V; ++v) ln.bias[v] = splat(0.f);
    }
}

// unit = (batch entry, chunk of convt_tc(len) steps, block of 64 * V channels), channel block fastest
template <class T, bool SILU>
AUM_DEV void convt_fwd_wave(const AumConvTmArgs& a, int wg) {
    constexpr int V = convt_vec<T, false>(), ES = (int)sizeof(T);
    const int ncb = convt_cblocks<T, false>(a.dim), nch = convt_chunks<false>(a.len), L = a.len;
    const int cb = wg % ncb, ch = (wg / ncb) % nch, b = wg / (ncb * nch);
    const bool rev = (a.flags & AUM_CONV_REVERSE) != 0;
    ConvtLane<T, false> ln;
    convt_lane_setup<T, false>(a, cb, ln);
    const gbuf<T> xb = make_gbuf(row_ptr<T>(a.x, (int64_t)b * a.x_bs));
    const gbuf<T> yb = make_gbuf(row_ptr<T>(a.y, (int64_t)b * a.y_bs));
    const vi coff = ln.c0 * ES;
    const int x_tb = (int)a.x_ts * ES, y_tb = (int)a.y_ts * ES;
    auto tok = [&](int it) { return rev ? L - 1 - it : it; };
    const int tc = convt_tc<false>(L);
    const int it0 = ch * tc, it1 = it0 + tc < L ? it0 + tc : L;
    // channel pairs (2 p, 2 p + 1) of the lane on packed fp32 (the same IEEE operations in the same order as one by one: bit-equal), the
    // window as a ring indexed at compile time: at step j of a block x[it - 3 + k] sits in slot (k + j) & 3, the row entering at k = 3 takes
    // the slot of the row that left at k = 0 (round 6; the shifting window was 20 of ~90 vector-ALU slots per step)
    constexpr int NP = V / 2;
    vf2 w2[CONVT_W][NP], bias2[NP], xr[CONVT_W][NP];
    AUM_UNROLL
    for (int p = 0; p < NP; ++p) {
        bias2[p] = mk2(ln.bias[2 * p], ln.bias[2 * p + 1]);
        AUM_UNROLL
        for (int k = 0; k < CONVT_W; ++k) {
            w2[k][p] = mk2(ln.w[k][2 * p], ln.w[k][2 * p + 1]);
            xr[k][p] = spl2(splat(0.f));
        }
    }
    auto unpack2 = [&](const convt_raw<T, false>& q, vf2 (&o)[NP]) {
        vf t[V];
        convt_unpack<T, false>(q, t);
        AUM_UNROLL
        for (int p = 0; p < NP; ++p) o[p] = mk2(t[2 * p], t[2 * p + 1]);
    };
    // the window before the chunk: steps it0 - 3 .. it0 - 1 (zero padding before the sequence)
    AUM_UNROLL
    for (int k = 0; k < CONVT_W - 1; ++k) {
        const int it = it0 - (CONVT_W - 1) + k;
        if (it >= 0) unpack2(convt_load<T, false>(xb, coff, tok(it) * x_tb), xr[k]);
    }
    const bool all_live = a.dim % (WAVE * V) == 0;
    // Two blocks of CONVT_UB steps are in flight: the rows of the block after next are requested before a block is computed (a wave that
    // fetched a block, waited, computed and only then fetched again left the memory pipe idle for the arithmetic of every block -- there
    // are fewer than two waves per SIMD to fill the gap at the bench shape).  Requests past the chunk are clamped to its last row: no
    // conditions around the loads (the compiler would wait for ALL of them at the join).
    auto load_blk = [&](int itb, convt_raw<T, false> (&raw)[CONVT_UB]) {
        AUM_UNROLL
        for (int j = 0; j < CONVT_UB; ++j) {
            const int it = itb + j < it1 ? itb + j : it1 - 1;
            raw[j] = convt_load<T, false>(xb, coff, tok(it) * x_tb);
        }
    };
    static_assert(CONVT_UB % CONVT_W == 0, "a block returns the ring to its phase");
    auto comp_blk = [&](int itb, const convt_raw<T, false> (&raw)[CONVT_UB]) {
        AUM_UNROLL
        for (int j = 0; j < CONVT_UB; ++j) {
            if (itb + j < it1) {
                auto X = [&](int k) -> vf2 (&)[NP] { return xr[(k + j) & (CONVT_W - 1)]; };
                unpack2(raw[j], X(CONVT_W - 1));
                vf y[V];
                AUM_UNROLL
                for (int p = 0; p < NP; ++p) {
                    vf2 acc = bias2[p];
                    AUM_UNROLL
                    for (int k = 0; k < CONVT_W; ++k) acc = vfma2(w2[k][p], X(k)[p], acc);
                    if (SILU) acc = acc * vsigmoid2(acc);
                    y[2 * p] = lo2(acc);
                    y[2 * p + 1] = hi2(acc);
                }
                if (all_live) convt_store<T, false>(yb, coff, tok(itb + j) * y_tb, y);
                else convt_store_m<T, false>(yb, coff, tok(itb + j) * y_tb, y, ln.live);
            }
        }
    };
    convt_raw<T, false> ra[CONVT_UB], rb[CONVT_UB];
    load_blk(it0, ra);
    for (int itb = it0; itb < it1; itb += 2 * CONVT_UB) {
        load_blk(itb + CONVT_UB, rb);
        comp_blk(itb, ra);
        load_blk(itb + 2 * CONVT_UB, ra);
        comp_blk(itb + CONVT_UB, rb);
    }
}

// (The backward keeps one block of eight steps in flight: two tensors and a wider register window leave room for two blocks of four only,
// and that measured slower -- 2.18 against 2.00 ms per step of the bench.)
// backward: steps it1 - 1 down to it0; the three steps after the chunk are recomputed first (their dpre enters dx of the chunk's
// last steps), their dw / dbias terms belong to the next chunk.
// Round 6: the arithmetic on PAIRS of channels (v_pk_fma_f32 / v_pk_mul_f32: the same IEEE operations in the same order, two per issue
// slot -- bit-equal to the one-by-one form) and the windows as compile-time rings (the listing of round 5 was ~120 vector-ALU slots per step
// for four channels, 22 of them window moves: as much SIMD time as the kernel's bytes take at the copy rate).  A second copy of the block
// without per-step conditions was built too: 182 registers against 114 (the third wave per SIMD lost), not kept.
template <class T, bool SILU>
AUM_DEV void convt_bwd_wave(const AumConvTmArgs& a, int wg) {
    constexpr int V = convt_vec<T, true>(), NP = V / 2, ES = (int)sizeof(T);
    const int ncb = convt_cblocks<T, true>(a.dim), nch = convt_chunks<true>(a.len), L = a.len;
    const int cb = wg % ncb, ch = (wg / ncb) % nch, b = wg / (ncb * nch);
    const bool rev = (a.flags & AUM_CONV_REVERSE) != 0;
    ConvtLane<T, true> ln;
    convt_lane_setup<T, true>(a, cb, ln);
    const gbuf<T> xb = make_gbuf(row_ptr<T>(a.x, (int64_t)b * a.x_bs));
    const gbuf<T> gb = make_gbuf(row_ptr<T>(a.dy, (int64_t)b * a.dy_bs));
    const gbuf<T> dxb = make_gbuf(row_ptr<T>(a.dx, (int64_t)b * a.dx_bs));
    const vi coff = ln.c0 * ES;
    const int x_tb = (int)a.x_ts * ES, g_tb = (int)a.dy_ts * ES, dx_tb = (int)a.dx_ts * ES;
    auto tok = [&](int it) { return rev ? L - 1 - it : it; };
    const int tc = convt_tc<true>(L);
    const int it0 = ch * tc, it1 = it0 + tc < L ? it0 + tc : L;
    const int itop = it1 + (CONVT_W - 1) < L ? it1 + (CONVT_W - 1) : L;        // first step NOT recomputed
    const bool all_live = a.dim % (WAVE * V) == 0;                              // no lane past the last channel: stores need no mask
    // channel pairs (2 p, 2 p + 1) of the lane
    vf2 w2[CONVT_W][NP], bias2[NP];
    AUM_UNROLL
    for (int p = 0; p < NP; ++p) {
        bias2[p] = mk2(ln.bias[2 * p], ln.bias[2 * p + 1]);
        AUM_UNROLL
        for (int k = 0; k < CONVT_W; ++k) w2[k][p] = mk2(ln.w[k][2 * p], ln.w[k][2 * p + 1]);
    }
    auto unpack2 = [&](const convt_raw<T, true>& q, vf2 (&o)[NP]) {
        vf t[V];
        convt_unpack<T, true>(q, t);
        AUM_UNROLL
        for (int p = 0; p < NP; ++p) o[p] = mk2(t[2 * p], t[2 * p + 1]);
    };
    const vf2 zero2 = spl2(splat(0.f)), one2 = spl2(splat(1.f));
    // Windows as RINGS indexed at compile time (a shifting window is 12 register-pair moves per step wherever a step is conditional): at
    // step j of a block (it = itb - j) x[it - 3 + k] sits in slot (k - j) & 3 of xr, dpre[it + k] in slot (k - j) & 3 of dr; the row that
    // enters at k = 0 takes the slot of the row that left at k = 3, and a block of eight steps ends where it began.
    vf2 xr[CONVT_W][NP], dr[CONVT_W][NP], dw[CONVT_W][NP], db[NP];
    AUM_UNROLL
    for (int p = 0; p < NP; ++p) {
        db[p] = zero2;
        AUM_UNROLL
        for (int k = 0; k < CONVT_W; ++k) dw[k][p] = dr[k][p] = xr[k][p] = zero2;
    }
    // window of the first step processed (itop - 1): x[itop - 4 .. itop - 1]; its k = 0 row is loaded in the loop, rows 1..3 here
    AUM_UNROLL
    for (int k = 1; k < CONVT_W; ++k) {
        const int it = itop - 1 - (CONVT_W - 1) + k;
        if (it >= 0) unpack2(convt_load<T, true>(xb, coff, tok(it) * x_tb), xr[k]);
    }
    auto load_blk = [&](int itb, convt_raw<T, true> (&rx)[CONVT_UB], convt_raw<T, true> (&rg)[CONVT_UB]) {
        AUM_UNROLL
        for (int j = 0; j < CONVT_UB; ++j) {                    // requests below the chunk are clamped to its first row: no conditions around the loads
            const int it = itb - j >= it0 ? itb - j : it0;
            const int itx = it - (CONVT_W - 1) >= 0 ? it - (CONVT_W - 1) : 0;
            rx[j] = convt_load<T, true>(xb, coff, tok(itx) * x_tb);
            rg[j] = convt_load<T, true>(gb, coff, tok(it) * g_tb);
        }
    };
    auto comp_blk = [&](int itb, const convt_raw<T, true> (&rx)[CONVT_UB], const convt_raw<T, true> (&rg)[CONVT_UB]) {
        AUM_UNROLL
        for (int j = 0; j < CONVT_UB; ++j) {
            const int it = itb - j;
            if (it >= it0) {                                    // (false for the rest of the block once false: the rings' phase is never observed again)
                constexpr int W = CONVT_W;
                auto X = [&](int k) -> vf2 (&)[NP] { return xr[(k - j) & (W - 1)]; };
                auto Dp = [&](int k) -> vf2 (&)[NP] { return dr[(k - j) & (W - 1)]; };
                if (it - (W - 1) >= 0) {
                    unpack2(rx[j], X(0));
                } else {
                    AUM_UNROLL
                    for (int p = 0; p < NP; ++p) X(0)[p] = zero2;
                }
                vf2 g[NP], dxv[NP];
                unpack2(rg[j], g);
                const bool own = it < it1;
                AUM_UNROLL
                for (int p = 0; p < NP; ++p) {
                    vf2 d = g[p];
                    if (SILU) {
                        vf2 pre = bias2[p];
                        AUM_UNROLL
                        for (int k = 0; k < W; ++k) pre = vfma2(w2[k][p], X(k)[p], pre);
                        const vf2 sg = vsigmoid2(pre);
                        d = d * (sg * vfma2(pre, one2 - sg, one2));
                    }
                    Dp(0)[p] = d;
                    if (own) {
                        db[p] = db[p] + d;
                        AUM_UNROLL
                        for (int k = 0; k < W; ++k) dw[k][p] = vfma2(d, X(k)[p], dw[k][p]);
                        vf2 s = zero2;
                        AUM_UNROLL
                        for (int k = 0; k < W; ++k) s = vfma2(w2[k][p], Dp(W - 1 - k)[p], s);
                        dxv[p] = s;
                    }
                }
                if (own) {
                    vf o[V];
                    AUM_UNROLL
                    for (int p = 0; p < NP; ++p) {
                        o[2 * p] = lo2(dxv[p]);
                        o[2 * p + 1] = hi2(dxv[p]);
                    }
                    if (all_live) convt_store<T, true>(dxb, coff, tok(it) * dx_tb, o);
                    else convt_store_m<T, true>(dxb, coff, tok(it) * dx_tb, o, ln.live);
                }
            }
        }
    };
    static_assert(CONVT_UB % CONVT_W == 0, "a block returns the rings to their phase");
#if AUM_CONVT_BWD_BLOCKS == 2
    // two blocks in flight, as in the forward: with 8 bytes per lane the second block's 32 raw registers fit under the three-waves-per-SIMD limit
    convt_raw<T, true> ax[CONVT_UB], ag[CONVT_UB], bx[CONVT_UB], bg[CONVT_UB];
    load_blk(itop - 1, ax, ag);
    for (int itb = itop - 1; itb >= it0; itb -= 2 * CONVT_UB) {
        load_blk(itb - CONVT_UB, bx, bg);
        comp_blk(itb, ax, ag);
        load_blk(itb - 2 * CONVT_UB, ax, ag);
        comp_blk(itb - CONVT_UB, bx, bg);
    }
#else
    for (int itb = itop - 1; itb >= it0; itb -= CONVT_UB) {
        convt_raw<T, true> rx[CONVT_UB], rg[CONVT_UB];
        load_blk(itb, rx, rg);
        comp_blk(itb, rx, rg);
    }
#endif
    // partial rows of this wave: dw_part[part][dim][k] (the weight's own layout: a lane's 8 channels x 4 taps are 128 contiguous bytes, and the
    // sum over the parts is the gradient as it is -- round 3 wrote [part][k][dim] and paid a transposing copy per layer), db_part[part][dim]
    const int part = b * nch + ch;
    AUM_UNROLL
    for (int v = 0; v < V; ++v) {
        AUM_UNROLL
        for (int k = 0; k < CONVT_W; ++k) {
            const int kk = k - (CONVT_W - a.width);
            if (kk >= 0) gstore(a.dw_part + (int64_t)part * a.dim * a.width + kk, (ln.c0 + v) * a.width, (v & 1) ? hi2(dw[k][v >> 1]) : lo2(dw[k][v >> 1]), ln.live);
        }
        if (a.db_part) gstore(a.db_part + (int64_t)part * a.dim, ln.c0 + v, (v & 1) ? hi2(db[v >> 1]) : lo2(db[v >> 1]), ln.live);
    }
}

}  // namespace aum
