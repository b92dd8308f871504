// conv_rows_kernels.h -- width-4 depthwise conv for the AuM row shape: len = 512 + TL with 0 <= TL <= 8 (L = 513: 512 patches
// + the cls token), 16-bit activations.  Same arithmetic as conv4_{fwd,bwd}_wave (conv_norm_kernels.h, which remain the
// general path) but organised for the memory system:
//   * one wavefront walks CONVR_ROWS rows of ONE channel (consecutive batch entries: contiguous in the channel-major
//     layout) and loads row r+1 -- raw 16-byte fragments, converted only when used -- before it computes row r;
//   * a row is ONE pass: lanes own the 512 main steps (8 each, one 16-byte access), the TL tail steps are wave-uniform
//     work on v_readlane'd values instead of a second, almost empty 512-step block with its own memory round trip;
//   * no halo loads: the causal halo of lane 0 is the zero padding, the anti-causal halo of lane 63 is the tail;
//   * dweight / dbias: one atomic per wave (CONVR_ROWS rows) instead of one per row.
// References: MS:272 / causal_conv1d_fn (forward), SSI:594-596 call site (backward); AUM_CONV_REVERSE as in aum_hip.h.
#pragma once
#include "conv_norm_kernels.h"
#include "proj_kernels.h"

namespace aum {

#ifndef AUM_CONVR_ROWS
#define AUM_CONVR_ROWS 8
#endif
constexpr int CONVR_ROWS = AUM_CONVR_ROWS;      // rows (batch entries of one channel) per wavefront
constexpr int CONVR_MAIN = 512;    // steps owned by the lanes
constexpr int CONVR_MAXTAIL = 8;

AUM_HOSTDEV bool convr_shape_ok(int len) { return len >= CONVR_MAIN && len <= CONVR_MAIN + CONVR_MAXTAIL; }

struct ConvrRow { frag8 main; vi tail; };   // raw bits (converted when used): 8 main steps of this lane; lane j < TL: tail step j

#ifndef AUM_EMU
template <class T> AUM_DEV vi gload_bits16(const T* p, vi idx, vm m) { return m ? (int)p[idx].bits : 0; }
template <class T> AUM_DEV vf bits16_to_f32(T, vi b) { T e; e.bits = (uint16_t)b; return elem_to_f32(e); }
#else
template <class T> inline vi gload_bits16(const T* p, const vi& idx, const vm& m) { vi r; AUM_LANES r.v[l] = m.v[l] ? (int)p[idx.v[l]].bits : 0; return r; }
template <class T> inline vf bits16_to_f32(T, const vi& b) { vf r; AUM_LANES { T e; e.bits = (uint16_t)b.v[l]; r.v[l] = elem_to_f32(e); } return r; }
#endif

template <class T> AUM_DEV ConvrRow convr_load(const T* rp, int tl) {
    const vi lane = lane_id();
    ConvrRow r;
    r.main = gload_frag(rp, lane * 8);
    r.tail = gload_bits16(rp + CONVR_MAIN, vmin_i(lane, CONVR_MAXTAIL - 1), lane < tl);
    return r;
}
template <class T> AUM_DEV vf convr_tail_f32(T t, const ConvrRow& r) { return bits16_to_f32(t, r.tail); }
AUM_DEV vf convr_silu(vf a) { return a * vsigmoid(a); }

// pre-activations of the 8 main steps from the 11 inputs a lane touches (causal: x[t0-3..t0+7]; REV: x[t0..t0+10])
template <bool REV> AUM_DEV void convr_pre8(const vf (&xin)[11], const float (&w)[4], float bias, vf (&acc)[8]) {
    AUM_UNROLL
    for (int j = 0; j < 8; ++j) {
        vf a = splat(bias);
        AUM_UNROLL
        for (int k = 0; k < 4; ++k) a = vfma(xin[REV ? j + 3 - k : j + k], splat(w[k]), a);
        acc[j] = a;
    }
}
// z[0..10]: the wave-uniform window around the tail.  causal: x[509..519] (lane 63's last three steps, then the tail);
// REV: x[512..522] (the tail, zeros beyond the row).
template <bool REV> AUM_DEV void convr_tail_window(const vf (&x)[8], vf tailv, int tl, float (&z)[11]) {
    AUM_UNROLL
    for (int i = 0; i < 11; ++i) z[i] = 0.f;
    if (!REV) {
        AUM_UNROLL
        for (int i = 0; i < 3; ++i) z[i] = readlane(x[5 + i], WAVE - 1);
        AUM_UNROLL
        for (int j = 0; j < CONVR_MAXTAIL; ++j) if (j < tl) z[3 + j] = readlane(tailv, j);
    } else {
        AUM_UNROLL
        for (int j = 0; j < CONVR_MAXTAIL; ++j) if (j < tl) z[j] = readlane(tailv, j);
    }
}
template <bool REV> AUM_DEV float convr_tail_pre(const float (&z)[11], int j, const float (&w)[4], float bias) {
    float a = bias;
    AUM_UNROLL
    for (int k = 0; k < 4; ++k) a = vfma(z[REV ? j + 3 - k : j + k], w[k], a);
    return a;
}
template <bool REV> AUM_DEV void convr_inputs(const vf (&x)[8], const float (&z)[11], vf (&xin)[11]) {
    if (!REV) {
        AUM_UNROLL
        for (int i = 0; i < 3; ++i) xin[i] = dpp_wave_shr1(x[5 + i], splat(0.f));           // t < 0: zero padding
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) xin[3 + j] = x[j];
    } else {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) xin[j] = x[j];
        AUM_UNROLL
        for (int i = 0; i < 3; ++i) xin[8 + i] = dpp_wave_shl1(x[i], splat(z[i]));          // lane 63: the tail steps
    }
}
template <class T> AUM_DEV void convr_store(T* rp, const vf (&y)[8], vf ytail, int tl) {
    const vi lane = lane_id();
    gstore_frag(rp, lane * 8, f32_to_frag(T{}, y));
    gstore(rp + CONVR_MAIN, vmin_i(lane, CONVR_MAXTAIL - 1), ytail, lane < tl);
}

template <class T, bool REV> AUM_DEV void conv4_rows_fwd_wave(const AumConvArgs& p, int wg) {
    const int ngrp = (p.batch + CONVR_ROWS - 1) / CONVR_ROWS;
    const int e = wg / ngrp, b0 = (wg % ngrp) * CONVR_ROWS;
    const int nrows = p.batch - b0 < CONVR_ROWS ? p.batch - b0 : CONVR_ROWS;
    const int tl = p.len - CONVR_MAIN;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xb = (const T*)p.x + (int64_t)b0 * p.x_bs + (int64_t)e * p.x_ds;
    T* yb = (T*)p.y + (int64_t)b0 * p.y_bs + (int64_t)e * p.y_ds;
    const float w[4] = {p.weight[e * 4 + 0], p.weight[e * 4 + 1], p.weight[e * 4 + 2], p.weight[e * 4 + 3]};
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    ConvrRow ra = convr_load(xb, tl), rb = ra;
#define CONVR_FWD_ROW(R, CUR, NXT)                                                                   \
    {                                                                                                \
        if ((R) + 1 < nrows) NXT = convr_load(xb + (int64_t)((R) + 1) * p.x_bs, tl);                 \
        vf x[8], xin[11], acc[8], y[8];                                                              \
        float z[11];                                                                                 \
        frag_to_f32(T{}, CUR.main, x);                                                               \
        convr_tail_window<REV>(x, convr_tail_f32(T{}, CUR), tl, z);                                  \
        convr_inputs<REV>(x, z, xin);                                                                \
        convr_pre8<REV>(xin, w, bias, acc);                                                          \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < 8; ++j) y[j] = silu ? convr_silu(acc[j]) : acc[j];                       \
        vf ytail = splat(0.f);                                                                       \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < CONVR_MAXTAIL; ++j) {                                                    \
            if (j < tl) {                                                                            \
                const vf a = splat(convr_tail_pre<REV>(z, j, w, bias));                              \
                ytail = vsel(lane == j, silu ? convr_silu(a) : a, ytail);                            \
            }                                                                                        \
        }                                                                                            \
        convr_store(yb + (int64_t)(R) * p.y_bs, y, ytail, tl);                                       \
    }
    for (int r = 0; r < nrows; r += 2) {
        CONVR_FWD_ROW(r, ra, rb)
        if (r + 1 < nrows) CONVR_FWD_ROW(r + 1, rb, ra)
    }
#undef CONVR_FWD_ROW
}

template <class T, bool REV> AUM_DEV void conv4_rows_bwd_wave(const AumConvArgs& p, int wg) {
    const int ngrp = (p.batch + CONVR_ROWS - 1) / CONVR_ROWS;
    const int e = wg / ngrp, b0 = (wg % ngrp) * CONVR_ROWS;
    const int nrows = p.batch - b0 < CONVR_ROWS ? p.batch - b0 : CONVR_ROWS;
    const int tl = p.len - CONVR_MAIN;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xb = (const T*)p.x + (int64_t)b0 * p.x_bs + (int64_t)e * p.x_ds;
    const T* gb = (const T*)p.dy + (int64_t)b0 * p.dy_bs + (int64_t)e * p.dy_ds;
    T* dxb = (T*)p.dx + (int64_t)b0 * p.dx_bs + (int64_t)e * p.dx_ds;
    const float w[4] = {p.weight[e * 4 + 0], p.weight[e * 4 + 1], p.weight[e * 4 + 2], p.weight[e * 4 + 3]};
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    vf dw[4] = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    vf db = splat(0.f);
    float dwt[4] = {0.f, 0.f, 0.f, 0.f}, dbt = 0.f;        // tail steps: wave-uniform
    ConvrRow xa = convr_load(xb, tl), xbb = xa, ga = convr_load(gb, tl), gbb = ga;
#define CONVR_BWD_ROW(R, XC, GC, XN, GN)                                                             \
    {                                                                                                \
        if ((R) + 1 < nrows) {                                                                       \
            XN = convr_load(xb + (int64_t)((R) + 1) * p.x_bs, tl);                                   \
            GN = convr_load(gb + (int64_t)((R) + 1) * p.dy_bs, tl);                                  \
        }                                                                                            \
        vf x[8], g[8], xin[11], acc[8], dpre[8], dxv[8];                                             \
        float z[11], dz[11];                                                                         \
        frag_to_f32(T{}, XC.main, x);                                                                \
        frag_to_f32(T{}, GC.main, g);                                                                \
        convr_tail_window<REV>(x, convr_tail_f32(T{}, XC), tl, z);                                   \
        convr_inputs<REV>(x, z, xin);                                                                \
        convr_pre8<REV>(xin, w, bias, acc);                                                          \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < 8; ++j) {                                                                \
            vf d = g[j];                                                                             \
            if (silu) d = d * vsilu_grad(acc[j]);                                                    \
            dpre[j] = d;                                                                             \
            db = db + d;                                                                             \
            AUM_UNROLL                                                                               \
            for (int k = 0; k < 4; ++k) dw[k] = vfma(xin[REV ? j + 3 - k : j + k], d, dw[k]);        \
        }                                                                                            \
        /* tail steps: dpre, their weight/bias terms; dz = the same window as z but of dpre */       \
        const vf gtail = convr_tail_f32(T{}, GC);                                                    \
        float dt[CONVR_MAXTAIL];                                                                     \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < CONVR_MAXTAIL; ++j) {                                                    \
            dt[j] = 0.f;                                                                             \
            if (j < tl) {                                                                            \
                float d = readlane(gtail, j);                                                        \
                if (silu) {                                                                          \
                    const float a = convr_tail_pre<REV>(z, j, w, bias);                              \
                    const float sg = vrcp(1.0f + vexp2(a * (-LOG2E)));                               \
                    d = d * (sg * vfma(a, 1.f - sg, 1.f));                                           \
                }                                                                                    \
                dt[j] = d;                                                                           \
                dbt += d;                                                                            \
                AUM_UNROLL                                                                           \
                for (int k = 0; k < 4; ++k) dwt[k] = vfma(z[REV ? j + 3 - k : j + k], d, dwt[k]);    \
            }                                                                                        \
        }                                                                                            \
        AUM_UNROLL                                                                                   \
        for (int i = 0; i < 11; ++i) dz[i] = 0.f;                                                    \
        vf dext[11];   /* causal: dpre[t0 .. t0+10];  REV: dpre[t0-3 .. t0+7] */                     \
        if (!REV) {                                                                                  \
            AUM_UNROLL                                                                               \
            for (int j = 0; j < 8; ++j) dext[j] = dpre[j];                                           \
            AUM_UNROLL                                                                               \
            for (int i = 0; i < 3; ++i) dext[8 + i] = dpp_wave_shl1(dpre[i], splat(i < tl ? dt[i] : 0.f)); \
            AUM_UNROLL                                                                               \
            for (int j = 0; j < CONVR_MAXTAIL; ++j) dz[j] = dt[j];                 /* dpre[512 + j] */ \
        } else {                                                                                     \
            AUM_UNROLL                                                                               \
            for (int i = 0; i < 3; ++i) dext[i] = dpp_wave_shr1(dpre[5 + i], splat(0.f));            \
            AUM_UNROLL                                                                               \
            for (int j = 0; j < 8; ++j) dext[3 + j] = dpre[j];                                       \
            AUM_UNROLL                                                                               \
            for (int i = 0; i < 3; ++i) dz[i] = readlane(dpre[5 + i], WAVE - 1);   /* dpre[509 + i] */ \
            AUM_UNROLL                                                                               \
            for (int j = 0; j < CONVR_MAXTAIL; ++j) dz[3 + j] = dt[j];                               \
        }                                                                                            \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < 8; ++j) {                                                                \
            /* causal: dx[s] = sum_w W[w] dpre[s+3-w] -> dext[j+3-w];  REV: dx[s] = sum_w W[w] dpre[s-3+w] -> dext[j+w] */ \
            vf a = splat(0.f);                                                                       \
            AUM_UNROLL                                                                               \
            for (int k = 0; k < 4; ++k) a = vfma(dext[REV ? j + k : j + 3 - k], splat(w[k]), a);     \
            dxv[j] = a;                                                                              \
        }                                                                                            \
        vf dxtail = splat(0.f);                                                                      \
        AUM_UNROLL                                                                                   \
        for (int j = 0; j < CONVR_MAXTAIL; ++j) {                                                    \
            /* causal: dz holds dpre[512..], dx[512+j] = sum_w W[w] dz[j+3-w];  REV: dz holds dpre[509..], sum_w W[w] dz[j+w] */ \
            if (j < tl) {                                                                            \
                float a = 0.f;                                                                       \
                AUM_UNROLL                                                                           \
                for (int k = 0; k < 4; ++k) a = vfma(dz[REV ? j + k : j + 3 - k], w[k], a);          \
                dxtail = vsel(lane == j, splat(a), dxtail);                                          \
            }                                                                                        \
        }                                                                                            \
        convr_store(dxb + (int64_t)(R) * p.dx_bs, dxv, dxtail, tl);                                  \
    }
    for (int r = 0; r < nrows; r += 2) {
        CONVR_BWD_ROW(r, xa, ga, xbb, gbb)
        if (r + 1 < nrows) CONVR_BWD_ROW(r + 1, xbb, gbb, xa, ga)
    }
#undef CONVR_BWD_ROW
    AUM_UNROLL
    for (int k = 0; k < 4; ++k)
        gatomic_add(p.dweight + (int64_t)e * 4 + k, spl_i(0), splat(wave_sum(dw[k]) + dwt[k]), lane == 0);
    if (p.bias && p.dbias) gatomic_add(p.dbias + e, spl_i(0), splat(wave_sum(db) + dbt), lane == 0);
}

}  // namespace aum
