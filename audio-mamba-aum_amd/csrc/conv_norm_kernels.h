// conv_norm_kernels.h -- depthwise causal conv1d (+bias +SiLU) and fused residual-add + RMSNorm,
// forward and backward, one wavefront per workgroup (wave.h).
//
// Replaces causal_conv1d_cuda.causal_conv1d_fwd/_bwd (call sites SSI:463, 532, 594-596; arithmetic
// MS:272) and the Triton kernels _layer_norm_fwd_1pass_kernel / _layer_norm_bwd_kernel (LN:51-120,
// 180-290).  Both are HBM-bound: lanes run along the unit-stride axis so every wave instruction
// touches one contiguous span.
#pragma once
#include "../../include/aum_hip.h"
#include "wave.h"

namespace aum {

constexpr int CONV_MAX_W = 8;
constexpr int CONV_LDS_FLOATS = WAVE + 2 * CONV_MAX_W;

AUM_DEV vf vsilu_grad(vf pre) {  // d/dpre [pre * sigmoid(pre)]
    const vf sg = vsigmoid(pre);
    return sg * vfma(pre, splat(1.f) - sg, splat(1.f));
}

// y[l] = act(bias + sum_w W[w] * x[l -/+ (W-1-w)])   (causal / AUM_CONV_REVERSE anti-causal)
template <class T> AUM_DEV vf conv_pre(const T* xp, const float* w, float bias, int W, int sgn, vi l, int len) {
    vf acc = splat(bias);
    for (int j = 0; j < W; ++j) {
        const vi src = l + sgn * (W - 1 - j);
        const vm m = (src >= 0) && (src < len);
        acc = vfma(gload(xp, src, m), splat(w[j]), acc);
    }
    return acc;
}

template <class T> AUM_DEV void conv_fwd_wave(const AumConvArgs& p, int wg) {
    const int b = wg / p.dim, e = wg % p.dim;
    const int W = p.width;
    const int sgn = (p.flags & AUM_CONV_REVERSE) ? 1 : -1;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xp = (const T*)p.x + (int64_t)b * p.x_bs + (int64_t)e * p.x_ds;
    T* yp = (T*)p.y + (int64_t)b * p.y_bs + (int64_t)e * p.y_ds;
    const float* w = p.weight + (int64_t)e * W;
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    for (int l0 = 0; l0 < p.len; l0 += WAVE) {
        const vi l = lane + l0;
        const vm m = l < p.len;
        vf pre = conv_pre(xp, w, bias, W, sgn, l, p.len);
        if (silu) pre = pre * vsigmoid(pre);
        gstore(yp, l, pre, m);
    }
}

// dx[s] = sum_w W[w] * dpre[s +/- (W-1-w)],  dpre = dy * act'(pre);  dW[w] += sum_l x[l -/+ (W-1-w)] dpre[l]
template <class T> AUM_DEV void conv_bwd_wave(const AumConvArgs& p, int wg, float* lds) {
    const int b = wg / p.dim, e = wg % p.dim;
    const int W = p.width;
    const int sgn = (p.flags & AUM_CONV_REVERSE) ? 1 : -1;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xp = (const T*)p.x + (int64_t)b * p.x_bs + (int64_t)e * p.x_ds;
    const T* gp = (const T*)p.dy + (int64_t)b * p.dy_bs + (int64_t)e * p.dy_ds;
    T* dxp = (T*)p.dx + (int64_t)b * p.dx_bs + (int64_t)e * p.dx_ds;
    const float* w = p.weight + (int64_t)e * W;
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    vf dw[CONV_MAX_W];
    AUM_UNROLL
    for (int j = 0; j < CONV_MAX_W; ++j) dw[j] = splat(0.f);
    vf db = splat(0.f);
    // LDS window holds dpre for l in [l0 - H, l0 + 64 + H), H = W-1, at index (l - l0 + H)
    const int H = W - 1;
    for (int l0 = 0; l0 < p.len; l0 += WAVE) {
        wave_sync();
        for (int part = 0; part < 2; ++part) {
            // part 0: the 64 main positions; part 1: the 2H halo positions (lanes < 2H)
            const vi off = part == 0 ? lane + H : vsel_i(lane < H, lane, lane + WAVE);   // index into window
            const vi l = off + (l0 - H);
            const vm active = part == 0 ? (lane >= 0) : (lane < 2 * H);
            const vm m = active && (l >= 0) && (l < p.len);
            vf pre = conv_pre(xp, w, bias, W, sgn, l, p.len);
            vf dpre = gload(gp, l, m);
            if (silu) dpre = dpre * vsilu_grad(pre);
            dpre = vsel(m, dpre, splat(0.f));
            // inactive lanes of part 1 must not write: redirect them to a scratch slot past the window
            lds_write(lds, vsel_i(active, off, spl_i(WAVE + 2 * CONV_MAX_W - 1)), dpre);
            if (part == 0) {
                db = db + dpre;
                AUM_UNROLL
                for (int j = 0; j < CONV_MAX_W; ++j) {   // static register indexing; W is wave-uniform
                    if (j < W) {
                        const vi src = l + sgn * (W - 1 - j);
                        const vm ms = m && (src >= 0) && (src < p.len);
                        dw[j] = vfma(gload(xp, src, ms), dpre, dw[j]);
                    }
                }
            }
        }
        wave_sync();
        const vi s = lane + l0;
        vf acc = splat(0.f);
        for (int j = 0; j < W; ++j) {
            // y[l] reads x[l + sgn*(W-1-j)]  =>  dx[s] += W[j] * dpre[s - sgn*(W-1-j)]
            const vi widx = lane + H - sgn * (W - 1 - j);
            acc = vfma(lds_read(lds, widx), splat(w[j]), acc);
        }
        gstore(dxp, s, acc, s < p.len);
    }
    AUM_UNROLL
    for (int j = 0; j < CONV_MAX_W; ++j) {
        if (j < W) {
            const float sum = wave_sum(dw[j]);
            gatomic_add(p.dweight + (int64_t)e * W + j, spl_i(0), splat(sum), lane == 0);
        }
    }
    if (p.bias && p.dbias) gatomic_add(p.dbias + e, spl_i(0), splat(wave_sum(db)), lane == 0);
}

// ------------------------------------------------------------------------------------------------
// Fused residual-add + RMSNorm.  One wave per row (forward) / per block of rows (backward).
// ------------------------------------------------------------------------------------------------
template <class TX, class TR> AUM_DEV void rmsnorm_fwd_wave(const AumNormArgs& p, int wg) {
    const int row = wg;
    const TX* xp = (const TX*)p.x + (int64_t)row * p.row_stride_x;
    const TR* rp = p.residual ? (const TR*)p.residual + (int64_t)row * p.row_stride_res : nullptr;
    TR* rop = p.residual_out ? (TR*)p.residual_out + (int64_t)row * p.row_stride_res_out : nullptr;
    TX* yp = (TX*)p.y + (int64_t)row * p.row_stride_y;
    const vi lane = lane_id();
    vf ss = splat(0.f);
    for (int c0 = 0; c0 < p.cols; c0 += WAVE) {
        const vi c = lane + c0;
        const vm m = c < p.cols;
        vf v = gload(xp, c, m);
        if (rp) v = v + gload(rp, c, m);
        if (rop) gstore(rop, c, v, m);
        ss = vfma(v, v, ss);
    }
    const float mean_sq = wave_sum(ss) / (float)p.cols;
    const float rstd = readlane(vdiv(splat(1.f), vsqrt(splat(mean_sq + p.eps))), 0);   // 1/sqrt as LN:44
    if (p.rstd_out) gstore(p.rstd_out + row, spl_i(0), splat(rstd), lane == 0);
    for (int c0 = 0; c0 < p.cols; c0 += WAVE) {
        const vi c = lane + c0;
        const vm m = c < p.cols;
        vf v = gload(xp, c, m);
        if (rp) v = v + gload(rp, c, m);
        gstore(yp, c, v * rstd * gload(p.weight, c, m), m);
    }
}

// lds: cols floats of per-lane-owned dweight accumulators (column c is only ever touched by lane c%64)
template <class TX, class TR> AUM_DEV void rmsnorm_bwd_wave(const AumNormArgs& p, int wg, int n_partials, float* lds) {
    const int rows_per = (p.rows + n_partials - 1) / n_partials;
    const int r0 = wg * rows_per;
    const int r1 = r0 + rows_per < p.rows ? r0 + rows_per : p.rows;
    const vi lane = lane_id();
    for (int c0 = 0; c0 < p.cols; c0 += WAVE) lds_write(lds, lane + c0, splat(0.f));   // lds holds ceil(cols/64)*64 floats
    for (int row = r0; row < r1; ++row) {
        const TR* xp = (const TR*)p.x + (int64_t)row * p.row_stride_x;   // saved residual_out (pre-norm input)
        const TX* gp = (const TX*)p.dy + (int64_t)row * p.row_stride_dy;
        const TR* drp = p.dresidual_out ? (const TR*)p.dresidual_out + (int64_t)row * p.row_stride_dres_out : nullptr;
        TX* dxp = (TX*)p.dx + (int64_t)row * p.row_stride_dx;
        TR* drip = p.dresidual_in ? (TR*)p.dresidual_in + (int64_t)row * p.row_stride_dres_in : nullptr;
        const float rstd = p.rstd_in[row];
        vf c1 = splat(0.f);
        for (int c0 = 0; c0 < p.cols; c0 += WAVE) {
            const vi c = lane + c0;
            const vm m = c < p.cols;
            const vf xhat = gload(xp, c, m) * rstd;
            const vf dyv = gload(gp, c, m);
            c1 = vfma(xhat * gload(p.weight, c, m), dyv, c1);
            const vf acc = lds_read(lds, c);
            lds_write(lds, c, vfma(dyv, xhat, acc));
        }
        const float cm = wave_sum(c1) / (float)p.cols;
        for (int c0 = 0; c0 < p.cols; c0 += WAVE) {
            const vi c = lane + c0;
            const vm m = c < p.cols;
            const vf xhat = gload(xp, c, m) * rstd;
            vf g = (gload(p.weight, c, m) * gload(gp, c, m) - xhat * cm) * rstd;
            if (drp) g = g + gload(drp, c, m);
            gstore(dxp, c, g, m);
            if (drip) gstore(drip, c, g, m);
        }
    }
    float* out = p.dweight_partial + (int64_t)wg * p.cols;
    for (int c0 = 0; c0 < p.cols; c0 += WAVE) {
        const vi c = lane + c0;
        gstore(out, c, lds_read(lds, c), c < p.cols);
    }
}

// ------------------------------------------------------------------------------------------------
// Width-4 conv (the only width Mamba uses, MS:39): 8 consecutive time steps per lane, one 16-byte access per
// tensor per lane, halo from the neighbouring lane through DPP.  HBM-bound: 2 (fwd) / 3 (bwd) tensor passes.
// ------------------------------------------------------------------------------------------------
// Load this lane's 8 elements [t0, t0+8) of a row (vector load when fully inside the row, masked scalars at the tail).
template <class T> AUM_DEV void row_load8(const T* rp, vi t0, int len, vf (&x)[8]) {
    const vm full = (t0 + 8) <= len;
    gload8(rp, t0, full, x);
    const vm part = !full && (t0 < len);
    if (any_lane(part)) {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) x[j] = x[j] + gload(rp, t0 + j, part && ((t0 + j) < len));
    }
}
template <class T> AUM_DEV void row_store8(T* rp, vi t0, int len, const vf (&x)[8]) {
    const vm full = (t0 + 8) <= len;
    gstore8(rp, t0, x, full);
    const vm part = !full && (t0 < len);
    if (any_lane(part)) {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) gstore(rp, t0 + j, x[j], part && ((t0 + j) < len));
    }
}

// xin[0..10]: the 11 inputs that the lane's 8 outputs touch.  causal: x[t0-3 .. t0+7]; REV: x[t0 .. t0+10].
template <class T, bool REV> AUM_DEV void conv4_inputs(const T* xp, vi t0, int len, vf (&xin)[11]) {
    const vi lane = lane_id();
    vf x[8];
    row_load8(xp, t0, len, x);
    if (!REV) {
        AUM_UNROLL
        for (int i = 0; i < 3; ++i) {
            const vi th = t0 - 3 + i;
            const vf edge = gload(xp, th, (lane == 0) && (th >= 0) && (th < len));
            xin[i] = dpp_wave_shr1(x[5 + i], edge);
        }
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) xin[3 + j] = x[j];
    } else {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) xin[j] = x[j];
        AUM_UNROLL
        for (int i = 0; i < 3; ++i) {
            const vi th = t0 + 8 + i;
            const vf edge = gload(xp, th, (lane == WAVE - 1) && (th < len));
            xin[8 + i] = dpp_wave_shl1(x[i], edge);
        }
    }
}

template <class T, bool REV> AUM_DEV void conv4_fwd_wave(const AumConvArgs& p, int wg) {
    const int b = wg / p.dim, e = wg % p.dim;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xp = (const T*)p.x + (int64_t)b * p.x_bs + (int64_t)e * p.x_ds;
    T* yp = (T*)p.y + (int64_t)b * p.y_bs + (int64_t)e * p.y_ds;
    const float w0 = p.weight[e * 4 + 0], w1 = p.weight[e * 4 + 1], w2 = p.weight[e * 4 + 2], w3 = p.weight[e * 4 + 3];
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    for (int blk = 0; blk * 512 < p.len; ++blk) {
        const vi t0 = lane * 8 + blk * 512;
        vf xin[11], y[8];
        conv4_inputs<T, REV>(xp, t0, p.len, xin);
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) {
            // causal: y[l] = b + sum_w W[w] x[l-3+w] -> xin[j+w];  REV: y[l] = b + sum_w W[w] x[l+3-w] -> xin[j+3-w]
            vf acc = splat(bias);
            acc = vfma(xin[REV ? j + 3 : j + 0], splat(w0), acc);
            acc = vfma(xin[REV ? j + 2 : j + 1], splat(w1), acc);
            acc = vfma(xin[REV ? j + 1 : j + 2], splat(w2), acc);
            acc = vfma(xin[REV ? j + 0 : j + 3], splat(w3), acc);
            y[j] = silu ? acc * vsigmoid(acc) : acc;
        }
        row_store8(yp, t0, p.len, y);
    }
}

template <class T, bool REV> AUM_DEV void conv4_bwd_wave(const AumConvArgs& p, int wg) {
    const int b = wg / p.dim, e = wg % p.dim;
    const bool silu = (p.flags & AUM_CONV_SILU) != 0;
    const T* xp = (const T*)p.x + (int64_t)b * p.x_bs + (int64_t)e * p.x_ds;
    const T* gp = (const T*)p.dy + (int64_t)b * p.dy_bs + (int64_t)e * p.dy_ds;
    T* dxp = (T*)p.dx + (int64_t)b * p.dx_bs + (int64_t)e * p.dx_ds;
    const float w[4] = {p.weight[e * 4 + 0], p.weight[e * 4 + 1], p.weight[e * 4 + 2], p.weight[e * 4 + 3]};
    const float bias = p.bias ? p.bias[e] : 0.f;
    const vi lane = lane_id();
    const int nblk = (p.len + 511) / 512;
    vf dw[4] = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    vf db = splat(0.f);
    // dx needs dpre of the 3 steps AFTER (causal) / BEFORE (REV) the lane's own 8: neighbour lane via DPP, and across
    // the 512-step blocks a wave-uniform carry -- so blocks are walked against the dependency.
    float carry[3] = {0.f, 0.f, 0.f};
    for (int bi = 0; bi < nblk; ++bi) {
        const int blk = REV ? bi : nblk - 1 - bi;
        const vi t0 = lane * 8 + blk * 512;
        vf xin[11], g[8], dpre[8], dxv[8];
        conv4_inputs<T, REV>(xp, t0, p.len, xin);
        row_load8(gp, t0, p.len, g);
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) {
            vf acc = splat(bias);
            AUM_UNROLL
            for (int k = 0; k < 4; ++k) acc = vfma(xin[REV ? j + 3 - k : j + k], splat(w[k]), acc);
            vf d = g[j];
            if (silu) d = d * vsilu_grad(acc);
            d = vsel((t0 + j) < p.len, d, splat(0.f));
            dpre[j] = d;
            db = db + d;
            AUM_UNROLL
            for (int k = 0; k < 4; ++k) dw[k] = vfma(xin[REV ? j + 3 - k : j + k], d, dw[k]);
        }
        vf dext[11];   // causal: dpre[t0 .. t0+10];  REV: dpre[t0-3 .. t0+7]
        if (!REV) {
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) dext[j] = dpre[j];
            AUM_UNROLL
            for (int i = 0; i < 3; ++i) dext[8 + i] = dpp_wave_shl1(dpre[i], splat(carry[i]));
            AUM_UNROLL
            for (int i = 0; i < 3; ++i) carry[i] = readlane(dpre[i], 0);
        } else {
            AUM_UNROLL
            for (int i = 0; i < 3; ++i) dext[i] = dpp_wave_shr1(dpre[5 + i], splat(carry[i]));
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) dext[3 + j] = dpre[j];
            AUM_UNROLL
            for (int i = 0; i < 3; ++i) carry[i] = readlane(dpre[5 + i], WAVE - 1);
        }
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) {
            // causal: dx[s] = sum_w W[w] dpre[s+3-w] -> dext[j+3-w];  REV: dx[s] = sum_w W[w] dpre[s-3+w] -> dext[j+w]
            vf acc = splat(0.f);
            AUM_UNROLL
            for (int k = 0; k < 4; ++k) acc = vfma(dext[REV ? j + k : j + 3 - k], splat(w[k]), acc);
            dxv[j] = acc;
        }
        row_store8(dxp, t0, p.len, dxv);
    }
    AUM_UNROLL
    for (int k = 0; k < 4; ++k)
        gatomic_add(p.dweight + (int64_t)e * 4 + k, spl_i(0), splat(wave_sum(dw[k])), lane == 0);
    if (p.bias && p.dbias) gatomic_add(p.dbias + e, spl_i(0), splat(wave_sum(db)), lane == 0);
}

// ------------------------------------------------------------------------------------------------
// RMSNorm, vectorised: each lane owns 8 consecutive columns per 512-column chunk, rows cached in registers
// (cols <= 512*NCH), one pass over memory.
// ------------------------------------------------------------------------------------------------
template <class TX, class TR, int NCH> AUM_DEV void rmsnorm_fwd_vec(const AumNormArgs& p, int wg) {
    const int row = wg;
    const TX* xp = (const TX*)p.x + (int64_t)row * p.row_stride_x;
    const TR* rp = p.residual ? (const TR*)p.residual + (int64_t)row * p.row_stride_res : nullptr;
    TR* rop = p.residual_out ? (TR*)p.residual_out + (int64_t)row * p.row_stride_res_out : nullptr;
    TX* yp = (TX*)p.y + (int64_t)row * p.row_stride_y;
    const vi lane = lane_id();
    vf v[NCH][8];
    vf ss = splat(0.f);
    AUM_UNROLL
    for (int c = 0; c < NCH; ++c) {
        const vi c0 = lane * 8 + c * 512;
        row_load8(xp, c0, p.cols, v[c]);
        if (rp) {
            vf r[8];
            row_load8(rp, c0, p.cols, r);
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) v[c][j] = v[c][j] + r[j];
        }
        if (rop) row_store8(rop, c0, p.cols, v[c]);
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) ss = vfma(v[c][j], v[c][j], ss);
    }
    const float mean_sq = wave_sum(ss) / (float)p.cols;
    const float rstd = readlane(vdiv(splat(1.f), vsqrt(splat(mean_sq + p.eps))), 0);
    if (p.rstd_out) gstore(p.rstd_out + row, spl_i(0), splat(rstd), lane == 0);
    AUM_UNROLL
    for (int c = 0; c < NCH; ++c) {
        const vi c0 = lane * 8 + c * 512;
        vf wv[8], y[8];
        row_load8(p.weight, c0, p.cols, wv);
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) y[j] = v[c][j] * rstd * wv[j];
        row_store8(yp, c0, p.cols, y);
    }
}

// The backward's waves: one per ~9 rows (aum_rmsnorm_bwd_partials(rows) of them), the weight-gradient terms of a wave's rows in registers.
// Round 6: NORM_BWD_NW waves form a workgroup and add their sums through LDS in wave order before one row of partials leaves -- 512 partial
// rows instead of 4 096 for the caller to add (one launch of aum_sum_rows instead of two: 12 -> 4 us per layer).
constexpr int NORM_BWD_NW = 8;
template <class TX, class TR, int NCH> AUM_DEV void rmsnorm_bwd_vec(const AumNormArgs& p, int wg, int n_waves, float* lds) {
    // rows dealt evenly: 64 x 513 rows on 4 096 waves are 8 rows each and a ninth for the first 64 (ceil(rows / waves) = 9 rows each left
    // 448 of the 4 096 waves without any)
    const int base = p.rows / n_waves, rem = p.rows - base * n_waves;
    const vi lane = lane_id();
    vf dwacc[AUM_PER_WAVE(NORM_BWD_NW)][NCH][8];
    AUM_FOR_EACH_WAVE(w, NORM_BWD_NW) {
    const int sub = wg * NORM_BWD_NW + w;
    const int r0 = sub < n_waves ? sub * base + (sub < rem ? sub : rem) : p.rows;
    const int r1 = sub < n_waves ? r0 + base + (sub < rem ? 1 : 0) : p.rows;
    vf wv[NCH][8];
    AUM_UNROLL
    for (int c = 0; c < NCH; ++c) {
        row_load8(p.weight, lane * 8 + c * 512, p.cols, wv[c]);
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) dwacc[AUM_W(w)][c][j] = splat(0.f);
    }
    for (int row = r0; row < r1; ++row) {
        const TR* xp = (const TR*)p.x + (int64_t)row * p.row_stride_x;
        const TX* gp = (const TX*)p.dy + (int64_t)row * p.row_stride_dy;
        const TR* drp = p.dresidual_out ? (const TR*)p.dresidual_out + (int64_t)row * p.row_stride_dres_out : nullptr;
        TX* dxp = (TX*)p.dx + (int64_t)row * p.row_stride_dx;
        TR* drip = p.dresidual_in ? (TR*)p.dresidual_in + (int64_t)row * p.row_stride_dres_in : nullptr;
        const float rstd = p.rstd_in[row];
        vf xh[NCH][8], wdy[NCH][8];
        vf c1 = splat(0.f);
        AUM_UNROLL
        for (int c = 0; c < NCH; ++c) {
            const vi c0 = lane * 8 + c * 512;
            vf dyv[8];
            row_load8(xp, c0, p.cols, xh[c]);
            row_load8(gp, c0, p.cols, dyv);
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) {
                xh[c][j] = xh[c][j] * rstd;
                wdy[c][j] = wv[c][j] * dyv[j];
                c1 = vfma(xh[c][j], wdy[c][j], c1);
                dwacc[AUM_W(w)][c][j] = vfma(dyv[j], xh[c][j], dwacc[AUM_W(w)][c][j]);
            }
        }
        const float cm = wave_sum(c1) / (float)p.cols;
        AUM_UNROLL
        for (int c = 0; c < NCH; ++c) {
            const vi c0 = lane * 8 + c * 512;
            vf g[8];
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) g[j] = (wdy[c][j] - xh[c][j] * cm) * rstd;
            if (drp) {
                vf dr[8];
                row_load8(drp, c0, p.cols, dr);
                AUM_UNROLL
                for (int j = 0; j < 8; ++j) g[j] = g[j] + dr[j];
            }
            row_store8(dxp, c0, p.cols, g);
            if (drip) row_store8(drip, c0, p.cols, g);
        }
    }
    // waves 1 .. NW-1 hand their sums over: [wave - 1][chunk][j][lane] (a lane's 8 values 64 floats apart: conflict-free)
    if (w > 0) {
        float* mine = lds + (w - 1) * (NCH * 512);
        AUM_UNROLL
        for (int c = 0; c < NCH; ++c)
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) lds_write(mine, lane + (c * 8 + j) * WAVE, dwacc[AUM_W(w)][c][j]);
    }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, NORM_BWD_NW) {
        if (w == 0) {
            AUM_UNROLL
            for (int c = 0; c < NCH; ++c)
                AUM_UNROLL
                for (int j = 0; j < 8; ++j) {
                    vf acc = dwacc[AUM_W(0)][c][j];
                    for (int o = 0; o < NORM_BWD_NW - 1; ++o) acc = acc + lds_read(lds + o * (NCH * 512), lane + (c * 8 + j) * WAVE);
                    dwacc[AUM_W(0)][c][j] = acc;
                }
            float* out = p.dweight_partial + (int64_t)wg * p.cols;
            AUM_UNROLL
            for (int c = 0; c < NCH; ++c) row_store8(out, lane * 8 + c * 512, p.cols, dwacc[AUM_W(0)][c]);
        }
    }
}

// float4-style streaming copy used to measure the achievable HBM bandwidth on the box.
#if !defined(AUM_EMU) && (!defined(AUM_API_PART) || AUM_API_PART == 3 || AUM_API_PART == 0)
// One 16-byte element per thread, workgroups dispatched in address order: 6.2 TB/s on 1-2 GiB buffers (the guide's 6.29 TB/s float4
// copy).  The grid-stride forms of the same copy (2048 x 256 threads walking the buffer with an 8 MiB stride, 1-4 loads in flight)
// measured 4.5-5.0 TB/s on the same box, hipMemcpyDtoD 5.0-5.2: every wave of the chip then touches a different DRAM page per
// iteration instead of streaming through consecutive ones (tools/hbm_probe.hip, profiles/r02_hbm_probe.txt).
typedef float aum_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_hbm_copy(const float4* __restrict__ src4, float4* __restrict__ dst4, int64_t n4) {
    const aum_f4* __restrict__ src = reinterpret_cast<const aum_f4*>(src4);
    aum_f4* __restrict__ dst = reinterpret_cast<aum_f4*>(dst4);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

// dst[i] = sum_o src[o][i]: 256 threads = CW columns of 8 elements x RG row groups; a thread walks rows rg, rg + RG, ... with four loads in
// flight, the row groups meet in LDS in a fixed order (no atomics: the result does not depend on the launch).
template <class T, int RG> __device__ __forceinline__ void sum_rows_body(const T* __restrict__ src, float* __restrict__ dst, int64_t outer, int64_t inner, int wgx, int tr_cols) {
    constexpr int CW = 256 / RG;
    __shared__ float part[RG > 1 ? 256 * 8 : 8];
    const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int64_t col = ((int64_t)wgx * CW + c) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < inner) {
        int64_t o = rg;
        for (; o + 7 * RG < outer; o += 8 * RG) {      // eight row loads in flight
            float v[8][8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                T e[8];
                __builtin_memcpy(e, src + (o + q * RG) * inner + col, 8 * sizeof(T));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[q][j] = elem_to_f32(e[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] += ((v[0][j] + v[1][j]) + (v[2][j] + v[3][j])) + ((v[4][j] + v[5][j]) + (v[6][j] + v[7][j]));
        }
        for (; o + 3 * RG < outer; o += 4 * RG) {
            float v[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T e[8];
                __builtin_memcpy(e, src + (o + q * RG) * inner + col, 8 * sizeof(T));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[q][j] = elem_to_f32(e[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
        }
        for (; o < outer; o += RG) {
            T e[8];
            __builtin_memcpy(e, src + o * inner + col, 8 * sizeof(T));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += elem_to_f32(e[j]);
        }
    }
    if constexpr (RG > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[(rg * 8 + j) * CW + c] = acc[j];
        __syncthreads();
        if (rg == 0 && col < inner) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = acc[j];
                for (int r = 1; r < RG; ++r) s += part[(r * 8 + j) * CW + c];
                acc[j] = s;
            }
        }
    }
    if (rg == 0 && col < inner) {
        if (tr_cols > 0) {              // the summed (inner / tr_cols, tr_cols) matrix leaves transposed (a few hundred KB: scattered 4-byte stores)
            const int64_t nrows = inner / tr_cols;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[((col + j) % tr_cols) * nrows + (col + j) / tr_cols] = acc[j];
        } else {
            // (the destination may be a view into a DistributedDataParallel bucket behind an odd-sized parameter: 4-byte aligned only.  16-byte
            // stores on a 4-byte aligned type -- global_store_dwordx4 in unaligned-access mode, as the row kernels' loads)
            struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
            F4* d4 = reinterpret_cast<F4*>(dst + col);
            d4[0] = F4{acc[0], acc[1], acc[2], acc[3]};
            d4[1] = F4{acc[4], acc[5], acc[6], acc[7]};
        }
    }
}

template <class T, int RG> __global__ __launch_bounds__(256) void k_sum_rows(const T* __restrict__ src, float* __restrict__ dst, int64_t outer, int64_t inner) {
    sum_rows_body<T, RG>(src + (int64_t)blockIdx.y * outer * inner, dst + (int64_t)blockIdx.y * inner, outer, inner, (int)blockIdx.x, 0);     // blockIdx.y: batch entry
}

// several independent sums in one launch (aum_sum_rows_multi): a workgroup finds its job from the running workgroup counts and runs it with the
// row grouping aum_sum_rows would pick for that job alone (the same additions in the same order as a launch of its own)
constexpr int SUM_MAX_JOBS = 8;
struct SumJobs {
    const float* src[SUM_MAX_JOBS];
    float* dst[SUM_MAX_JOBS];
    int64_t outer[SUM_MAX_JOBS], inner[SUM_MAX_JOBS];
    int32_t tr_cols[SUM_MAX_JOBS], rg[SUM_MAX_JOBS], wg_end[SUM_MAX_JOBS];
};
__global__ __launch_bounds__(256) void k_sum_rows_multi(SumJobs js) {
    const int wg = (int)blockIdx.x;
    int q = 0;
    while (q < SUM_MAX_JOBS - 1 && wg >= js.wg_end[q]) ++q;
    const int wgx = wg - (q ? js.wg_end[q - 1] : 0);
    switch (js.rg[q]) {         // uniform over the workgroup
        case 1: sum_rows_body<float, 1>(js.src[q], js.dst[q], js.outer[q], js.inner[q], wgx, js.tr_cols[q]); break;
        case 4: sum_rows_body<float, 4>(js.src[q], js.dst[q], js.outer[q], js.inner[q], wgx, js.tr_cols[q]); break;
        case 16: sum_rows_body<float, 16>(js.src[q], js.dst[q], js.outer[q], js.inner[q], wgx, js.tr_cols[q]); break;
        default: sum_rows_body<float, 64>(js.src[q], js.dst[q], js.outer[q], js.inner[q], wgx, js.tr_cols[q]); break;
    }
}
#endif

}  // namespace aum
