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

// float4-style streaming copy used to measure the achievable HBM bandwidth on the box.
#if !defined(AUM_EMU) && (!defined(AUM_API_PART) || AUM_API_PART == 3 || AUM_API_PART == 0)
__global__ void k_hbm_copy(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = src[i];
}
#endif

}  // namespace aum
