// fbank_kernels.h -- Kaldi-style log-mel filterbank frontend on the GPU: one 256-thread workgroup per frame.
//
// Replaces the CPU DataLoader chain of the reference (torchaudio.compliance.kaldi.fbank + pad + normalise,
// src/dataloader.py:134-147, 220-221; arguments at dataloader.py:134-135): per frame DC removal, pre-emphasis,
// Hann window, zero-pad to a power of two, radix-2 FFT in LDS, power spectrum, sparse triangular mel filters, log,
// (x - mean) / (2 std).  Frames past the end of the clip are the reference's zero padding (normalised).
// The tables (window, twiddles, sparse filterbank) come from the host (aum/frontend.py).
#pragma once
#include "../../include/aum_hip.h"
#include "wave.h"

namespace aum {

constexpr int FBANK_NW = 4;                 // waves per workgroup
constexpr int FBANK_THREADS = FBANK_NW * WAVE;
constexpr int FBANK_MAX_FFT = 1024;
constexpr int FBANK_LDS_FLOATS = 2 * FBANK_MAX_FFT + FBANK_MAX_FFT / 2 + 1 + 8;

AUM_DEV vi bit_reverse(vi x, int bits) {
    vi r = spl_i(0);
    for (int i = 0; i < bits; ++i) r = r + (((x >> i) & 1) * (1 << (bits - 1 - i)));
    return r;
}

// per-clip augmentation table (ABI 4) decoded: ragged frame count, SpecAug bands, noise amplitude, roll
struct FbankAug { int n_frames, f_lo, f_hi, t_lo, t_hi, roll; float amp; };
AUM_DEV FbankAug fbank_aug(const AumFbankArgs& p, int b) {
    FbankAug a{p.num_frames, 0, 0, 0, 0, 0, 0.f};
    if (p.aug) {
        const float* t = p.aug + (int64_t)b * AUM_FBANK_AUG;
        if (t[0] >= 0.f) a.n_frames = (int)t[0] < p.num_frames ? (int)t[0] : p.num_frames;
        a.f_lo = (int)t[1]; a.f_hi = (int)t[2]; a.t_lo = (int)t[3]; a.t_hi = (int)t[4];
        a.roll = (int)t[5];
        a.amp = t[6];
    }
    return a;
}

// One frame: `frame` = index into the clip (the frame the arithmetic runs on).  emit(m, value, ok) receives the finished log-mel
// values -- normalised, SpecAug-masked, noise added -- of mel bins m (one call per 64-bin group and wave).
// lds: re[padded], im[padded], pw[padded/2+1], partial[4]
template <class EMIT>
AUM_DEV void fbank_frame_core(const AumFbankArgs& p, int b, int frame, const FbankAug& ag, float* lds, EMIT emit) {
    float* re = lds;
    float* im = lds + FBANK_MAX_FFT;
    float* pw = lds + 2 * FBANK_MAX_FFT;
    float* part = pw + FBANK_MAX_FFT / 2 + 1;
    const float pad_value = (0.f - p.norm_mean) * p.norm_inv2std;
    const int f_lo = ag.f_lo, f_hi = ag.f_hi;
    const float amp = ag.amp;
    const float* nz = (p.noise && amp != 0.f) ? p.noise + ((int64_t)b * p.target_length + frame) * p.num_mel : nullptr;
    const bool t_masked = frame >= ag.t_lo && frame < ag.t_hi;
    if (frame >= ag.n_frames || t_masked) {    // zero padding of the reference (DL:139-145) / SpecAug time band: 0 before normalisation
        AUM_FOR_EACH_WAVE(w, FBANK_NW) {
            for (int m0 = w * WAVE; m0 < p.num_mel; m0 += FBANK_THREADS) {
                const vi m = lane_id() + m0;
                const vm ok = m < p.num_mel;
                vf v = splat(pad_value);
                if (nz) v = vfma(gload(nz, vmin_i(m, p.num_mel - 1), ok), splat(amp), v);
                emit(m, v, ok);
            }
        }
        return;
    }
    const float* x = p.wave + (int64_t)b * p.wave_bs + (int64_t)frame * p.shift;
    int bits = 0;
    while ((1 << bits) < p.padded) ++bits;
    // ---- frame mean
    AUM_FOR_EACH_WAVE(w, FBANK_NW) {
        vf s = splat(0.f);
        for (int i0 = w * WAVE; i0 < p.win; i0 += FBANK_THREADS) {
            const vi i = lane_id() + i0;
            s = s + gload(x, i, i < p.win);
        }
        lds_write(part, spl_i(w), splat(wave_sum(s)));
    }
    AUM_WG_BARRIER();
    // ---- DC removal, pre-emphasis, window, bit-reversed store
    AUM_FOR_EACH_WAVE(w, FBANK_NW) {
        float tot = 0.f;
        for (int q = 0; q < FBANK_NW; ++q) tot += readlane(lds_read(part, spl_i(q)), 0);
        const float mean = tot / (float)p.win;
        for (int i0 = w * WAVE; i0 < p.padded; i0 += FBANK_THREADS) {
            const vi i = lane_id() + i0;
            const vm in = i < p.win;
            const vf cur = gload(x, i, in) - mean;
            const vf prv = gload(x, vmax_i(i - 1, 0), in) - mean;          // i = 0: replicate (x[0] - c*x[0])
            vf v = (cur - p.preemph * prv) * gload(p.window, i, in);
            v = vsel(in, v, splat(0.f));
            const vi j = bit_reverse(i, bits);
            lds_write(re, j, v);
            lds_write(im, j, splat(0.f));
        }
    }
    AUM_WG_BARRIER();
    // ---- radix-2 decimation-in-time FFT, one butterfly per thread and pass
    for (int s = 0; s < bits; ++s) {
        const int half = 1 << s;
        AUM_FOR_EACH_WAVE(w, FBANK_NW) {
            for (int j0 = w * WAVE; j0 < p.padded / 2; j0 += FBANK_THREADS) {
                const vi j = lane_id() + j0;
                const vi grp = j >> s;
                const vi pos = j - (grp << s);           // j % half
                const vi i0 = (grp << (s + 1)) + pos;
                const vi i1 = i0 + half;
                const vi tk = pos * (p.padded / 2 / half);      // twiddle index: exp(-2 pi i tk / padded)
                const vf wr = gload(p.twiddle, tk * 2, j >= 0), wi = gload(p.twiddle, tk * 2 + 1, j >= 0);
                const vf ar = lds_read(re, i0), ai = lds_read(im, i0);
                const vf br = lds_read(re, i1), bi = lds_read(im, i1);
                const vf tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                lds_write(re, i0, ar + tr);
                lds_write(im, i0, ai + ti);
                lds_write(re, i1, ar - tr);
                lds_write(im, i1, ai - ti);
            }
        }
        AUM_WG_BARRIER();
    }
    // ---- power spectrum, bins 0 .. padded/2
    AUM_FOR_EACH_WAVE(w, FBANK_NW) {
        for (int k0 = w * WAVE; k0 < p.padded / 2 + 1; k0 += FBANK_THREADS) {
            const vi k = vmin_i(lane_id() + k0, p.padded / 2);
            const vf r = lds_read(re, k), q = lds_read(im, k);
            lds_write(pw, k, vfma(r, r, q * q));
        }
    }
    AUM_WG_BARRIER();
    // ---- sparse triangular filters, log, normalise
    AUM_FOR_EACH_WAVE(w, FBANK_NW) {
        for (int m0 = w * WAVE; m0 < p.num_mel; m0 += FBANK_THREADS) {
            const vi m = lane_id() + m0;
            const vm ok = m < p.num_mel;
            const vi mc = vmin_i(m, p.num_mel - 1);
            const vf startf = gload(p.mel_start_f, mc, ok), countf = gload(p.mel_count_f, mc, ok);
            vf e = splat(0.f);
            for (int c = 0; c < p.mel_wstride; ++c) {
                const vm use = ok && (countf > (float)c);
                // start/count travel as floats (exact small integers) so the kernel needs no integer load primitive
                const vi bin = vmin_i(vcvt_i(startf) + c, p.padded / 2);
                e = vfma(gload(p.mel_w, mc * p.mel_wstride + c, use), vsel(use, lds_read(pw, bin), splat(0.f)), e);
            }
            const vf v = vlog2(vmax(e, splat(p.log_floor))) * LN2;
            vf o = (v - p.norm_mean) * p.norm_inv2std;
            o = vsel((m >= f_lo) && (m < f_hi), splat(pad_value), o);          // SpecAug frequency band
            if (nz) o = vfma(gload(nz, mc, ok), splat(amp), o);
            emit(m, o, ok);
        }
    }
}

// =================================================================================================
// Wave-per-frame path (padded == 512, the 16 kHz / 25 ms configuration every recipe of the reference uses): one wavefront owns
// a frame end to end, so no workgroup barrier and no per-pass twiddle fetch sits on the critical path.  512 = 8 x 8 x 8:
// sample n = 64 n2 + 8 n1 + n0, bin k = k0 + 8 k1 + 64 k2,
//   X[k] = sum_n0 W8^(n0 k2) W64^(n0 k1) [ sum_n1 W8^(n1 k1) W512^((8 n1 + n0) k0) [ sum_n2 W8^(n2 k0) x[n] ] ]
// -> three 8-point DFTs in registers (lane = the two indices not summed over) with two transposes through the wave's
// private LDS strip; the 14 + 14 twiddle registers are loaded once per wave and reused for every frame it processes.
// LDS strip per wave: re[8][72] | im[8][72] (pitch 72 and the n0*9 skew keep all four transposes bank-conflict free);
// the power spectrum (257 floats) reuses the re half.
// =================================================================================================
constexpr int FBW_N = 512;
constexpr int FBW_PITCH = 72;
constexpr int FBW_XCH = 8 * FBW_PITCH;
constexpr int FBW_WAVE_FLOATS = 2 * FBW_XCH;

struct FbwTw { vf c1[7], s1[7], c2[7], s2[7]; };

AUM_DEV void fbw_load_tw(const AumFbankArgs& p, vi tk, vf& c, vf& s) {      // W512^tk from the half-circle table of the host
    tk = tk & (FBW_N - 1);
    const vm neg = tk >= FBW_N / 2;
    const vi t2 = tk & (FBW_N / 2 - 1);
    const vf cc = gload(p.twiddle, t2 * 2, t2 >= 0), ss = gload(p.twiddle, t2 * 2 + 1, t2 >= 0);
    c = vsel(neg, splat(0.f) - cc, cc);
    s = vsel(neg, splat(0.f) - ss, ss);
}
AUM_DEV void fbw_twiddles(const AumFbankArgs& p, FbwTw& t) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int k = 1; k < 8; ++k) {
        fbw_load_tw(p, lane * k, t.c1[k - 1], t.s1[k - 1]);                 // W512^((8 n1 + n0) k0), lane = 8 n1 + n0
        fbw_load_tw(p, (lane & 7) * (8 * k), t.c2[k - 1], t.s2[k - 1]);     // W64^(n0 k1), n0 = lane & 7
    }
}

// 8-point forward DFT in registers (decimation in frequency), result in natural order
AUM_DEV void dft8(vf (&re)[8], vf (&im)[8]) {
    constexpr float H = 0.70710678118654752f;
    vf br[8], bi[8], cr[8], ci[8];
    AUM_UNROLL
    for (int j = 0; j < 4; ++j) { br[j] = re[j] + re[j + 4]; bi[j] = im[j] + im[j + 4]; }
    {
        const vf d0r = re[0] - re[4], d0i = im[0] - im[4], d1r = re[1] - re[5], d1i = im[1] - im[5];
        const vf d2r = re[2] - re[6], d2i = im[2] - im[6], d3r = re[3] - re[7], d3i = im[3] - im[7];
        br[4] = d0r;                 bi[4] = d0i;                              // * W8^0
        br[5] = (d1r + d1i) * H;     bi[5] = (d1i - d1r) * H;                  // * W8^1 = (1 - i) / sqrt 2
        br[6] = d2i;                 bi[6] = splat(0.f) - d2r;                 // * W8^2 = -i
        br[7] = (d3i - d3r) * H;     bi[7] = splat(0.f) - (d3r + d3i) * H;     // * W8^3 = (-1 - i) / sqrt 2
    }
    AUM_UNROLL
    for (int h = 0; h < 8; h += 4) {
        cr[h] = br[h] + br[h + 2];         ci[h] = bi[h] + bi[h + 2];
        cr[h + 1] = br[h + 1] + br[h + 3]; ci[h + 1] = bi[h + 1] + bi[h + 3];
        cr[h + 2] = br[h] - br[h + 2];     ci[h + 2] = bi[h] - bi[h + 2];
        cr[h + 3] = bi[h + 1] - bi[h + 3]; ci[h + 3] = br[h + 3] - br[h + 1];  // (b1 - b3) * -i
    }
    constexpr int ORD[8] = {0, 4, 2, 6, 1, 5, 3, 7};          // d[q] -> X[ORD[q]]
    AUM_UNROLL
    for (int q = 0; q < 8; q += 2) {
        re[ORD[q]] = cr[q] + cr[q + 1];     im[ORD[q]] = ci[q] + ci[q + 1];
        re[ORD[q + 1]] = cr[q] - cr[q + 1]; im[ORD[q + 1]] = ci[q] - ci[q + 1];
    }
}

// one frame on one wave; wl = this wave's FBW_WAVE_FLOATS strip
template <class EMIT>
AUM_DEV void fbank_frame_wave(const AumFbankArgs& p, int b, int frame, const FbankAug& ag, const FbwTw& tw, float* wl, EMIT emit) {
    const float pad_value = (0.f - p.norm_mean) * p.norm_inv2std;
    const float amp = ag.amp;
    const float* nz = (p.noise && amp != 0.f) ? p.noise + ((int64_t)b * p.target_length + frame) * p.num_mel : nullptr;
    const vi lane = lane_id();
    if (frame >= ag.n_frames || (frame >= ag.t_lo && frame < ag.t_hi)) {      // zero padding (DL:139-145) / SpecAug time band
        for (int m0 = 0; m0 < p.num_mel; m0 += WAVE) {
            const vi m = lane + m0;
            const vm ok = m < p.num_mel;
            vf v = splat(pad_value);
            if (nz) v = vfma(gload(nz, vmin_i(m, p.num_mel - 1), ok), splat(amp), v);
            emit(m, v, ok);
        }
        return;
    }
    const float* x = p.wave + (int64_t)b * p.wave_bs + (int64_t)frame * p.shift;
    float* xr = wl;
    float* xi = wl + FBW_XCH;
    vf re[8], im[8];
    // ---- frame mean, DC removal, pre-emphasis, window: lane = 8 n1 + n0 holds samples 64 n2 + lane
    vf s = splat(0.f);
    AUM_UNROLL
    for (int r = 0; r < 8; ++r) {
        const vi i = lane + 64 * r;
        re[r] = gload(x, i, i < p.win);
        s = s + re[r];
    }
    const float mean = wave_sum(s) / (float)p.win;
    AUM_UNROLL
    for (int r = 0; r < 8; ++r) {
        const vi i = lane + 64 * r;
        const vm in = i < p.win;
        const vf cur = re[r] - mean;
        const vf prv = gload(x, vmax_i(i - 1, 0), in) - mean;                 // i = 0: replicate (x[0] - c*x[0])
        const vf v = (cur - p.preemph * prv) * gload(p.window, i, in);
        re[r] = vsel(in, v, splat(0.f));
        im[r] = splat(0.f);
    }
    // ---- pass 1: DFT over n2 -> k0, twiddle W512^(lane k0), transpose to lane = 8 k0 + n0
    dft8(re, im);
    lds_write(xr, lane, re[0]);
    lds_write(xi, lane, im[0]);
    AUM_UNROLL
    for (int k = 1; k < 8; ++k) {
        const vf c = tw.c1[k - 1], sn = tw.s1[k - 1];
        lds_write(xr, lane + k * FBW_PITCH, re[k] * c - im[k] * sn);
        lds_write(xi, lane + k * FBW_PITCH, re[k] * sn + im[k] * c);
    }
    const vi hi = lane >> 3, lo = lane & 7;
    AUM_UNROLL
    for (int n1 = 0; n1 < 8; ++n1) {
        const vi a = hi * FBW_PITCH + lo + n1 * 8;
        re[n1] = lds_read(xr, a);
        im[n1] = lds_read(xi, a);
    }
    // ---- pass 2: DFT over n1 -> k1, twiddle W64^(n0 k1), transpose to lane = 8 k0 + k1
    dft8(re, im);
    {
        const vi a = hi * FBW_PITCH + lo * 9;
        lds_write(xr, a, re[0]);
        lds_write(xi, a, im[0]);
        AUM_UNROLL
        for (int k = 1; k < 8; ++k) {
            const vf c = tw.c2[k - 1], sn = tw.s2[k - 1];
            lds_write(xr, a + k, re[k] * c - im[k] * sn);
            lds_write(xi, a + k, re[k] * sn + im[k] * c);
        }
    }
    AUM_UNROLL
    for (int n0 = 0; n0 < 8; ++n0) {
        const vi a = hi * FBW_PITCH + lo + n0 * 9;
        re[n0] = lds_read(xr, a);
        im[n0] = lds_read(xi, a);
    }
    // ---- pass 3: DFT over n0 -> k2; power spectrum of bins k0 + 8 k1 + 64 k2 <= 256 into the re half of the strip
    dft8(re, im);
    float* pw = wl;
    const vi bin0 = hi + lo * 8;
    AUM_UNROLL
    for (int k2 = 0; k2 < 4; ++k2) lds_write(pw, bin0 + 64 * k2, vfma(re[k2], re[k2], im[k2] * im[k2]));
    lds_write_m(pw, bin0 + 256, vfma(re[4], re[4], im[4] * im[4]), lane == 0);
    // ---- sparse triangular filters, log, normalise
    for (int m0 = 0; m0 < p.num_mel; m0 += WAVE) {
        const vi m = lane + m0;
        const vm ok = m < p.num_mel;
        const vi mc = vmin_i(m, p.num_mel - 1);
        const vf startf = gload(p.mel_start_f, mc, ok), countf = gload(p.mel_count_f, mc, ok);
        const vi start = vcvt_i(startf);
        vf e = splat(0.f);
        for (int c = 0; c < p.mel_wstride; ++c) {
            const vm use = ok && (countf > (float)c);
            if (!any_lane(use)) break;
            const vi bin = vmin_i(start + c, FBW_N / 2);
            e = vfma(gload(p.mel_w, mc * p.mel_wstride + c, use), vsel(use, lds_read(pw, bin), splat(0.f)), e);
        }
        const vf v = vlog2(vmax(e, splat(p.log_floor))) * LN2;
        vf o = (v - p.norm_mean) * p.norm_inv2std;
        o = vsel((m >= ag.f_lo) && (m < ag.f_hi), splat(pad_value), o);          // SpecAug frequency band
        if (nz) o = vfma(gload(nz, mc, ok), splat(amp), o);
        emit(m, o, ok);
    }
}

// stand-alone log-mel kernel, wave-per-frame form: FBANK_NW waves x FBW_FPW consecutive frames each
constexpr int FBW_FPW = 4;
constexpr int FBW_WG_FRAMES = FBANK_NW * FBW_FPW;
AUM_DEV void fbank_frames_wg(const AumFbankArgs& p, int wg, float* lds) {
    const int total = p.batch * p.target_length;
    AUM_FOR_EACH_WAVE(w, FBANK_NW) {
        FbwTw tw;
        fbw_twiddles(p, tw);
        for (int q = 0; q < FBW_FPW; ++q) {
            const int g = wg * FBW_WG_FRAMES + w * FBW_FPW + q;
            if (g >= total) break;
            const int b = g / p.target_length, frame = g % p.target_length;
            const FbankAug ag = fbank_aug(p, b);
            int frame_out = (frame + ag.roll) % p.target_length;
            if (frame_out < 0) frame_out += p.target_length;
            float* out = p.out + (int64_t)b * p.out_bs + (int64_t)frame_out * p.num_mel;
            fbank_frame_wave(p, b, frame, ag, tw, lds + w * FBW_WAVE_FLOATS, [&](vi m, vf v, vm ok) { gstore(out, m, v, ok); });
        }
    }
}

// the stand-alone log-mel kernel: one workgroup per frame, output row = (frame + roll) mod target_length
AUM_DEV void fbank_frame(const AumFbankArgs& p, int wg, float* lds) {
    const int b = wg / p.target_length, frame = wg % p.target_length;
    const FbankAug ag = fbank_aug(p, b);
    int frame_out = (frame + ag.roll) % p.target_length;
    if (frame_out < 0) frame_out += p.target_length;
    float* out = p.out + (int64_t)b * p.out_bs + (int64_t)frame_out * p.num_mel;
    fbank_frame_core(p, b, frame, ag, lds, [&](vi m, vf v, vm ok) { gstore(out, m, v, ok); });
}

}  // namespace aum
