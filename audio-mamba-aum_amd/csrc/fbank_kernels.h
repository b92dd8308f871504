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

// lds: re[padded], im[padded], pw[padded/2+1], partial[4]
AUM_DEV void fbank_frame(const AumFbankArgs& p, int wg, float* lds) {
    const int b = wg / p.target_length, frame = wg % p.target_length;
    float* re = lds;
    float* im = lds + FBANK_MAX_FFT;
    float* pw = lds + 2 * FBANK_MAX_FFT;
    float* part = pw + FBANK_MAX_FFT / 2 + 1;
    const float pad_value = (0.f - p.norm_mean) * p.norm_inv2std;
    // per-clip augmentation table (ABI 4): ragged frame count, SpecAug bands, noise amplitude, roll -- all applied in the store
    int n_frames = p.num_frames, f_lo = 0, f_hi = 0, t_lo = 0, t_hi = 0, roll = 0;
    float amp = 0.f;
    if (p.aug) {
        const float* a = p.aug + (int64_t)b * AUM_FBANK_AUG;
        if (a[0] >= 0.f) n_frames = (int)a[0] < p.num_frames ? (int)a[0] : p.num_frames;
        f_lo = (int)a[1]; f_hi = (int)a[2]; t_lo = (int)a[3]; t_hi = (int)a[4];
        roll = (int)a[5];
        amp = a[6];
    }
    int frame_out = (frame + roll) % p.target_length;
    if (frame_out < 0) frame_out += p.target_length;
    float* out = p.out + (int64_t)b * p.out_bs + (int64_t)frame_out * p.num_mel;
    const float* nz = (p.noise && amp != 0.f) ? p.noise + ((int64_t)b * p.target_length + frame) * p.num_mel : nullptr;
    const bool t_masked = frame >= t_lo && frame < t_hi;
    if (frame >= n_frames || t_masked) {    // zero padding of the reference (DL:139-145) / SpecAug time band: 0 before normalisation
        AUM_FOR_EACH_WAVE(w, FBANK_NW) {
            for (int m0 = w * WAVE; m0 < p.num_mel; m0 += FBANK_THREADS) {
                const vi m = lane_id() + m0;
                const vm ok = m < p.num_mel;
                vf v = splat(pad_value);
                if (nz) v = vfma(gload(nz, vmin_i(m, p.num_mel - 1), ok), splat(amp), v);
                gstore(out, m, v, ok);
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
            gstore(out, m, o, ok);
        }
    }
}

}  // namespace aum
