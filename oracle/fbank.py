"""ORACLE (test infrastructure only) -- numpy restatement of the log-mel frontend the reference computes on CPU workers:
`torchaudio.compliance.kaldi.fbank(waveform, htk_compat=True, sample_frequency=sr, use_energy=False,
window_type='hanning', num_mel_bins=128, dither=0.0, frame_shift=10)` on the mean-removed waveform, zero-padded /
cut to `target_length` frames and normalised `(x - mean) / (2*std)` (/root/reference/src/dataloader.py:98-101, 134-147,
220-221).

PARITY UNPINNED: torchaudio 2.1.1 is a third-party dependency that is absent from /root/reference and from this image
(SURVEY.md 8c), and the reference holds no recorded fbank output.  This file restates Kaldi's published
`compute-fbank-feats` algorithm with exactly the arguments above and Kaldi/torchaudio defaults for the rest
(25 ms frames, snip_edges, remove_dc_offset, preemphasis 0.97, round window to a power of two, power spectrum,
low_freq 20 Hz, high_freq = Nyquist, triangular filters equally spaced on mel(f) = 1127 ln(1 + f/700) evaluated at the
FFT bin centres 0..N/2-1 with the Nyquist bin weighted 0, log floor = float32 epsilon).

Cross-check (tests/test_frontend.py::test_oracle_matches_transformers_kaldi_fbank): agrees to 6e-7 with the numpy Kaldi
fbank of Hugging Face `transformers.audio_utils` (ASTFeatureExtractor's torchaudio-free path) -- an independent
restatement, not the reference's torchaudio call, so the status above stands.
"""
import numpy as np

EPS = np.float32(1.1920928955078125e-07)


def mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_banks(num_bins=128, padded=512, sr=16000.0, low=20.0, high=0.0):
    """(num_bins, padded//2 + 1) triangular filter weights; last column (Nyquist) is zero."""
    nfft = padded // 2
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    bw = sr / padded
    ml, mh = mel(low), mel(high)
    delta = (mh - ml) / (num_bins + 1)
    i = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = ml + i * delta, ml + (i + 1) * delta, ml + (i + 2) * delta
    m = mel(bw * np.arange(nfft, dtype=np.float64))[None, :]
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([w, np.zeros((num_bins, 1))], axis=1)


def sparse_banks(num_bins=128, padded=512, sr=16000.0, low=20.0, high=0.0):
    """Per mel bin: first FFT bin with non-zero weight, count, and the weights (what the HIP kernel consumes)."""
    w = mel_banks(num_bins, padded, sr, low, high)
    start = np.zeros(num_bins, np.int32)
    count = np.zeros(num_bins, np.int32)
    maxc = 0
    for i in range(num_bins):
        nz = np.nonzero(w[i])[0]
        if len(nz):
            start[i], count[i] = nz[0], nz[-1] - nz[0] + 1
            maxc = max(maxc, count[i])
    wt = np.zeros((num_bins, maxc), np.float32)
    for i in range(num_bins):
        wt[i, :count[i]] = w[i, start[i]:start[i] + count[i]]
    return start, count, wt


def fbank(wave, sr=16000, num_mel_bins=128, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, prec=np.float64):
    """wave: (n,) -> (num_frames, num_mel_bins) log-mel energies (before padding / normalisation)."""
    x = np.asarray(wave, prec)
    win = int(sr * frame_length_ms * 0.001)
    shift = int(sr * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    if len(x) < win:
        return np.zeros((0, num_mel_bins), prec)
    m = 1 + (len(x) - win) // shift
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx]                                                       # (m, win)
    fr = fr - fr.mean(axis=1, keepdims=True)                          # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)            # replicate-pad the first sample
    fr = fr - preemph * prev
    k = np.arange(win, dtype=np.float64)
    hann = (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (win - 1))).astype(prec)
    fr = fr * hann
    fr = np.concatenate([fr, np.zeros((m, padded - win), prec)], axis=1)
    spec = np.abs(np.fft.rfft(fr.astype(np.float64), axis=1)) ** 2    # (m, padded/2+1) power spectrum
    e = spec @ mel_banks(num_mel_bins, padded, float(sr)).T
    return np.log(np.maximum(e, EPS)).astype(prec)


def frontend(wave, target_length=1024, norm_mean=-4.2677393, norm_std=4.5689974, **kw):
    """dataloader.py:98-101,134-147,220-221: mean removal, fbank, pad/cut to target_length with zeros, normalise."""
    x = np.asarray(wave, np.float64)
    x = x - x.mean()
    fb = fbank(x, **kw)
    out = np.zeros((target_length, fb.shape[1]), np.float64)
    n = min(target_length, fb.shape[0])
    out[:n] = fb[:n]
    return (out - norm_mean) / (2.0 * norm_std)
