"""Waveform -> normalised log-mel spectrogram on the GPU (aum_fbank_fwd), replacing the reference's CPU DataLoader
chain `torchaudio.load -> waveform - mean -> kaldi.fbank -> pad/cut -> (x - mean) / (2 std)`
(/root/reference/src/dataloader.py:98-101, 134-147, 220-221).  The filterbank tables are built here on the host,
once per (sample rate, window, mel bins), with the Kaldi defaults the reference relies on."""
import math

import numpy as np
import torch

import aum_hip

AUDIOSET_MEAN, AUDIOSET_STD = -4.2677393, 4.5689974      # norm_mean / norm_std of the reference's AudioSet scripts


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


class FbankTables:
    def __init__(self, device, sample_rate=16000, num_mel_bins=128, frame_length_ms=25.0, frame_shift_ms=10.0,
                 low_freq=20.0, high_freq=0.0):
        win = int(sample_rate * frame_length_ms * 0.001)
        shift = int(sample_rate * frame_shift_ms * 0.001)
        padded = 1 << (win - 1).bit_length()
        k = np.arange(win, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * math.pi * k / (win - 1))                 # hann, symmetric
        tk = np.arange(padded // 2, dtype=np.float64)
        tw = np.stack([np.cos(2.0 * math.pi * tk / padded), -np.sin(2.0 * math.pi * tk / padded)], axis=1)
        # triangular filters on the mel scale, evaluated at the centres of FFT bins 0 .. padded/2-1
        nyq = 0.5 * sample_rate
        hi = high_freq + nyq if high_freq <= 0 else high_freq
        ml, mh = _mel(low_freq), _mel(hi)
        delta = (mh - ml) / (num_mel_bins + 1)
        fm = _mel(sample_rate / padded * np.arange(padded // 2, dtype=np.float64))
        start = np.zeros(num_mel_bins, np.float32)
        count = np.zeros(num_mel_bins, np.float32)
        rows = []
        for i in range(num_mel_bins):
            left, center, right = ml + i * delta, ml + (i + 1) * delta, ml + (i + 2) * delta
            w = np.maximum(0.0, np.minimum((fm - left) / (center - left), (right - fm) / (right - center)))
            nz = np.nonzero(w)[0]
            if len(nz):
                start[i], count[i] = nz[0], nz[-1] - nz[0] + 1
                rows.append(w[nz[0]:nz[-1] + 1])
            else:
                rows.append(np.zeros(0))
        stride = max(1, int(count.max()))
        mel_w = np.zeros((num_mel_bins, stride), np.float32)
        for i, r in enumerate(rows):
            mel_w[i, :len(r)] = r
        t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)
        self.tables = dict(window=t(window), twiddle=t(tw), mel_start_f=t(start), mel_count_f=t(count), mel_w=t(mel_w),
                           win=win, shift=shift, padded=padded)


class WaveInput:
    """What AudioMamba.forward(wave, frontend=...) needs besides the waveform to run waveform -> tokens in one launch
    (aum_frontend_tokens_fwd): the filterbank tables, the normalisation and this batch's augmentation draw."""

    def __init__(self, tables, target_length=1024, norm_mean=AUDIOSET_MEAN, norm_std=AUDIOSET_STD, aug=None, noise=None):
        self.tables, self.target_length = tables, target_length
        self.norm_mean, self.norm_std, self.aug, self.noise = norm_mean, norm_std, aug, noise

    def spectrogram(self, wave):
        """the same clips through the stand-alone log-mel kernel (configurations the one-launch path does not cover)"""
        return aum_hip.fbank_fwd(wave, self.tables.tables, self.target_length, self.norm_mean, self.norm_std, aug=self.aug,
                                 noise=self.noise)


def prepare_wave(wave, n_valid, tables, aug=None):
    """Mean removal (dataloader.py:101) and, for ragged batches, the per-clip frame counts in column 0 of the augmentation
    table: the host-side half of wav2fbank / wav2fbank_ragged, shared with the one-launch path."""
    wave = (wave - wave.mean(dim=1, keepdim=True)).contiguous()
    if n_valid is not None:
        win, shift = tables.tables["win"], tables.tables["shift"]
        n_valid = torch.as_tensor(n_valid, device=wave.device)
        frames = torch.where(n_valid >= win, 1 + (n_valid - win) // shift, torch.zeros_like(n_valid))
        if aug is None:
            aug = torch.zeros((wave.shape[0], aum_hip.FBANK_AUG), dtype=torch.float32, device=wave.device)
        aug[:, 0] = frames.to(torch.float32)
    return wave, aug


def wav2fbank(wave, tables, target_length=1024, norm_mean=AUDIOSET_MEAN, norm_std=AUDIOSET_STD, aug=None, noise=None):
    """wave: (batch, n_samples) fp32 on the GPU -> (batch, target_length, num_mel) normalised log-mel, what
    AudiosetDataset.__getitem__ returns per clip (without mixup).  aug / noise: the per-clip augmentation table and noise
    field of aum.augment.draw_augmentation, applied inside the kernel's store."""
    wave = wave - wave.mean(dim=1, keepdim=True)                # dataloader.py:101
    return aum_hip.fbank_fwd(wave.contiguous(), tables.tables, target_length, norm_mean, norm_std, aug=aug, noise=noise)


def pad_fill(norm_mean=AUDIOSET_MEAN, norm_std=AUDIOSET_STD):
    """what a zero-padded (or SpecAug-masked) fbank frame becomes after (x - mean) / (2 std) (dataloader.py:141-144, 220)"""
    return (0.0 - norm_mean) / (2.0 * norm_std)


def wav2fbank_ragged(wave, n_valid, tables, target_length=1024, norm_mean=AUDIOSET_MEAN, norm_std=AUDIOSET_STD, aug=None,
                     noise=None):
    """Clips of different lengths in one launch: wave (batch, max_samples) zero-padded, n_valid (batch,) samples.
    Kaldi's snip_edges framing only emits frames that lie inside the clip, so frames past 1 + (n - win)//shift are
    the reference's ZeroPad2d rows (dataloader.py:139-145) -- written by the kernel itself (column 0 of the per-clip table).
    aug / noise: SpecAug bands, noise and roll of aum.augment.draw_augmentation, applied in the same store."""
    wave, aug = prepare_wave(wave, n_valid, tables, aug)
    return aum_hip.fbank_fwd(wave, tables.tables, target_length, norm_mean, norm_std, aug=aug, noise=noise)
