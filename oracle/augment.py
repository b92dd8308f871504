"""TEST INFRASTRUCTURE (oracle/): the training-time spectrogram augmentation of the reference's AudiosetDataset as plain numpy, one clip at
a time, in the reference's order.  Parity UNPINNED: torchaudio is absent from the image, so the masking rule below is restated from
torchaudio 2.1.1's `functional.mask_along_axis` (what `transforms.FrequencyMasking` / `TimeMasking` call with p = 1.0, mask_value = 0,
iid_masks = False) and cannot be run against it here.  It is an INDEPENDENT statement all the same: it takes the raw uniform draws and an
un-normalised log-mel, and shares no code with aum/augment.py or the HIP kernel's fused store that it checks.

  DL:206-217   fbank (time, mel) -> transpose -> (1, mel, time); FrequencyMasking(freqm) masks axis 1, TimeMasking(timem) axis 2; back
               mask_along_axis(specgram, mask_param, mask_value=0, axis):
                   value = rand() * mask_param;  min_value = rand() * (size(axis) - value)
                   mask_start = long(min_value);  mask_end = long(min_value) + long(value)
                   specgram[..., mask_start:mask_end (along axis)] = mask_value
  DL:220-221   (fbank - norm_mean) / (norm_std * 2)
  DL:226-228   fbank = fbank + rand(T, F) * rand() / 10;  fbank = roll(fbank, randint(-10, 10), 0)
"""
import numpy as np


def mask_along_axis(spec, u_value, u_min, mask_param, axis, mask_value=0.0):
    """spec (1, freq, time); u_value, u_min: the two uniform [0, 1) draws in torchaudio's order"""
    size = spec.shape[axis]
    value = u_value * mask_param
    min_value = u_min * (size - value)
    start = int(np.floor(min_value))           # .long() of a non-negative number
    end = start + int(np.floor(value))
    out = spec.copy()
    if axis == 1:
        out[:, start:end, :] = mask_value
    else:
        out[:, :, start:end] = mask_value
    return out


def augment_clip(fbank, draws, freqm, timem, noise, norm_mean, norm_std):
    """fbank: (time, mel) un-normalised log-mel, zero rows where the clip was padded (DL:139-145).
    draws: dict with the uniform draws u_fv, u_fm (frequency band), u_tv, u_tm (time band), and for noise: u_amp (scalar in [0, 1)),
    field (time, mel) uniform [0, 1), shift (integer in [-10, 10))."""
    x = np.asarray(fbank, np.float64).T[None]                      # DL:208-210
    if freqm != 0:
        x = mask_along_axis(x, draws["u_fv"], draws["u_fm"], freqm, 1)
    if timem != 0:
        x = mask_along_axis(x, draws["u_tv"], draws["u_tm"], timem, 2)
    x = x[0].T                                                     # DL:216-217
    x = (x - norm_mean) / (norm_std * 2)                           # DL:221
    if noise:
        x = x + np.asarray(draws["field"], np.float64) * draws["u_amp"] / 10      # DL:227
        x = np.roll(x, int(draws["shift"]), 0)                     # DL:228
    return x
