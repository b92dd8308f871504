"""Training-time augmentation of the reference's AudiosetDataset (/root/reference/src/dataloader.py = "DL"), restated
as batched device ops that run AFTER the log-mel kernel instead of per clip inside CPU DataLoader workers:

  SpecAug  DL:206-217  torchaudio FrequencyMasking / TimeMasking: one band per clip, width ~ U[0, param),
                       start ~ U[0, size - width), filled with 0 BEFORE normalisation
  noise    DL:226-228  fbank += U[0,1) * U[0,1)/10 ; roll along time by randint(-10, 10)
  mixup    DL:104-129  lambda ~ Beta(10, 10) on mean-removed waveforms, re-centred; labels mixed DL:178-184

The kernel normalises in the same pass, so a pre-normalisation 0 is `fill = (0 - mean) / (2 std)` here.
"""
import numpy as np
import torch


def _band_mask(batch, size, param, device, generator):
    """torchaudio.functional.mask_along_axis semantics: value = rand * param, start = rand * (size - value),
    mask = [floor(start), floor(start) + floor(value))."""
    value = torch.rand(batch, device=device, generator=generator) * param
    start = torch.rand(batch, device=device, generator=generator) * (size - value)
    lo = start.long()
    hi = lo + value.long()
    idx = torch.arange(size, device=device)
    return (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])


def spec_augment(fbank, freqm, timem, fill=0.0, generator=None):
    """fbank: (batch, time, mel) normalised log-mel.  Returns a masked copy."""
    Bsz, T, Fd = fbank.shape
    if freqm:
        fbank = fbank.masked_fill(_band_mask(Bsz, Fd, freqm, fbank.device, generator)[:, None, :], fill)
    if timem:
        fbank = fbank.masked_fill(_band_mask(Bsz, T, timem, fbank.device, generator)[:, :, None], fill)
    return fbank


def noise_roll(fbank, generator=None):
    Bsz, T, Fd = fbank.shape
    dev = fbank.device
    amp = torch.rand(Bsz, 1, 1, device=dev, generator=generator) / 10
    fbank = fbank + torch.rand(fbank.shape, device=dev, generator=generator) * amp
    shift = torch.randint(-10, 10, (Bsz,), device=dev, generator=generator)
    idx = (torch.arange(T, device=dev)[None, :] - shift[:, None]) % T          # torch.roll(x, s)[t] = x[t - s]
    return torch.gather(fbank, 1, idx[:, :, None].expand(-1, -1, Fd))


def mixup_waveforms(w1, w2, rng=np.random):
    """numpy 1-D waveforms (already mean-removed).  Pads/cuts w2 to len(w1) (DL:115-123)."""
    if len(w2) < len(w1):
        w2 = np.concatenate([w2, np.zeros(len(w1) - len(w2), w2.dtype)])
    else:
        w2 = w2[:len(w1)]
    lam = float(rng.beta(10, 10))
    mix = lam * w1 + (1.0 - lam) * w2
    return (mix - mix.mean()).astype(np.float32), lam
