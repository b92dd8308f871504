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


def pack_augmentation(n_time, n_mel, freqm, timem, u_fv=None, u_fm=None, u_tv=None, u_tm=None, u_amp=None, field=None, shift=None):
    """The log-mel kernel's per-clip augmentation table from the raw draws (each a (batch,) tensor of uniform [0, 1) numbers; `shift`
    integers in [-10, 10); `field` the (batch, n_time, n_mel) uniform noise field): the band limits by torchaudio's mask_along_axis rule
    (DL:206-217), the noise amplitude u / 10 and the roll (DL:226-228).  Column 0 -- the frame count -- is left at -1."""
    ref = next(t for t in (u_fv, u_tv, u_amp) if t is not None)
    aug = torch.zeros((ref.shape[0], 8), dtype=torch.float32, device=ref.device)
    aug[:, 0] = -1.0

    def band(size, param, u_v, u_m):
        value = u_v * param
        lo = (u_m * (size - value)).long()
        return lo.float(), (lo + value.long()).float()
    if freqm:
        aug[:, 1], aug[:, 2] = band(n_mel, freqm, u_fv, u_fm)
    if timem:
        aug[:, 3], aug[:, 4] = band(n_time, timem, u_tv, u_tm)
    if u_amp is not None:
        aug[:, 6] = u_amp / 10
        aug[:, 5] = shift.float()
    return aug, field


def draw_augmentation(batch, n_time, n_mel, freqm, timem, noise, device, generator=None):
    """The random numbers of spec_augment + noise_roll for one batch, in THEIR draw order (so a seeded generator gives the same
    augmentation either way), packed for the log-mel kernel's fused epilogue (aum_hip.fbank_fwd aug= / noise=):
    returns (aug (batch, 8) fp32 with column 0 -- the frame count -- left at -1, noise (batch, n_time, n_mel) or None)."""
    u = lambda: torch.rand(batch, device=device, generator=generator)
    kw = {}
    if freqm:
        kw["u_fv"], kw["u_fm"] = u(), u()
    if timem:
        kw["u_tv"], kw["u_tm"] = u(), u()
    if noise:
        kw["u_amp"] = torch.rand(batch, 1, 1, device=device, generator=generator).flatten()
        kw["field"] = torch.rand((batch, n_time, n_mel), device=device, generator=generator)
        kw["shift"] = torch.randint(-10, 10, (batch,), device=device, generator=generator)
    if not kw:
        aug = torch.zeros((batch, 8), dtype=torch.float32, device=device)
        aug[:, 0] = -1.0
        return aug, None
    return pack_augmentation(n_time, n_mel, freqm, timem, **kw)


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
