"""Log-mel frontend kernel (aum_fbank_fwd) vs the numpy oracle (oracle/fbank.py; parity UNPINNED: torchaudio is absent,
see the oracle's header).  CPU: kernel source on the lane-array build.  GPU (-m gpu): the real library."""
import os
import sys

import numpy as np
import pytest
import torch

import aum_hip
from oracle import fbank as OF

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


def _wave(n, seed):
    r = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    return (0.1 * r.normal(0, 1, n) + 0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 3000 * t * (1 + t))).astype(np.float32)


def _check(lib, dev, n_samples, target, batch=2, sr=16000):
    from aum.frontend import FbankTables, wav2fbank
    tabs = FbankTables(dev, sample_rate=sr)
    # host tables agree with the oracle's dense filterbank
    s, c, w = OF.sparse_banks(padded=tabs.tables["padded"], sr=float(sr))
    assert np.array_equal(tabs.tables["mel_start_f"].cpu().numpy().astype(np.int32), s)
    assert np.array_equal(tabs.tables["mel_count_f"].cpu().numpy().astype(np.int32), c)
    assert np.allclose(tabs.tables["mel_w"].cpu().numpy(), w, atol=1e-7)
    waves = np.stack([_wave(n_samples, 10 + i) for i in range(batch)])
    old = aum_hip._product
    aum_hip._product = lib
    try:
        out = wav2fbank(torch.tensor(waves, device=dev), tabs, target_length=target).cpu().numpy()
    finally:
        aum_hip._product = old
    for i in range(batch):
        ref = OF.frontend(waves[i].astype(np.float64), target_length=target, sr=sr)
        assert out[i].shape == ref.shape
        # log-mel values are O(1) after normalisation; fp32 FFT vs fp64: a few 1e-5 typical, bins with tiny energy larger
        err = np.abs(out[i] - ref)
        assert err.max() < 2e-3, (err.max(), np.unravel_index(err.argmax(), err.shape))
        assert np.median(err) < 2e-5


def test_fbank_emu():
    import build_emu
    lib = aum_hip.Lib(build_emu.build(), host=True)
    _check(lib, "cpu", 16000 + 37, 120)         # 98 frames + zero padding rows
    _check(lib, "cpu", 9000, 40, batch=1)       # cut: more frames than target_length
    # other sample rates = other FFT sizes: the workgroup-per-frame kernel (the 512-point case above runs one wavefront per frame)
    _check(lib, "cpu", 4000 + 11, 60, batch=1, sr=8000)        # 25 ms = 200 samples -> 256-point FFT
    _check(lib, "cpu", 9000, 20, batch=1, sr=32000)            # 800 samples -> 1024-point FFT


@pytest.mark.gpu
def test_fbank_gpu():
    lib = aum_hip.get()
    _check(lib, "cuda", 160000, 1024, batch=4)   # the AudioSet clip: 998 frames padded to 1024
    _check(lib, "cuda", 16000 + 37, 120)
    _check(lib, "cuda", 8000 * 3, 320, sr=8000)
    _check(lib, "cuda", 32000 * 2, 200, sr=32000)


def test_oracle_matches_transformers_kaldi_fbank():
    """Independent cross-check of oracle/fbank.py: Hugging Face `transformers.audio_utils` carries a numpy implementation of
    Kaldi's fbank (the fallback of ASTFeatureExtractor when torchaudio is missing, written to reproduce
    torchaudio.compliance.kaldi.fbank with the arguments AST -- and the reference's dataloader -- use).  It is not the
    reference's torchaudio call itself (absent here), so the oracle stays formally unpinned, but two independent
    restatements of the published algorithm agree to float32 rounding."""
    import warnings
    au = pytest.importorskip("transformers.audio_utils")
    rng = np.random.default_rng(3)
    sr = 16000
    t = np.arange(2 * sr) / sr
    wave = 0.3 * np.sin(2 * np.pi * 523.0 * t) + 0.2 * np.sin(2 * np.pi * 3111.0 * t + 1.0) + 0.05 * rng.standard_normal(len(t))
    wave = (wave - wave.mean()).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mel_filters = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=sr // 2,
                                         sampling_rate=sr, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    window = au.window_function(400, "hann", periodic=False)
    ref = au.spectrogram(wave, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                         preemphasis=0.97, mel_filters=mel_filters, log_mel="log", mel_floor=1.192092955078125e-07,
                         remove_dc_offset=True).T
    mine = OF.fbank(wave.astype(np.float64))
    assert mine.shape == ref.shape == (198, 128)
    assert np.abs(mine - ref).max() < 2e-5


# ---- waveform -> tokens in one launch (aum_frontend_tokens_fwd) vs the two-stage path (aum_fbank_fwd, then a GEMM) --------
def _tokens_two_stage(spec, w, bias, pos_embed, cls_token, dtype, cls_pos, time_major):
    """spec (B, T, 128) fp32 from aum_fbank_fwd -> what MM:509-541 computes under autocast: 16-bit conv output, fp32 position add"""
    B, T, F = spec.shape
    nf, nt = F // 16, T // 16
    x = spec.to(dtype).double().transpose(1, 2)                                   # B, F, T
    cols = x.reshape(B, nf, 16, nt, 16).permute(0, 1, 3, 2, 4).reshape(B, nf * nt, 256)
    out = cols @ w.reshape(w.shape[0], 256).to(dtype).double().t() + bias.double()
    out = out.float().to(dtype).float() + pos_embed[0, 1:]
    if time_major:
        out = out.reshape(B, nf, nt, -1).transpose(1, 2).reshape(B, nt * nf, -1)
    cls = (cls_token + pos_embed[:, :1]).expand(B, -1, -1)
    return torch.cat((out[:, :cls_pos], cls, out[:, cls_pos:]), dim=1), cols.float().to(dtype).reshape(B * nf * nt, 256)


def _check_tokens(lib, dev, target, dim, dtype, time_major, augment, batch=2, n_samples=None):
    from aum.frontend import FbankTables, prepare_wave
    from aum.augment import draw_augmentation
    tabs = FbankTables(dev)
    n = n_samples or (400 + (target - 1) * 160 - 900)                             # a few zero-padded frames at the end
    waves = torch.tensor(np.stack([_wave(n, 20 + i) for i in range(batch)]), device=dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    w = (torch.randn(dim, 1, 16, 16, generator=g) * 0.06).to(dev)
    bias = (torch.randn(dim, generator=g) * 0.1).to(dev)
    n_patches = target // 16 * 8
    pos_embed = (torch.randn(1, n_patches + 1, dim, generator=g) * 0.02).to(dev)
    cls_token = (torch.randn(1, 1, dim, generator=g) * 0.02).to(dev)
    aug = nz = None
    if augment:
        aug, nz = draw_augmentation(batch, target, 128, 24, target // 5, True, dev, generator=torch.Generator(device=dev).manual_seed(9))
    n_valid = torch.tensor([n - 4000 * (i % 2) for i in range(batch)], device=dev)          # ragged: every second clip is shorter
    wave, aug = prepare_wave(waves, n_valid, tabs, aug)
    cls_pos = n_patches // 2
    old = aum_hip._product
    aum_hip._product = lib
    try:
        spec = aum_hip.fbank_fwd(wave, tabs.tables, target, -4.27, 4.57, aug=aug, noise=nz)
        tok, patches = aum_hip.frontend_tokens(wave, tabs.tables, target, -4.27, 4.57, w.reshape(dim, 256).to(dtype).contiguous(), bias,
                                               pos_embed[0, 1:].contiguous(), (cls_token + pos_embed[:, :1]).reshape(dim).contiguous(),
                                               cls_pos, time_major=time_major, save_patches=True, aug=aug, noise=nz)
        if dev == "cuda":
            torch.cuda.synchronize()
    finally:
        aum_hip._product = old
    ref, ref_patches = _tokens_two_stage(spec, w, bias, pos_embed, cls_token, dtype, cls_pos, time_major)
    assert tok.shape == ref.shape and tok.dtype == torch.float32
    assert torch.equal(patches.view(torch.int16), ref_patches.view(torch.int16))          # the same log-mel values, bit for bit
    err = (tok - ref).abs()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    # fp32 MFMA accumulation vs fp64: the 16-bit rounding of the conv output may land one step apart on a few values
    assert err.max() <= 1.01 * ulp * max(1.0, ref.abs().max().item()), err.max()
    assert (err > 1e-5).float().mean() < 0.02
    assert torch.equal(tok[:, cls_pos], ref[:, cls_pos])


def test_frontend_tokens_emu():
    import build_emu
    lib = aum_hip.Lib(build_emu.build(), host=True)
    _check_tokens(lib, "cpu", 64, 48, torch.bfloat16, False, False, batch=1)
    _check_tokens(lib, "cpu", 128, 32, torch.float16, True, True)


@pytest.mark.gpu
def test_frontend_tokens_gpu():
    lib = aum_hip.get()
    _check_tokens(lib, "cuda", 1024, 768, torch.bfloat16, False, True, batch=4)
    _check_tokens(lib, "cuda", 1024, 768, torch.bfloat16, False, False, batch=3, n_samples=160000)
    _check_tokens(lib, "cuda", 128, 192, torch.float16, True, True)


def _check_model_tokens(lib, dev, target, dim, time_major):
    """AudioMamba.tokens_from_wave (one launch + its autograd function) vs AudioMamba.tokens on the stand-alone log-mel output,
    both under bf16 autocast: token values and the gradients of patch_embed / pos_embed / cls_token."""
    from aum.model import AudioMamba
    from aum.frontend import FbankTables, WaveInput, prepare_wave
    old = aum_hip._product
    aum_hip._product = lib
    try:
        torch.manual_seed(0)
        model = AudioMamba(spectrogram_size=(128, target), depth=1, embed_dim=dim, num_classes=3, transpose_token_sequence=time_major).to(dev)
        with torch.no_grad():
            model.patch_embed.proj.bias.normal_(0, 0.1)
        tabs = FbankTables(dev)
        n = 400 + (target - 1) * 160
        waves = torch.tensor(np.stack([_wave(n, 30 + i) for i in range(2)]), device=dev)
        wave, aug = prepare_wave(waves, torch.tensor([n, n - 3000], device=dev), tabs)
        fe = WaveInput(tabs, target, aug=aug)
        probe = torch.randn(2, target // 16 * 8 + 1, dim, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        grads = []
        toks = []
        for fused in (True, False):
            model.zero_grad()
            with torch.autocast(dev, dtype=torch.bfloat16):
                tok, pos = model.tokens_from_wave(wave, fe) if fused else model.tokens(fe.spectrogram(wave))
            assert tok.dtype == torch.float32 and pos == (target // 16 * 8) // 2
            (tok * probe).sum().backward()
            toks.append(tok.detach())
            grads.append([p.grad.clone() for p in (model.patch_embed.proj.weight, model.patch_embed.proj.bias,
                                                   model.pos_embed.pos_embed, model.cls_token)])
    finally:
        aum_hip._product = old
    assert (toks[0] - toks[1]).abs().max() <= 2.0 ** -7 * max(1.0, toks[1].abs().max().item())
    # both sides round an fp32-accumulated GEMM to bf16; the library's summation order differs, so a share of the values lands one
    # bf16 step apart (never more: the max above) -- on average well under half a step
    assert (toks[0] - toks[1]).abs().mean() <= 2.0 ** -10 * toks[1].abs().mean()
    for name, a, b in zip(("weight", "bias", "pos_embed", "cls_token"), *grads):
        assert a.shape == b.shape
        tol = 2e-2 if name in ("weight", "bias") else 1e-5          # weight/bias gradients pass through a bf16 GEMM output on both sides
        assert (a - b).abs().max() <= tol * max(1e-6, b.abs().max().item()), (name, (a - b).abs().max(), b.abs().max())


def test_model_tokens_from_wave_emu():
    import build_emu
    lib = aum_hip.Lib(build_emu.build(), host=True)
    _check_model_tokens(lib, "cpu", 64, 32, False)
    _check_model_tokens(lib, "cpu", 64, 32, True)


@pytest.mark.gpu
def test_model_tokens_from_wave_gpu():
    _check_model_tokens(aum_hip.get(), "cuda", 1024, 768, False)
    _check_model_tokens(aum_hip.get(), "cuda", 128, 192, True)


# ---- real audio: excerpts of the reference's five example clips (tests/golden/wav_excerpts.npz, made by make_golden.py) ----------
def _check_real_audio(lib, dev, target):
    """Kernel vs oracle on real recordings (quiet stretches, onsets, spectral tilt) as ONE ragged batch: the clips have different
    lengths, so this is also the zero-padding path of DL:139-145 on real data.  Bins whose energy sits at the fp32 noise floor of the
    frame's FFT (~1e-7 of its strongest bin) are where fp32 and fp64 may differ visibly after the log; they are bounded separately."""
    from aum.frontend import FbankTables, wav2fbank_ragged
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wav_excerpts.npz"))
    clips = [z[f"sample{i}"].astype(np.float32) / 32768.0 for i in range(5)]
    n = [len(c) for c in clips]
    batch = np.zeros((5, max(n)), np.float32)
    for i, c in enumerate(clips):
        batch[i, :len(c)] = c - c.mean()                 # mean removal over the clip itself (dataloader.py:101), then zero padding
    tabs = FbankTables(dev)
    old = aum_hip._product
    aum_hip._product = lib
    try:
        # wav2fbank_ragged removes the batch row's mean again: rows are already zero-mean over their valid part, the padding adds
        # n_pad zeros -> the row mean is 0 up to rounding, as for a clip processed on its own
        out = wav2fbank_ragged(torch.tensor(batch, device=dev), torch.tensor(n), tabs, target_length=target).cpu().numpy()
    finally:
        aum_hip._product = old
    for i, c in enumerate(clips):
        ref = OF.frontend(c.astype(np.float64), target_length=target)
        err = np.abs(out[i] - ref)
        assert np.median(err) < 1e-6, (i, np.median(err))              # measured 2e-8 .. 1e-7
        assert np.quantile(err, 0.999) < 2e-4, (i, np.quantile(err, 0.999))
        assert err.max() < 1e-3, (i, err.max())                        # measured <= 1.1e-4 (values are O(1); the log floor is reached)
        assert ref.min() < -1.27                                       # digital silence: log(eps) after normalisation
        frames = 1 + (n[i] - 400) // 160
        assert np.allclose(out[i, frames:], (0.0 + 4.2677393) / (2 * 4.5689974), atol=1e-6)     # zero padding, normalised


def test_fbank_real_audio_emu():
    import build_emu
    _check_real_audio(aum_hip.Lib(build_emu.build(), host=True), "cpu", 304)


@pytest.mark.gpu
def test_fbank_real_audio_gpu():
    _check_real_audio(aum_hip.get(), "cuda", 320)

