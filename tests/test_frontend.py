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


def _check(lib, dev, n_samples, target, batch=2):
    from aum.frontend import FbankTables, wav2fbank
    tabs = FbankTables(dev)
    # host tables agree with the oracle's dense filterbank
    s, c, w = OF.sparse_banks()
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
        ref = OF.frontend(waves[i].astype(np.float64), target_length=target)
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


@pytest.mark.gpu
def test_fbank_gpu():
    lib = aum_hip.get()
    _check(lib, "cuda", 160000, 1024, batch=4)   # the AudioSet clip: 998 frames padded to 1024
    _check(lib, "cuda", 16000 + 37, 120)


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
