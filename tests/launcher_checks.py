"""Bodies of the SURVEY 8(f1-f3) parity checks, shared by the CPU tests (tests/test_train_launcher.py: the lane-array build stands in
for libaum_hip.so) and the GPU tests (tests/test_gpu_launcher.py: the product library on an MI355X).  The expected values are the
reference's own (tests/golden/launcher.npz, tests/golden/make_golden.py::launcher_goldens) or separate torch ops on the same stream."""
import os

import numpy as np
import torch

def check_fused_augmentation(device):
    """SpecAug bands, noise and roll applied inside the log-mel kernel's store (aug= / noise= of aum_fbank_fwd) give
    the spectrogram the separate torch ops give after the kernel, for the same generator stream; ragged clips included."""
    from aum.frontend import FbankTables, wav2fbank, wav2fbank_ragged, pad_fill
    from aum.augment import spec_augment, noise_roll, draw_augmentation
    tabs = FbankTables(device)
    g = torch.Generator().manual_seed(3)
    lens = [5200, 3000, 4100, 5200, 300]
    wave_b = torch.zeros(5, 5200)
    for i, n in enumerate(lens):
        w = torch.randn(n, generator=g) * 0.1
        wave_b[i, :n] = w - w.mean()
    wave_b = wave_b.to(device)
    T_, F_ = 32, 128
    # separate ops (what the launcher did before): kernel -> pad ragged -> masks -> noise + roll
    plain = wav2fbank(wave_b, tabs, target_length=T_)
    frames = torch.tensor([0 if n < 400 else min(T_, 1 + (n - 400) // 160) for n in lens], device=device)
    plain = plain.masked_fill((torch.arange(T_, device=device)[None, :] >= frames[:, None])[:, :, None], pad_fill())
    g1 = torch.Generator(device=device).manual_seed(11)
    want = noise_roll(spec_augment(plain, 24, 10, pad_fill(), generator=g1), generator=g1)
    g2 = torch.Generator(device=device).manual_seed(11)
    aug, nz = draw_augmentation(5, T_, F_, 24, 10, True, device, generator=g2)
    got = wav2fbank_ragged(wave_b, torch.tensor(lens, device=device), tabs, target_length=T_, aug=aug, noise=nz)
    assert torch.allclose(got, want, atol=1e-5), float((got - want).abs().max())
    assert (aug[:, 2] > aug[:, 1]).any() and (aug[:, 4] > aug[:, 3]).any() and (aug[:, 5] != 0).any()
    # no augmentation: the plain ragged result
    assert torch.allclose(wav2fbank_ragged(wave_b, torch.tensor(lens, device=device), tabs, target_length=T_), plain, atol=1e-6)


def check_fused_augmentation_vs_oracle(device):
    """f3 against an INDEPENDENT statement: raw uniform draws from numpy go (i) through
    aum.augment.pack_augmentation into the log-mel kernel's fused store and (ii) with the un-normalised, un-augmented log-mel of the same
    clips through oracle/augment.py -- DL:206-228 and torchaudio's mask_along_axis rule restated in numpy, clip by clip.  (Still
    "unpinned": torchaudio itself is not in the image.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import augment as OA
    from aum.frontend import FbankTables, wav2fbank_ragged, AUDIOSET_MEAN, AUDIOSET_STD
    from aum.augment import pack_augmentation
    tabs = FbankTables(device)
    rng = np.random.default_rng(21)
    lens = [5200, 3000, 4100, 5200, 300, 4800]
    Bn, T_, F_ = len(lens), 32, 128
    wave = np.zeros((Bn, 5200), np.float32)
    for i, n in enumerate(lens):
        w = rng.standard_normal(n).astype(np.float32) * 0.1
        wave[i, :n] = w - w.mean()
    wave_t, lens_t = torch.tensor(wave, device=device), torch.tensor(lens, device=device)
    plain = wav2fbank_ragged(wave_t, lens_t, tabs, target_length=T_).double().cpu().numpy()       # normalised, padded rows = pad fill
    raw = plain * (2 * AUDIOSET_STD) + AUDIOSET_MEAN                                               # un-normalised: padded rows -> 0
    for freqm, timem, noise in ((24, 10, True), (48, 0, False), (0, 20, True), (0, 0, True)):
        d = {k: rng.random(Bn) for k in ("u_fv", "u_fm", "u_tv", "u_tm", "u_amp")}
        field, shift = rng.random((Bn, T_, F_)), rng.integers(-10, 10, Bn)
        tt = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        kw = {}
        if freqm:
            kw.update(u_fv=tt(d["u_fv"]), u_fm=tt(d["u_fm"]))
        if timem:
            kw.update(u_tv=tt(d["u_tv"]), u_tm=tt(d["u_tm"]))
        if noise:
            kw.update(u_amp=tt(d["u_amp"]), field=tt(field), shift=torch.tensor(shift, device=device))
        aug, nz = pack_augmentation(T_, F_, freqm, timem, **kw)
        got = wav2fbank_ragged(wave_t, lens_t, tabs, target_length=T_, aug=aug, noise=nz).double().cpu().numpy()
        masked_any = False
        for b in range(Bn):
            # the oracle sees the float32 draws the kernel's table was built from (the band limits are floors of products of them)
            draws = {k: float(np.float32(d[k][b])) for k in d}
            draws.update(field=field[b].astype(np.float32), shift=int(shift[b]))
            want = OA.augment_clip(raw[b], draws, freqm, timem, noise, AUDIOSET_MEAN, AUDIOSET_STD)
            assert np.abs(got[b] - want).max() < 2e-5, (freqm, timem, noise, b, float(np.abs(got[b] - want).max()))
            masked_any = masked_any or bool(np.abs(want - plain[b]).max() > 0.1)
        assert masked_any


def check_training_loop(tmp_path, monkeypatch):
    """aum.train.train against the reference's own src/traintest.py `train` (golden/launcher.npz, generated by
    tests/golden/make_golden.py::launcher_goldens): Adam with batch-scaled betas / eps (TT:25-33), the warm-up stairs that start
    at lr = 0 and later override the MultiStepLR decay (TT:118-124), MultiStepLR (TT:72-74), BCE training loss, the
    sigmoid-into-BCE validation loss (TT:266-283), result.csv, and the parameters after 8 optimizer steps."""
    import cases
    from conftest import load_golden, rel_err
    from aum import train as T
    from aum.model import AudioMamba
    g = load_golden("launcher")
    t = cases.TRAIN_CASE
    tr, va = cases.train_inputs()

    class DS(torch.utils.data.Dataset):
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return len(self.d["x"])

        def __getitem__(self, i):       # (wave, n_valid, labels, path): the "waveform" already is the spectrogram here
            return torch.tensor(self.d["x"][i]), 0, torch.tensor(self.d["y"][i]), f"clip{i}"

    class IdentityFrontend:
        def __init__(self, *a, **kw):
            pass

        def __call__(self, wave, n_valid):
            return wave, None
    monkeypatch.setattr(T, "Frontend", IdentityFrontend)
    lrs, hyper = [], {}
    RealAdam = torch.optim.Adam

    class SpyAdam(RealAdam):
        def __init__(self, params, lr, **kwargs):
            hyper.update(dict(lr=lr, **kwargs))
            super().__init__(params, lr, **kwargs)

        def step(self, *a, **k):
            lrs.append(self.param_groups[0]["lr"])
            return super().step(*a, **k)
    monkeypatch.setattr(torch.optim, "Adam", SpyAdam)
    exp = str(tmp_path / "exp")
    os.makedirs(exp + "/models")
    args = T.build_parser().parse_args(
        ["--model", "aum", "--n_class", str(t["n_class"]), "--lr", str(t["lr"]), "--n-epochs", str(t["n_epochs"]), "-b", str(t["batch"]),
         "--bs_scale_factor", str(t["bs_scale_factor"]), "--lrscheduler_start", str(t["lrscheduler_start"]),
         "--lrscheduler_step", str(t["lrscheduler_step"]), "--lrscheduler_decay", str(t["lrscheduler_decay"]), "--warmup", "True",
         "--loss", "BCE", "--metrics", "mAP", "--mixed_precision", "no", "--exp-dir", exp, "--weight_decay", str(t["weight_decay"]),
         "--n-print-steps", "1000"])
    model = AudioMamba(spectrogram_size=t["spec"], depth=t["depth"], embed_dim=t["embed_dim"], num_classes=t["n_class"], bimamba_type="v1")
    init = cases.model_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, "train_init")
    model.load_state_dict({k: torch.tensor(v) for k, v in init.items()})
    T.train(model, torch.utils.data.DataLoader(DS(tr), batch_size=t["batch"], shuffle=False),
            torch.utils.data.DataLoader(DS(va), batch_size=2 * t["batch"], shuffle=False), args, T.Dist())
    # optimizer: same hyper-parameters, same learning rate at every step
    assert np.allclose([hyper["betas"][0], hyper["betas"][1], hyper["eps"], hyper["weight_decay"], hyper["lr"]], g["train.adam"], rtol=1e-12)
    assert len(lrs) == len(g["train.lr_per_step"]) and np.allclose(lrs, g["train.lr_per_step"], rtol=1e-12, atol=0)
    # result.csv: metrics, train loss, valid loss (sigmoid outputs fed to BCEWithLogits, as the reference does), lr
    res = np.loadtxt(exp + "/result.csv", delimiter=",")
    assert res.shape == g["train.result"].shape
    assert np.allclose(res[:, :5], g["train.result"][:, :5], atol=1e-6), (res[:, :5], g["train.result"][:, :5])
    assert np.allclose(res[:, 5:7], g["train.result"][:, 5:7], rtol=2e-5), (res[:, 5:7], g["train.result"][:, 5:7])
    assert np.allclose(res[:, 7], g["train.result"][:, 7], rtol=1e-12)
    pred = np.loadtxt(exp + f"/predictions/predictions_{t['n_epochs']}.csv", delimiter=",")
    assert rel_err(pred, g["train.predictions"]) < 1e-4
    # parameters after training: compare what the 8 steps CHANGED (the updates are ~1e-4 of the weights)
    worst = 0.0
    for k, v in model.state_dict().items():
        ref_delta = g["train.final." + k].astype(np.float64) - init[k]
        got_delta = v.cpu().numpy().astype(np.float64) - init[k]
        scale = np.abs(ref_delta).max()
        if scale > 0:
            worst = max(worst, np.abs(got_delta - ref_delta).max() / scale)
    assert worst < 2e-2, worst


def check_checkpoint_regrid(device):
    """load_aum_checkpoint against the reference's AudioMamba(aum_pretrain=True) (MM:397-446, FlexiPosEmbed +
    resample_abs_pos_embed TOK:26-66, 349-372): a `module.`-prefixed 128 x 256-frame, 7-class checkpoint loaded into a 128 x
    512-frame, 3-class model -- re-gridded position embedding and the logits of the loaded model (golden/launcher.npz)."""
    import cases
    from conftest import load_golden, rel_err
    from aum.model import AudioMamba
    from aum.checkpoint import load_aum_checkpoint
    g = load_golden("launcher")
    c = cases.CKPT_CASE
    kw = dict(depth=c["depth"], embed_dim=c["embed_dim"], bimamba_type="v1")
    src = AudioMamba(spectrogram_size=c["src_spec"], num_classes=c["src_classes"], **kw)
    vals = cases.model_state({k: tuple(v.shape) for k, v in src.state_dict().items()}, "ckpt_src")
    assert abs(cases.checksum(dict(vals, **cases.ckpt_inputs())) - float(g["ckpt.checksum"])) < 1e-6 * abs(float(g["ckpt.checksum"]))
    dst = AudioMamba(spectrogram_size=c["dst_spec"], num_classes=c["dst_classes"], **kw)
    res = load_aum_checkpoint(dst, {"module." + k: torch.tensor(v) for k, v in vals.items()})
    assert set(res.missing_keys) == {"head.weight", "head.bias"}
    head = cases.model_state({k: tuple(v.shape) for k, v in dst.state_dict().items() if k.startswith("head.")}, "ckpt_dst_head")
    dst.load_state_dict({k: torch.tensor(v) for k, v in head.items()}, strict=False)
    dst = dst.to(device)
    assert rel_err(dst.pos_embed.pos_embed.detach().cpu().numpy(), g["ckpt.pos_embed"]) < 1e-6
    with torch.no_grad():
        logits = dst(torch.tensor(cases.ckpt_inputs()["x"], device=device))
    assert rel_err(logits.cpu().numpy(), g["ckpt.logits"]) < 1e-4

