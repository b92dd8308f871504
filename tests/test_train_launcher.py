"""launcher, data files, augmentation, statistics and checkpoint loading around the hot path.
CPU only: the lane-array build stands in for libaum_hip.so (as in test_host_package.py)."""
import json
import os
import pickle
import sys
import wave

import numpy as np
import pytest
import torch

import aum_hip
import launcher_checks

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module", autouse=True)
def emu_as_product():
    import build_emu
    old = aum_hip._product
    aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    yield
    aum_hip._product = old


def _write_wav(path, x, sr=16000, width=2):
    with wave.open(path, "wb") as f:
        f.setnchannels(1), f.setsampwidth(width), f.setframerate(sr)
        if width == 2:
            f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        else:
            f.writeframes((np.clip(x, -1, 1) * 2147483647).astype("<i4").tobytes())


@pytest.fixture(scope="module")
def toy_dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("toy")
    rng = np.random.default_rng(0)
    mids = [f"/m/{i:03d}" for i in range(4)]
    with open(d / "labels.csv", "w") as f:
        f.write("index,mid,display_name\n" + "".join(f"{i},{m},\"class {i}\"\n" for i, m in enumerate(mids)))
    items = []
    for i in range(10):
        n = int(rng.integers(6000, 12000))
        t = np.arange(n) / 16000.0
        x = 0.3 * np.sin(2 * np.pi * (300 + 400 * (i % 4)) * t) + 0.02 * rng.standard_normal(n)
        p = str(d / f"clip{i}.wav")
        _write_wav(p, x, width=2 if i % 2 else 4)
        lab = mids[i % 4] + ("," + mids[(i + 1) % 4] if i % 3 == 0 else "")
        items.append({"wav": p, "labels": lab})
    with open(d / "train.json", "w") as f:
        json.dump({"data": items}, f)
    with open(d / "val.json", "w") as f:
        json.dump({"data": items[:6]}, f)
    return d


def test_read_audio_widths(tmp_path):
    from aum.data import read_audio
    x = np.linspace(-0.9, 0.9, 1000).astype(np.float32)
    for width, tol in ((2, 1e-4), (4, 1e-6)):
        p = str(tmp_path / f"w{width}.wav")
        _write_wav(p, x, width=width)
        y, sr = read_audio(p)
        assert sr == 16000 and y.shape == x.shape and np.abs(y - x).max() < tol
    np.save(tmp_path / "a.npy", x)
    y, sr = read_audio(str(tmp_path / "a.npy"))
    assert sr is None and np.array_equal(y, x)
    with pytest.raises(ValueError):
        read_audio(str(tmp_path / "a.flac"))


def test_dataset_items_and_mixup(toy_dataset):
    from aum.data import WaveformDataset
    ds = WaveformDataset(str(toy_dataset / "train.json"), str(toy_dataset / "labels.csv"), 10480)
    w, n, lab, path = ds[0]
    assert w.shape == (10480,) and 6000 <= n <= 10480 and path.endswith("clip0.wav")
    from aum.data import read_audio
    raw, _ = read_audio(path)
    assert n == min(len(raw), 10480) and torch.allclose(w[:n], torch.from_numpy(raw - raw.mean())[:n])
    assert n == 10480 or float(w[n:].abs().max()) == 0
    assert lab.tolist() == [1.0, 1.0, 0.0, 0.0]                       # "/m/000,/m/001"
    mix = WaveformDataset(str(toy_dataset / "train.json"), str(toy_dataset / "labels.csv"), 10480, mixup=1.0)
    w, n, lab, _ = mix[1]
    assert 1.99 < float(lab.sum()) < 3.01 or 0.99 < float(lab.sum()) < 2.01      # lam * |labels1| + (1 - lam) * |labels2|
    assert n == 10480 or abs(float(w[:n].mean())) < 1e-6                         # re-centred after mixing (DL:129)


def test_spec_augment_and_noise():
    from aum.augment import spec_augment, noise_roll
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 64, 32)
    y = spec_augment(x, 8, 12, fill=-7.0, generator=g)
    m = y == -7.0
    assert m.any()
    for b in range(16):
        fcols = m[b].all(dim=0).nonzero().flatten()                    # fully masked mel bins: one contiguous band < 8
        trows = m[b].all(dim=1).nonzero().flatten()
        for idx, lim in ((fcols, 8), (trows, 12)):
            if len(idx):
                assert len(idx) < lim and int(idx[-1] - idx[0]) + 1 == len(idx)
        keep = ~m[b]
        assert torch.equal(y[b][keep], x[b][keep])
    assert torch.equal(spec_augment(x, 0, 0), x)
    z = noise_roll(x, generator=g)
    assert z.shape == x.shape and not torch.equal(z, x)
    # rolling by s and adding noise in [0, 0.1): some shift in [-10, 10) explains every clip
    for b in range(4):
        best = min(float((z[b] - torch.roll(x[b], s, 0)).abs().max()) for s in range(-10, 10))
        assert best <= 0.1 + 1e-6


def test_ragged_frontend_matches_per_clip():
    from aum.frontend import FbankTables, wav2fbank, wav2fbank_ragged, pad_fill
    tabs = FbankTables("cpu")
    g = torch.Generator().manual_seed(1)
    lens = [3000, 5200, 300, 4000]
    wave_b = torch.zeros(4, 5200)
    for i, n in enumerate(lens):
        w = torch.randn(n, generator=g) * 0.1
        wave_b[i, :n] = w - w.mean()
    out = wav2fbank_ragged(wave_b, torch.tensor(lens), tabs, target_length=32)
    for i, n in enumerate(lens):
        ref = wav2fbank(wave_b[i:i + 1, :n].contiguous(), tabs, target_length=32)[0]
        assert torch.allclose(out[i], ref, atol=2e-4), i
    assert torch.allclose(out[2], torch.full_like(out[2], pad_fill()), atol=1e-6)     # shorter than one window: all padding


def test_fused_augmentation_matches_separate_ops():
    """SURVEY 8(f3) on the lane-array build (GPU: tests/test_gpu_launcher.py)"""
    launcher_checks.check_fused_augmentation("cpu")


def test_fused_augmentation_matches_independent_oracle():
    """f3 against oracle/augment.py for fixed raw draws, on the lane-array build (GPU: tests/test_gpu_launcher.py)"""
    launcher_checks.check_fused_augmentation_vs_oracle("cpu")


def test_stats_against_sklearn():
    from sklearn import metrics
    from aum.stats import calculate_stats, summarize, d_prime
    rng = np.random.default_rng(0)
    tgt = (rng.random((200, 5)) < 0.3).astype(np.float32)
    out = np.clip(tgt * 0.4 + rng.random((200, 5)) * 0.6, 0, 1)
    st = calculate_stats(out, tgt)
    s = summarize(st, "mAP")
    assert abs(s["mAP"] - metrics.average_precision_score(tgt, out, average="macro")) < 1e-9
    assert abs(s["mAUC"] - metrics.roc_auc_score(tgt, out, average="macro")) < 1e-9
    assert abs(d_prime(0.5)) < 1e-12 and abs(d_prime(0.9) - 1.8124) < 1e-3
    assert s["main"] == s["mAP"] and summarize(st, "acc")["main"] == st[0]["acc"]


def test_checkpoint_prefix_head_and_regrid():
    from aum.model import AudioMamba
    from aum.checkpoint import load_aum_checkpoint, resample_pos_embed
    torch.manual_seed(0)
    kw = dict(depth=1, embed_dim=32, patch_size=(16, 16), strides=(16, 16))
    src = AudioMamba(spectrogram_size=(128, 256), num_classes=7, **kw)            # 8 x 16 grid
    sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    same = AudioMamba(spectrogram_size=(128, 256), num_classes=7, **kw)
    res = load_aum_checkpoint(same, sd)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in src.state_dict().items():
        assert torch.equal(same.state_dict()[k], v), k
    # different clip length and class count: backbone loads, pos-embed is re-gridded, head stays at its init
    dst = AudioMamba(spectrogram_size=(128, 512), num_classes=3, **kw)            # 8 x 32 grid
    head0 = dst.head.weight.clone()
    res = load_aum_checkpoint(dst, sd)
    assert set(res.missing_keys) == {"head.weight", "head.bias"} and torch.equal(dst.head.weight, head0)
    pe_src, pe_dst = src.pos_embed.pos_embed, dst.pos_embed.pos_embed
    assert pe_dst.shape == (1, 1 + 8 * 32, 32) and torch.equal(pe_dst[:, 0], pe_src[:, 0])
    want = torch.nn.functional.interpolate(pe_src[:, 1:].reshape(1, 8, 16, 32).permute(0, 3, 1, 2), size=(8, 32),
                                           mode="bilinear", antialias=True).permute(0, 2, 3, 1).reshape(1, 256, 32)
    assert torch.allclose(pe_dst[:, 1:], want)
    assert resample_pos_embed(pe_src, (8, 16), (8, 16)) is pe_src
    with pytest.raises(RuntimeError):
        bad = dict(sd)
        bad.pop("module.norm_f.weight")
        load_aum_checkpoint(AudioMamba(spectrogram_size=(128, 256), num_classes=7, **kw), bad)


def test_launcher_train_then_eval(toy_dataset, tmp_path):
    from aum import train as T
    exp = str(tmp_path / "exp")
    common = ["--model", "aum", "--model_type", "tiny", "--depth", "2", "--aum_type", "Fo-Bi", "--n_class", "4",
              "--label-csv", str(toy_dataset / "labels.csv"), "--data-val", str(toy_dataset / "val.json"),
              "--audio_length", "64", "--num-workers", "0", "-b", "4", "--loss", "BCE", "--metrics", "mAP",
              "--mixed_precision", "no", "--exp-dir", exp]
    T.main(common + ["--data-train", str(toy_dataset / "train.json"), "--n-epochs", "2", "--freqm", "8", "--timem", "8",
                     "--mixup", "0.5", "--noise", "True", "--warmup", "True", "--lr", "1e-3", "--n-print-steps", "1"])
    res = np.loadtxt(exp + "/result.csv", delimiter=",")
    assert res.shape == (2, 8) and np.isfinite(res).all() and 0 <= res[0, 0] <= 1
    for f in ("args.pkl", "progress.pkl", "stats_1.pickle", "stats_2.pickle", "predictions/target.csv",
              "predictions/predictions_2.csv", "models/best_audio_model.pth", "models/latest_audio_model.2.pth",
              "models/best_optim_state.pth"):
        assert os.path.exists(os.path.join(exp, f)), f
    prog = pickle.load(open(exp + "/progress.pkl", "rb"))
    assert len(prog) == 2 and prog[-1][0] == 2 and prog[-1][1] == 6           # 10 clips / batch 4 -> 3 steps / epoch
    assert np.loadtxt(exp + "/predictions/target.csv", delimiter=",").shape == (6, 4)
    # evaluation run initialised from the checkpoint the training run wrote
    exp2 = str(tmp_path / "exp_eval")
    T.main([a if a != exp else exp2 for a in common] +
           ["--run_type", "eval", "--aum_pretrain", "True", "--aum_pretrain_path", exp + "/models/latest_audio_model.2.pth"])
    r2 = np.loadtxt(exp2 + "/result_eval.csv", delimiter=",")
    assert r2.shape == (6,) and abs(r2[0] - res[1, 0]) < 1e-6 and abs(r2[5] - res[1, 6]) < 1e-5


def test_hot_loop_has_no_per_step_host_sync(toy_dataset, tmp_path, monkeypatch):
    """SURVEY 5 / TT:157-174: the reference reads the loss back and gathers it every step.  Here the running loss stays on the
    device; the loop touches the host only every --n-print-steps steps and once per epoch."""
    from aum import train as T
    calls = {"item": 0, "tolist": 0, "at_validate": None, "fe_calls": 0, "at_step2": None}
    real_item, real_tolist, real_validate = torch.Tensor.item, torch.Tensor.tolist, T.validate
    pkg = os.sep + "audio-mamba-aum_amd" + os.sep

    def spy(real, name):      # only calls made by the package itself count (torch's CPU Adam reads its host-side step counters)
        def f(self):
            if pkg in sys._getframe(1).f_code.co_filename:
                calls[name] += 1
            return real(self)
        return f
    monkeypatch.setattr(torch.Tensor, "item", spy(real_item, "item"))
    monkeypatch.setattr(torch.Tensor, "tolist", spy(real_tolist, "tolist"))

    def validate_spy(*a, **kw):
        if calls["at_validate"] is None:
            calls["at_validate"] = (calls["item"], calls["tolist"])
        return real_validate(*a, **kw)
    monkeypatch.setattr(T, "validate", validate_spy)
    real_fe = T.Frontend.__call__

    def fe_spy(self, *a, **kw):          # the training frontend runs once per step: the second call marks the steady state
        calls["fe_calls"] += 1
        if calls["fe_calls"] == 2:
            calls["at_step2"] = (calls["item"], calls["tolist"])
        return real_fe(self, *a, **kw)
    monkeypatch.setattr(T.Frontend, "__call__", fe_spy)
    before = dict(T.HOST_SYNCS)
    T.main(["--model", "aum", "--model_type", "tiny", "--depth", "1", "--n_class", "4", "--label-csv", str(toy_dataset / "labels.csv"),
            "--data-train", str(toy_dataset / "train.json"), "--data-val", str(toy_dataset / "val.json"), "--audio_length", "64",
            "--num-workers", "0", "-b", "2", "--mixed_precision", "no", "--exp-dir", str(tmp_path / "exp"), "--n-epochs", "1",
            "--n-print-steps", "1000", "--freqm", "0", "--timem", "0", "--mixup", "0"])
    assert T.HOST_SYNCS["steps"] - before["steps"] == 5                  # 10 clips / batch 2
    assert T.HOST_SYNCS["loss_syncs"] - before["loss_syncs"] == 1         # the end-of-epoch exchange only
    # steps 2..5 and the end-of-epoch exchange: no .item() at all, one .tolist() (the exchange)
    assert (calls["at_validate"][0] - calls["at_step2"][0], calls["at_validate"][1] - calls["at_step2"][1]) == (0, 1), calls


def test_validation_drops_sampler_padding():
    """rows gathered from world ranks, DistributedSampler order: positions >= len(dataset) are the sampler's padding"""
    from aum.train import _gather_order
    # 7 samples, 2 ranks -> 4 per rank (one pad), batch 3: steps of 3 + 1 rows per rank
    order = _gather_order(2, 3, 4, 2)
    assert order.tolist() == [0, 2, 4, 1, 3, 5, 6, 7]
    assert (order < 7).sum() == 7 and sorted(order[order < 7].tolist()) == list(range(7))


def test_training_loop_matches_reference_traintest(tmp_path, monkeypatch):
    """SURVEY 8(f1) on the lane-array build (GPU: tests/test_gpu_launcher.py)"""
    launcher_checks.check_training_loop(tmp_path, monkeypatch)


def test_checkpoint_regrid_matches_reference_constructor():
    """SURVEY 8(f2) on the lane-array build (GPU: tests/test_gpu_launcher.py)"""
    launcher_checks.check_checkpoint_regrid("cpu")


def test_launcher_rejects_out_of_scope():
    from aum import train as T
    for extra in (["--model", "ast"], ["--dataset", "epic_sounds"], ["--flexible_training", "True"]):
        with pytest.raises(NotImplementedError):
            T.check_scope(T.build_parser().parse_args(extra))


def test_launcher_two_ranks_gloo(toy_dataset, tmp_path):
    """world_size 2: DistributedSampler shards, DDP step, ragged validation batches gathered to rank 0"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = str(tmp_path / "exp2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "launcher_worker.py"),
           "--model_type", "tiny", "--depth", "1", "--n_class", "4", "--label-csv", str(toy_dataset / "labels.csv"),
           "--data-train", str(toy_dataset / "train.json"), "--data-val", str(toy_dataset / "val.json"),
           "--audio_length", "64", "--num-workers", "0", "-b", "2", "--mixed_precision", "no", "--exp-dir", exp,
           "--n-epochs", "1", "--metrics", "acc", "--loss", "CE"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = np.loadtxt(exp + "/result.csv", delimiter=",").reshape(1, 8)
    assert np.isfinite(res).all()
    assert np.loadtxt(exp + "/predictions/target.csv", delimiter=",").shape == (6, 4)    # 6 clips over 2 ranks, all kept
    sd = torch.load(exp + "/models/best_audio_model.pth")
    assert all(k.startswith("module.") for k in sd)                          # what the reference's DDP runs save
    for r_ in (0, 1):    # 5 clips per rank / batch 2 = 3 steps; the loss crossed ranks once (end of epoch), never per step
        hs = json.load(open(exp + f"/host_syncs_rank{r_}.json"))
        assert hs == {"steps": 3, "loss_syncs": 1}, hs


def test_launcher_two_ranks_gloo_nan_steps(toy_dataset, tmp_path):
    """world_size 2 with --if_nan2num False --if_continue_inf True (TT:153-164) and ONE rank's loss non-finite at steps 0 and 2: the
    finite flag is MIN-reduced, so both ranks skip the same steps (no rank is left waiting in the gradient all-reduce), nothing
    non-finite reaches the parameters, the skipped steps are not counted and the run finishes with finite metrics."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = str(tmp_path / "exp_nan")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "launcher_worker.py"),
           "--model_type", "tiny", "--depth", "1", "--n_class", "4", "--label-csv", str(toy_dataset / "labels.csv"),
           "--data-train", str(toy_dataset / "train.json"), "--data-val", str(toy_dataset / "val.json"),
           "--audio_length", "64", "--num-workers", "0", "-b", "2", "--mixed_precision", "no", "--exp-dir", exp,
           "--n-epochs", "1", "--metrics", "acc", "--loss", "CE", "--if_nan2num", "False", "--if_continue_inf", "True"]
    env = dict(os.environ, OMP_NUM_THREADS="2", AUM_TEST_NAN_STEPS="0,2", AUM_TEST_NAN_RANK="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("Loss is not finite on some rank, continuing training") == 2, r.stdout[-2000:]
    res = np.loadtxt(exp + "/result.csv", delimiter=",").reshape(1, 8)
    assert np.isfinite(res).all()
    sd = torch.load(exp + "/models/best_audio_model.pth")
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    for r_ in (0, 1):    # 3 steps per rank, two of them skipped on BOTH ranks
        hs = json.load(open(exp + f"/host_syncs_rank{r_}.json"))
        assert hs["steps"] == 1, hs


def test_launcher_forced_ddp_world1_gloo_nan_steps(toy_dataset, tmp_path):
    """AUM_FORCE_DDP=1: process group, DistributedDataParallel wrapper, gradient-exchange hook and the MIN-reduced finite flag at world
    size 1 (the form tests/test_gpu_ddp.py runs over RCCL on a one-GPU box), here over gloo on the lane-array build."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = str(tmp_path / "exp_f")
    cmd = [sys.executable, os.path.join(root, "tests", "launcher_worker.py"),
           "--model_type", "tiny", "--depth", "1", "--n_class", "4", "--label-csv", str(toy_dataset / "labels.csv"),
           "--data-train", str(toy_dataset / "train.json"), "--data-val", str(toy_dataset / "val.json"),
           "--audio_length", "64", "--num-workers", "0", "-b", "2", "--mixed_precision", "no", "--exp-dir", exp,
           "--n-epochs", "1", "--metrics", "acc", "--loss", "CE", "--if_nan2num", "False", "--if_continue_inf", "True"]
    env = dict(os.environ, OMP_NUM_THREADS="2", AUM_FORCE_DDP="1", AUM_TEST_NAN_STEPS="1", AUM_TEST_NAN_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("Loss is not finite on some rank, continuing training") == 1, r.stdout[-2000:]
    sd = torch.load(exp + "/models/best_audio_model.pth")
    assert all(k.startswith("module.") for k in sd) and all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    assert json.load(open(exp + "/host_syncs_rank0.json"))["steps"] == 4          # 5 steps, one skipped


def test_tunableop_solution_file_is_seeded_per_rank(monkeypatch, tmp_path):
    """aum.tunable.enable: the recorded GEMM solutions are copied where TunableOp looks for them (file name + device
    ordinal) for this rank's ordinal and for ordinal 0 (ranks masked to one visible device); env defaults are not
    overridden when the user already set them; AUM_NO_TUNABLEOP=1 turns the whole thing off."""
    from aum import tunable
    for k in ("PYTORCH_TUNABLEOP_ENABLED", "PYTORCH_TUNABLEOP_TUNING", "PYTORCH_TUNABLEOP_FILENAME",
              "PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "PYTORCH_TUNABLEOP_VERBOSE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    monkeypatch.setattr(tempfile, "tempdir", None)
    monkeypatch.setenv("AUM_NO_TUNABLEOP", "1")
    assert tunable.enable(3) is None and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ
    monkeypatch.delenv("AUM_NO_TUNABLEOP")
    monkeypatch.setenv("PYTORCH_TUNABLEOP_VERBOSE", "2")
    d = tunable.enable(3)
    src = open(os.path.join(os.path.dirname(tunable.__file__), "tunableop_gfx950.csv")).read()
    assert "GemmTunableOp" in src and "GemmStridedBatchedTunableOp" in src
    for ordinal in (0, 3):
        assert open(os.path.join(d, f"results{ordinal}.csv")).read() == src
    assert os.environ["PYTORCH_TUNABLEOP_FILENAME"] == os.path.join(d, "results.csv")
    assert os.environ["PYTORCH_TUNABLEOP_ENABLED"] == "1" and os.environ["PYTORCH_TUNABLEOP_VERBOSE"] == "2"
    for k in ("PYTORCH_TUNABLEOP_ENABLED", "PYTORCH_TUNABLEOP_TUNING", "PYTORCH_TUNABLEOP_FILENAME",
              "PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"):
        os.environ.pop(k, None)                   # enable() wrote them with setdefault: do not leak into other tests
