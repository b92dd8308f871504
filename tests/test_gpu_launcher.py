"""SURVEY 8(f1-f3) on the product library: the launcher's training loop, checkpoint re-gridding and the augmentation fused into the
log-mel kernel, on an MI355X, against the reference's own numbers (tests/golden/launcher.npz) -- the same checks the CPU suite runs
on the lane-array build (tests/launcher_checks.py)."""
import pytest
import torch

import aum_hip
import launcher_checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def product_library():
    assert torch.cuda.is_available(), "these tests need a GPU"
    lib = aum_hip.get()          # ImportError if the extension is missing: no fallback
    assert not lib.host
    return lib


def test_training_loop_matches_reference_traintest_gpu(tmp_path, monkeypatch):
    """f1: aum.train.train on the GPU (HIP frontend stubs aside, every block kernel is the product's) reproduces the reference's
    traintest.train: Adam hyper-parameters, the learning rate of every step, result.csv, predictions, parameters after 8 steps"""
    launcher_checks.check_training_loop(tmp_path, monkeypatch)


def test_checkpoint_regrid_matches_reference_constructor_gpu():
    """f2: a 256-frame, 7-class checkpoint loaded into a 512-frame, 3-class model: re-gridded position embedding and the loaded
    model's logits computed by the HIP kernels"""
    launcher_checks.check_checkpoint_regrid("cuda")


def test_fused_augmentation_matches_separate_ops_gpu():
    """f3: SpecAug bands, noise and roll inside aum_fbank_fwd's store against the separate torch ops after the kernel"""
    launcher_checks.check_fused_augmentation("cuda")


def test_fused_augmentation_matches_independent_oracle_gpu():
    """f3 against oracle/augment.py (numpy restatement of DL:206-228 + torchaudio's mask_along_axis rule) for fixed raw draws"""
    launcher_checks.check_fused_augmentation_vs_oracle("cuda")
