"""Outer drop-in boundary: the reference's OWN `src/models/mamba_models.py` (MM:18,26,126 import
`mamba_ssm.modules.mamba_simple.Mamba` and `mamba_ssm.ops.triton.layernorm.{RMSNorm, layer_norm_fn, rms_norm_fn}`),
loaded UNMODIFIED from /root/reference on top of THIS repo's `mamba_ssm` package, must reproduce the goldens that the
same file produced on top of the reference's `*_ref` arithmetic (tests/golden/model.npz).

Build-container test only: nothing of the reference travels, so it is skipped wherever /root/reference is absent (the
GPU box).  The kernels underneath are the lane-array build of the kernel sources (tests/emu), as in
test_host_package.py.  timm / wget are not installed in the image: init-only stand-ins (`to_2tuple`, `DropPath` =
identity at drop_path 0, two initialisers -- every parameter is overwritten by the seeded state dict afterwards)."""
import contextlib
import importlib.machinery
import importlib.util
import io
import os
import sys
import types

import pytest
import torch

import aum_hip
import cases
from conftest import load_golden, rel_err

REF = "/root/reference"
MM_FILE = os.path.join(REF, "src", "models", "mamba_models.py")
pytestmark = pytest.mark.skipif(not os.path.isfile(MM_FILE), reason="reference checkout not present (build container only)")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def reference_mm():
    import build_emu
    import torch.nn as nn
    old_product = aum_hip._product
    aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    saved = {k: sys.modules.get(k) for k in ("timm", "timm.models", "timm.models.layers", "wget", "src", "src.models",
                                             "src.utilities", "src.models.mamba_models")}
    timm, tm, tl = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.layers"))

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.DropPath = DropPath
    tl.trunc_normal_ = lambda t, std=0.02, **kw: nn.init.trunc_normal_(t, std=std)
    tl.lecun_normal_ = lambda t: nn.init.trunc_normal_(t, std=(1.0 / t[0].numel()) ** 0.5)
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "wget": types.ModuleType("wget")})
    for name, path in (("src", REF + "/src"), ("src.models", REF + "/src/models"), ("src.utilities", REF + "/src/utilities")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
    import mamba_ssm
    assert os.path.realpath(mamba_ssm.__file__).startswith(os.path.realpath(os.path.join(os.path.dirname(__file__), ".."))), \
        "the reference file must sit on THIS repo's mamba_ssm, not the reference's"
    spec = importlib.util.spec_from_file_location("src.models.mamba_models", MM_FILE)
    mm = importlib.util.module_from_spec(spec)
    sys.modules["src.models.mamba_models"] = mm
    spec.loader.exec_module(mm)          # the reference source, byte for byte
    yield mm
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    for k in [k for k in sys.modules if k.startswith("src.utilities.")]:
        sys.modules.pop(k, None)
    aum_hip._product = old_product


def test_reference_file_binds_this_package(reference_mm):
    import mamba_ssm.modules.mamba_simple as ms
    import mamba_ssm.ops.triton.layernorm as ln
    assert reference_mm.Mamba is ms.Mamba                         # MM:18
    assert reference_mm.RMSNorm is ln.RMSNorm                     # MM:26 (isinstance checks at MM:54-56,77,648)
    assert reference_mm.rms_norm_fn is ln.rms_norm_fn and reference_mm.layer_norm_fn is ln.layer_norm_fn


@pytest.mark.parametrize("case", cases.MODEL_CASES, ids=lambda c: c[0])
def test_reference_audio_mamba_runs_unmodified_on_the_package(reference_mm, case):
    g = load_golden("model")
    name, btype, depth, dim, spec, ncls, batch = case[:7]
    if depth > 4:
        pytest.skip("deep goldens are covered on the GPU (lane-array build is slow)")
    with contextlib.redirect_stdout(io.StringIO()):
        model = reference_mm.AudioMamba(spectrogram_size=spec, depth=depth, embed_dim=dim, num_classes=ncls,
                                        bimamba_type=btype, **cases.model_kwargs(case))
    sd = model.state_dict()
    assert sorted(sd.keys()) == list(g[name + ".keys"])
    vals = cases.model_state({k: tuple(v.shape) for k, v in sd.items()}, name)
    model.load_state_dict({k: torch.tensor(v) for k, v in vals.items()})
    d = cases.model_inputs(*case)
    logits = model(torch.tensor(d["x"]))
    (logits * torch.tensor(d["dlogits"])).sum().backward()
    assert rel_err(logits.detach().numpy(), g[name + ".logits"]) < 1e-4
    for k, p_ in model.named_parameters():
        gn = float(p_.grad.double().norm().item())
        ref = float(g[f"{name}.gnorm.{k}"])
        assert abs(gn - ref) <= 1e-3 * max(ref, 1e-6), (k, gn, ref)
        if f"{name}.grad.{k}" in g:
            assert rel_err(p_.grad.numpy(), g[f"{name}.grad.{k}"]) < 1e-3, k
