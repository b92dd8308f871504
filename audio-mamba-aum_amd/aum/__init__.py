"""Minimal AuM assembly + training harness used by bench.py / tests on top of the drop-in `mamba_ssm` package.
The reference's own src/models/mamba_models.py runs unmodified on that package; this module exists because nothing
Python from the reference travels to the GPU box."""
def __getattr__(name):          # lazy: `from aum import tunable` must not import torch
    if name in ("AudioMamba", "AUM_SIZES", "build_aum"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
