"""Minimal AuM assembly + training harness used by bench.py / tests on top of the drop-in `mamba_ssm` package.
The reference's own src/models/mamba_models.py runs unmodified on that package; this module exists because nothing
Python from the reference travels to the GPU box."""
from .model import AudioMamba, AUM_SIZES, build_aum  # noqa: F401
