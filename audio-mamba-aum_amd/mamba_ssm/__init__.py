"""Drop-in `mamba_ssm` package for kaistmm/Audio-Mamba-AuM on MI355X (gfx950).

Same import surface as the reference's overlay (/root/reference/vim-mamba_ssm/mamba_ssm/__init__.py:1-5):
selective_scan_fn, mamba_inner_fn, bimamba_inner_fn, Mamba.  MambaLMHeadModel (language-model wrapper, unused by
AuM and out of scope) resolves lazily to an explanatory error instead of importing transformers.
"""
__version__ = "1.1.1"

from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, mamba_inner_fn, bimamba_inner_fn  # noqa: E402,F401
from mamba_ssm.modules.mamba_simple import Mamba  # noqa: E402,F401


def __getattr__(name):
    if name == "MambaLMHeadModel":
        raise AttributeError("MambaLMHeadModel (LM wrapper) is out of scope for the AuM hot path on this build")
    raise AttributeError(name)
