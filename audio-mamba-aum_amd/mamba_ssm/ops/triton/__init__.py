"""Import-path compatibility only: the reference keeps its fused add+norm under mamba_ssm.ops.triton
(/root/reference/src/models/mamba_models.py:26 imports RMSNorm, layer_norm_fn, rms_norm_fn from here).
Nothing in this package uses Triton; the kernels are hand-written HIP (libaum_hip.so)."""
