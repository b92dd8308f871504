"""Drop-in for the reference's mamba_ssm.ops.triton.layernorm
(/root/reference/vim-mamba_ssm/mamba_ssm/ops/triton/layernorm.py = "LN"): RMSNorm, rms_norm_fn, layer_norm_fn,
rms_norm_ref, layer_norm_ref with the same signatures and return conventions.  The fused residual-add + RMSNorm
forward/backward (LN:51-120, 180-290 in Triton upstream) is the hand-written gfx950 kernel pair behind
aum_rmsnorm_fwd / aum_rmsnorm_bwd.  No Triton is involved despite the module path (kept for import parity).
"""
import torch
import torch.nn.functional as F

import aum_hip


def layer_norm_ref(x, weight, bias, residual=None, eps=1e-6, prenorm=False, upcast=False):
    """Pure-PyTorch LayerNorm with optional residual add (contract of LN:19-32)."""
    dtype = x.dtype
    if upcast:
        x, weight = x.float(), weight.float()
        bias = bias.float() if bias is not None else None
        residual = residual.float() if residual is not None else None
    if residual is not None:
        x = (x + residual).to(x.dtype)
    out = F.layer_norm(x.to(weight.dtype), x.shape[-1:], weight=weight, bias=bias, eps=eps).to(dtype)
    return (out, x) if prenorm else out


def rms_norm_ref(x, weight, bias, residual=None, eps=1e-6, prenorm=False, upcast=False):
    """Pure-PyTorch RMSNorm with optional residual add (contract of LN:35-48)."""
    dtype = x.dtype
    if upcast:
        x, weight = x.float(), weight.float()
        bias = bias.float() if bias is not None else None
        residual = residual.float() if residual is not None else None
    if residual is not None:
        x = (x + residual).to(x.dtype)
    inv = torch.rsqrt(x.square().mean(dim=-1, keepdim=True) + eps)
    out = x * inv * weight
    if bias is not None:
        out = out + bias
    out = out.to(dtype)
    return (out, x) if prenorm else out


class LayerNormFn(torch.autograd.Function):
    """LN:380-461.  is_rms_norm=True is the HIP kernel pair; plain LayerNorm (never reached by AuM: rms_norm=True,
    MM:206) is composed from torch ops so the public function keeps working."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                is_rms_norm=False):
        if not is_rms_norm or bias is not None:
            raise NotImplementedError("HIP fused add+norm implements RMSNorm without bias (what AuM uses, MM:77-97)")
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            assert residual.shape == shape
            r2 = residual.reshape(-1, shape[-1])
            if r2.stride(-1) != 1:
                r2 = r2.contiguous()
        res_dtype = r2.dtype if r2 is not None else (torch.float32 if residual_in_fp32 else None)
        y, rstd, res_out = aum_hip.rmsnorm_fwd(x2, weight, r2, eps, residual_dtype=res_dtype)
        ctx.save_for_backward(res_out, weight, rstd)
        ctx.shape, ctx.has_residual, ctx.prenorm, ctx.x_dtype = shape, residual is not None, prenorm, x.dtype
        ctx.wparam = weight
        y = y.reshape(shape)
        return (y, res_out.reshape(shape)) if prenorm else y

    @staticmethod
    def backward(ctx, dy, *args):
        x, weight, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        dres = None
        if ctx.prenorm and args[0] is not None:
            dres = args[0].reshape(-1, dy.shape[-1])
            if dres.stride(-1) != 1:
                dres = dres.contiguous()
            dres = dres.to(x.dtype)
        from mamba_ssm.ops.selective_scan_interface import _homed, grad_home
        home = grad_home(ctx.wparam)          # the weight gradient's place in a DistributedDataParallel bucket, if the parameter has one
        dx, dw, dres_in = aum_hip.rmsnorm_bwd(dy2.to(ctx.x_dtype), x, weight, rstd, dres, ctx.has_residual,
                                              x_dtype=ctx.x_dtype, dw_out=home)
        return (dx.reshape(ctx.shape), _homed(dw.to(weight.dtype), home), None,
                dres_in.reshape(ctx.shape) if ctx.has_residual else None, None, None, None, None)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    if is_rms_norm and bias is None:
        return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)
    # plain LayerNorm / biased RMSNorm: off the AuM path; torch composition with the reference's return convention
    res_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else x.dtype)
    pre = x.to(res_dtype) if residual is None else (x.float() + residual.float()).to(res_dtype)
    ref = rms_norm_ref if is_rms_norm else layer_norm_ref
    out = ref(pre.float(), weight.float(), bias.float() if bias is not None else None, eps=eps).to(x.dtype)
    return (out, pre) if prenorm else out


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return layer_norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)


class RMSNorm(torch.nn.Module):
    """LN:481-502: .weight (ones), .bias = None, .eps; forward(x, residual=None, prenorm=False, residual_in_fp32=False)."""

    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)
