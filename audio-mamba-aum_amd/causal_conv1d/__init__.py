"""Drop-in for the `causal_conv1d` wheel's public function used by the reference
(`from causal_conv1d import causal_conv1d_fn`, SSI:9, MS:14): depthwise causal conv + optional SiLU, HIP-backed."""
import torch

import aum_hip

__version__ = "1.1.3.post1+aum.gfx950"


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in (None, "silu", "swish"):
            raise NotImplementedError("activation must be None, silu, or swish")
        if x.stride(-1) != 1:
            x = x.contiguous()
        ctx.silu = activation in ("silu", "swish")
        ctx.save_for_backward(x, weight, bias)
        return aum_hip.conv1d_fwd(x, weight, bias, ctx.silu)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        dx, dw, db = aum_hip.conv1d_bwd(x, weight, bias, dout.to(x.dtype), ctx.silu)
        return dx, dw.to(weight.dtype).reshape(weight.shape), (db.to(bias.dtype) if bias is not None else None), None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim,); activation: None | "silu" | "swish"."""
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_update(*args, **kwargs):
    raise NotImplementedError("single-token decode (Mamba.step) is out of scope: AuM never decodes (SURVEY 2.1 #2)")
