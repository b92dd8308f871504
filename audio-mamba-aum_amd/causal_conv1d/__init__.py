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


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """One token of streaming inference (the wheel's function of the same name, call site MS:328-334): x (batch, dim); conv_state (batch, dim,
    width) is shifted left and extended by x IN PLACE; returns act(sum(conv_state * weight, -1) + bias) in x's dtype (aum_causal_conv1d_update;
    a cache that is not fp32 goes through an fp32 copy and is written back)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    st = conv_state if conv_state.dtype == torch.float32 and conv_state.is_contiguous() else conv_state.float().contiguous()
    out = aum_hip.conv1d_update(x, st, weight, bias, activation in ("silu", "swish"))
    if st is not conv_state:
        conv_state.copy_(st)
    return out
