"""Drop-in for mamba_ssm.ops.triton.selective_state_update (the reference's Triton kernel + its torch statement, MS:27-29 imports it for
`Mamba.step`): the single-step selective scan of streaming inference on the HIP library (aum_selective_state_update)."""
import torch
import torch.nn.functional as F

import aum_hip


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state (batch, dim, dstate) advanced IN PLACE: exp(dt A) state + dt x B; returns (<state, C> + D x) * silu(z), (batch, dim) in x's dtype
    (selective_state_update.py:157-192 of the reference; a cache that is not fp32 goes through an fp32 copy and is written back)"""
    st = state if state.dtype == torch.float32 and state.is_contiguous() else state.float().contiguous()
    out = aum_hip.state_update(st, x, dt, A, B, C, D, z, dt_bias, dt_softplus)
    if st is not state:
        state.copy_(st)
    return out


def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """the same step in plain torch (own statement of the recurrence; the checker of tests/test_host_package.py)"""
    if dt_bias is not None:
        dt = dt + dt_bias
    dt = F.softplus(dt) if dt_softplus else dt
    decay = torch.exp(dt.unsqueeze(-1) * A)                                   # (batch, dim, dstate)
    drive = (dt * x).unsqueeze(-1) * B.unsqueeze(1)
    state.copy_(state * decay + drive)
    y = (state.to(C.dtype) * C.unsqueeze(1)).sum(-1)
    if D is not None:
        y = y + (x * D).to(y.dtype)
    return (y if z is None else y * F.silu(z)).to(x.dtype)
