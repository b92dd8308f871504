"""Drop-in for the reference's mamba_ssm.modules.mamba_simple.Mamba
(/root/reference/vim-mamba_ssm/mamba_ssm/modules/mamba_simple.py = "MS"): same constructor signature (MS:35-56),
parameter / sub-module names (the checkpoint contract: in_proj, conv1d, x_proj, dt_proj, out_proj, A_log, D, A_b_log,
and for v2 conv1d_b, x_proj_b, dt_proj_b, D_b), initialisation (MS:94-127) and forward dispatch (MS:169-311).

Differences: the kernels underneath are libaum_hip.so; Bi-Bi (v2) passes reverse=True instead of flipping xz and the
result (MS:229-246); single-token decoding (`step`, inference caches, MS:313-400) exists for the causal block
(bimamba_type="none") as the reference's element-wise composition -- AuM itself never passes inference_params
(MM:620-622).
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from mamba_ssm.ops.selective_scan_interface import (InProjFn, bimamba_inner_fn, mamba_inner_fn, mamba_inner_fn_no_out_proj,
                                                    neg_exp, selective_scan_fn)
import mamba_ssm.ops.selective_scan_interface as ssi
from causal_conv1d import causal_conv1d_fn, causal_conv1d_update
from mamba_ssm.ops.triton.selective_state_update import selective_state_update


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None, bimamba_type="none", if_devide_out=False,
                 init_layer_scale=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx = use_fast_path, layer_idx
        self.bimamba_type, self.if_devide_out = bimamba_type, if_devide_out
        self.init_layer_scale = init_layer_scale
        if init_layer_scale is not None:
            self.gamma = nn.Parameter(init_layer_scale * torch.ones(d_model), requires_grad=True)

        self.in_proj = nn.Linear(d_model, self.d_inner * 2, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1,
                                bias=conv_bias, **fk)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        self._init_dt(self.dt_proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk)
        self.A_log = self._make_A_log(device)
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))          # fp32 skip, MS:126
        self.D._no_weight_decay = True

        if bimamba_type in ("v1", "v2"):
            self.A_b_log = self._make_A_log(device)                              # MS:130-147
        if bimamba_type == "v2":                                                 # MS:149-165
            self.conv1d_b = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1,
                                      bias=conv_bias, **fk)
            self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
            self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
            self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
            self.D_b._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

    # ---- initialisers (MS:94-123) ----------------------------------------------------------------
    def _init_dt(self, dt_proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk):
        std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(dt_proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(dt_proj.weight, -std, std)
        else:
            raise NotImplementedError
        # bias such that softplus(bias) is log-uniform in [dt_min, dt_max]
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        dt_proj.bias._no_reinit = True

    def _make_A_log(self, device):
        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1)
        p = nn.Parameter(torch.log(A))       # S4D-real, kept in fp32
        p._no_weight_decay = True
        return p

    # ---- forward (MS:169-311) ---------------------------------------------------------------------
    def forward(self, hidden_states, inference_params=None, *, time_reversed=False):
        """MS:169.  `time_reversed` (extension, keyword only): the block applied to the time-reversed sequence and reversed
        back, flip(forward(flip(h))), without the copies -- the odd layers of an `if_bidirectional` model (MM:623-638): every stage but
        the conv and the scan is token-wise, and those two take a direction flag."""
        conv_state = ssm_state = None
        if time_reversed and (inference_params is not None or not self.use_fast_path):
            return self.forward(hidden_states.flip([1]), inference_params).flip([1])
        if inference_params is not None:                                   # streaming inference, MS:176-182
            if self.bimamba_type != "none":
                raise NotImplementedError("inference caches only make sense for the causal (bimamba_type='none') block")
            conv_state, ssm_state = self._get_states_from_cache(inference_params, hidden_states.shape[0])
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(hidden_states, conv_state, ssm_state)    # states updated in place
                return out
        batch, seqlen, _ = hidden_states.shape
        tm = (ssi.TOKEN_MAJOR and self.use_fast_path and inference_params is None
              and not (ssi._REF_DZ_DROP and self.bimamba_type == "v1")      # that option lives in the channel-major block
              and ssi.token_major_preferred(batch, self.d_inner, self.bimamba_type != "none", seqlen=seqlen)
              and ssi.token_major_ok(self.d_inner, self.d_state, self.d_conv, self.dt_rank,
                                     torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else hidden_states.dtype))
        if tm:
            # token-major block: xz rows [x | z] as the GEMM writes them; (B, 2E, L) is the transposed VIEW the interface expects
            xz = ssi.InProjTmFn.apply(self.in_proj.weight, hidden_states.reshape(batch * seqlen, -1))
            xz = xz.view(batch, seqlen, -1).transpose(1, 2)
        else:
            # matmul + transpose in one GEMM: xz is (B, 2E, L) stored channel-major, like MS:185-189
            xz = InProjFn.apply(self.in_proj.weight, hidden_states.reshape(batch * seqlen, -1))
            xz = xz.reshape(-1, batch, seqlen).permute(1, 0, 2)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        A = neg_exp(self.A_log)
        if self.use_fast_path and inference_params is None:                # MS:190
            if self.bimamba_type == "v1":
                A_b = neg_exp(self.A_b_log)
                out = bimamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                       self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias, A, A_b, None,
                                       None, self.D.float(), delta_bias=self.dt_proj.bias.float(),
                                       delta_softplus=True, reverse=time_reversed)
            elif self.bimamba_type == "v2":
                A_b = neg_exp(self.A_b_log)
                # Bi-Bi's two pipelines are independent until their outputs are added; each is one-direction launches of 1 536 waves at
                # the bench shape (half of what the time-serial kernels hold), so the two run next to each other on two side streams
                # (forward here; autograd runs each pipeline's backward on the stream its forward ran on; why BOTH leave the calling
                # stream: ssi.side_streams).  AUM_V2_STREAMS=0: in line.
                two = tm and xz.is_cuda and ssi.v2_two_streams(tuple(self.parameters()), module=self)
                if two:
                    main = torch.cuda.current_stream(xz.device)
                    s_f, s_b = ssi.side_streams(xz.device)
                    s_f.wait_stream(main)
                    s_b.wait_stream(main)
                    for t_ in (xz, A, A_b):
                        t_.record_stream(s_f)
                        t_.record_stream(s_b)
                with (torch.cuda.stream(s_f) if two else contextlib.nullcontext()):
                    out_f = mamba_inner_fn_no_out_proj(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                                       self.dt_proj.weight, A, None, None, self.D.float(),
                                                       delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                                                       reverse=time_reversed)
                with (torch.cuda.stream(s_b) if two else contextlib.nullcontext()):
                    out_b = mamba_inner_fn_no_out_proj(xz, self.conv1d_b.weight, self.conv1d_b.bias,
                                                       self.x_proj_b.weight, self.dt_proj_b.weight, A_b, None, None,
                                                       self.D_b.float(), delta_bias=self.dt_proj_b.bias.float(),
                                                       delta_softplus=True, reverse=not time_reversed)
                if two:
                    main.wait_stream(s_f)
                    main.wait_stream(s_b)
                    out_f.record_stream(main)          # allocated on the side streams, consumed on the calling one
                    out_b.record_stream(main)
                y = out_f + out_b                                          # (B, E, L) logical, both in xz's storage order
                if self.if_devide_out:
                    y = y / 2
                E = y.shape[1]
                if tm:
                    # token-major pipelines: y is the transposed view of (B, L, E) rows -- out_proj on the rows (SSI:517 dispatch)
                    out = ssi.OutProjTmFn.apply(self.out_proj.weight, y.transpose(1, 2).reshape(batch * seqlen, E))
                else:
                    y2 = y.permute(1, 0, 2).reshape(E, batch * seqlen)
                    out = torch.matmul(y2.t(), self.out_proj.weight.t().to(y2.dtype))
                if self.out_proj.bias is not None:
                    out = out + self.out_proj.bias.to(out.dtype)
                out = out.reshape(batch, seqlen, -1)
            else:
                out = mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                     self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias, A, None, None,
                                     self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                                     reverse=time_reversed)
        else:       # un-fused composition of the same ops (MS:264-308)
            x, z = xz.chunk(2, dim=1)
            if conv_state is not None:                                     # MS:268-271: the last d_conv inputs
                conv_state.copy_(F.pad(x, (self.d_conv - x.shape[-1], 0)))
            x = causal_conv1d_fn(x, self.conv1d.weight.reshape(self.d_inner, -1), self.conv1d.bias, self.activation)
            x_dbl = self.x_proj(x.transpose(1, 2).reshape(batch * seqlen, -1))
            dt, Bm, Cm = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
            dt = (self.dt_proj.weight.to(dt.dtype) @ dt.t()).reshape(-1, batch, seqlen).permute(1, 0, 2)
            Bm = Bm.reshape(batch, seqlen, -1).transpose(1, 2).contiguous()
            Cm = Cm.reshape(batch, seqlen, -1).transpose(1, 2).contiguous()
            y = selective_scan_fn(x, dt.to(x.dtype), A, Bm.to(x.dtype), Cm.to(x.dtype), self.D.float(),
                                  z=z.to(x.dtype), delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                                  return_last_state=ssm_state is not None)
            if ssm_state is not None:                                      # MS:300-302
                y, last_state = y
                ssm_state.copy_(last_state)
            out = self.out_proj(y.transpose(1, 2))
        if self.init_layer_scale is not None:
            out = out * self.gamma
        return out

    def step(self, hidden_states, conv_state, ssm_state):
        """One token of streaming inference for the causal block (MS:313-358): (batch, 1, d_model) in, (batch, 1, d_model) out, the caches
        conv_state (batch, d_inner, d_conv) and ssm_state (batch, d_inner, d_state) advanced in place -- the block's four stages on a
        sequence of length one, the two recurrent ones on the library's per-token kernels (round 4):
            xc = causal_conv1d_update(x, conv_state, w, b, silu)                 window <- last d_conv inputs; silu(<window, w> + b)
            (dt, B, C) = x_proj(xc);   dt = dt W_dt^T                            (bias and softplus inside the state update, MS:340)
            y = selective_state_update(ssm_state, xc, dt, A, B, C, D, z, dt_bias, softplus)
            out = out_proj(y)"""
        if hidden_states.dim() != 3 or hidden_states.shape[1] != 1:
            raise ValueError("step() advances the caches by exactly one token: hidden_states must be (batch, 1, d_model)")
        E, N, R = self.d_inner, self.d_state, self.dt_rank
        x_new, z = self.in_proj(hidden_states[:, 0]).split(E, dim=-1)
        xc = causal_conv1d_update(x_new, conv_state, self.conv1d.weight.view(E, self.d_conv), self.conv1d.bias, self.activation)
        proj = self.x_proj(xc)
        dt_in, B_t, C_t = proj[:, :R], proj[:, R:R + N], proj[:, R + N:R + 2 * N]
        dt = F.linear(dt_in, self.dt_proj.weight)                                   # the bias is not added here (MS:340)
        y = selective_state_update(ssm_state, xc, dt, -torch.exp(self.A_log.float()), B_t, C_t, self.D, z=z, dt_bias=self.dt_proj.bias,
                                   dt_softplus=True)
        return self.out_proj(y).unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        """MS:360-373"""
        device = self.out_proj.weight.device
        conv_state = torch.zeros(batch_size, self.d_inner, self.d_conv, device=device,
                                 dtype=self.conv1d.weight.dtype if dtype is None else dtype)
        ssm_state = torch.zeros(batch_size, self.d_inner, self.d_state, device=device,
                                dtype=self.dt_proj.weight.dtype if dtype is None else dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        """MS:375-400"""
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            inference_params.key_value_memory_dict[self.layer_idx] = self.allocate_inference_cache(batch_size, 0)
        conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
        if initialize_states:
            conv_state.zero_()
            ssm_state.zero_()
        return conv_state, ssm_state
