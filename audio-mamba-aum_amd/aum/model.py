"""AuM (Audio Mamba) assembled from the drop-in mamba_ssm package: patch embed -> middle cls token -> abs pos embed
-> depth x [fused add+RMSNorm -> Mamba mixer] -> final fused add+RMSNorm -> cls -> linear head.

Mirrors the default configuration of the reference's AudioMamba (/root/reference/src/models/mamba_models.py = "MM":
forward_features MM:509-667, Block.forward MM:58-99, defaults MM:191-242) and produces the SAME state-dict keys
(patch_embed.proj.*, cls_token, pos_embed.pos_embed, layers.{i}.mixer.*, layers.{i}.norm.weight, norm_f.weight,
head.*) so published checkpoints load.  Also mirrored: the cls-token placements of RUN's flags (middle / end / head,
MM:528-535), `transpose_token_sequence` (time-major token order, MM:545-566) and `if_bidirectional` layer pairing
(MM:623-638).  Options off that surface (rope, double cls, flexible patch sizes, drop-path > 0) are rejected.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from mamba_ssm.modules.mamba_simple import Mamba
from mamba_ssm.ops.selective_scan_interface import _autocast_dtype, step_cache
from mamba_ssm.ops.triton.layernorm import RMSNorm, rms_norm_fn

AUM_SIZES = {"base": 768, "small": 384, "tiny": 192}      # embed dims; depth 24 for all three (RUN:227-237)


class PatchEmbed(nn.Module):
    """FlexiPatchEmbed default branch (TOK:278-310): conv2d(kernel=stride=patch) -> flatten -> (B, N, Dm)."""

    def __init__(self, patch_size, strides, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=strides)
        fan_in = in_chans * patch_size[0] * patch_size[1]
        nn.init.trunc_normal_(self.proj.weight, std=math.sqrt(1.0 / fan_in) / 0.87962566103423978)   # lecun_normal_
        nn.init.zeros_(self.proj.bias)

    def forward(self, x):
        # non-overlapping patches: the conv is a [B*N, C*ph*pw] x [C*ph*pw, Dm] GEMM (no MIOpen find/autotune in the loop)
        Bsz, C, Fd, Td = x.shape
        ph, pw = self.proj.kernel_size
        nf, nt = Fd // ph, Td // pw
        cols = x[:, :, :nf * ph, :nt * pw].reshape(Bsz, C, nf, ph, nt, pw).permute(0, 2, 4, 1, 3, 5)
        cols = cols.reshape(Bsz * nf * nt, C * ph * pw)
        out = F.linear(cols, self.proj.weight.reshape(self.proj.out_channels, -1), self.proj.bias)
        return out.reshape(Bsz, nf * nt, -1)


class FrontendTokensFn(torch.autograd.Function):
    """Waveform -> (B, N+1, Dm) token sequence in one launch (aum_frontend_tokens_fwd): log-mel frames, patch GEMM, bias,
    position rows and the cls row, with the arithmetic of the autocast reference (16-bit conv output, fp32 position add;
    DL:134-147 -> TOK:278-310 -> MM:509-541).  Backward: the patch matrix saved by the kernel gives the weight gradient."""

    @staticmethod
    def forward(ctx, wave, weight, bias, pos_embed, cls_token, fe, dtype, cls_pos, time_major):
        import aum_hip
        Dm = weight.shape[0]
        w16 = weight.detach().reshape(Dm, -1).to(dtype).contiguous()
        pe = pos_embed.detach().float()
        cls_row = (cls_token.detach().float() + pe[:, :1]).reshape(Dm).contiguous()
        need = any(ctx.needs_input_grad[1:5])
        tokens, patches = aum_hip.frontend_tokens(
            wave, fe.tables.tables, fe.target_length, fe.norm_mean, fe.norm_std, w16, bias.detach().float().contiguous(),
            pe[0, 1:].contiguous(), cls_row, cls_pos, time_major=time_major, save_patches=need, aug=fe.aug, noise=fe.noise)
        ctx.save_for_backward(patches)
        ctx.cfg = (weight.shape, weight.dtype, dtype, cls_pos, time_major, fe.target_length // 16)
        return tokens

    @staticmethod
    def backward(ctx, g):
        (patches,) = ctx.saved_tensors
        wshape, wdtype, dtype, cls_pos, time_major, nt = ctx.cfg
        Bsz, _, Dm = g.shape
        gp = torch.cat((g[:, :cls_pos], g[:, cls_pos + 1:]), dim=1)              # (B, N, Dm) in sequence order
        if time_major:                                                          # back to the cell order f * n_t + t
            gp = gp.reshape(Bsz, nt, -1, Dm).transpose(1, 2).reshape(Bsz, -1, Dm)
        g16 = gp.to(dtype).reshape(-1, Dm)                                      # gradient of the 16-bit conv output
        import aum_hip
        if g16.is_cuda and aum_hip.gemm_wgrad_supported(g16, patches):          # 3 output tiles x 64 token splits instead of a one-round library GEMM over 32 768 tokens
            dweight = aum_hip.gemm_wgrad(g16, patches).to(wdtype).reshape(wshape)
        else:
            dweight = (g16.t() @ patches).to(wdtype).reshape(wshape)
        dbias = g16.sum(0).to(wdtype)
        dcls = g[:, cls_pos].sum(0)
        dpos = torch.cat((dcls[None], gp.sum(0)), dim=0)[None]
        return None, dweight, dbias, dpos.to(wdtype), dcls.reshape(1, 1, Dm).to(wdtype), None, None, None, None


class PosEmbed(nn.Module):
    """FlexiPosEmbed with pos_embed_prefix=True (TOK:330-451): row 0 belongs to the cls token, rows 1.. to the patches."""

    def __init__(self, n_tokens, embed_dim):
        super().__init__()
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tokens, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)


class Block(nn.Module):
    """MM:30-99 with fused_add_norm=True, residual_in_fp32=True, drop_path=0."""

    def __init__(self, dim, mixer, eps):
        super().__init__()
        self.mixer = mixer
        self.norm = RMSNorm(dim, eps=eps)

    def forward(self, hidden_states, residual=None, inference_params=None, *, time_reversed=False):
        """MM:58-99 (the third positional slot is the reference's inference_params).  time_reversed (keyword only): the block on the
        time-reversed sequence, reversed back (the odd layers of `if_bidirectional`) -- add + norm is token-wise, the mixer takes the
        flag: no flipped copies of hidden_states / residual."""
        hidden_states, residual = rms_norm_fn(hidden_states, self.norm.weight, self.norm.bias, residual=residual,
                                              prenorm=True, residual_in_fp32=True, eps=self.norm.eps)
        if time_reversed:
            return self.mixer(hidden_states, inference_params=inference_params, time_reversed=True), residual
        return self.mixer(hidden_states, inference_params=inference_params), residual


class AudioMamba(nn.Module):
    def __init__(self, spectrogram_size=(128, 1024), patch_size=(16, 16), strides=(16, 16), depth=24, embed_dim=768,
                 channels=1, num_classes=527, norm_epsilon=1e-5, bimamba_type="v1", if_devide_out=True,
                 use_middle_cls_token=True, use_end_cls_token=False, transpose_token_sequence=False,
                 if_bidirectional=False, ssm_cfg=None, device=None, dtype=None, **unsupported):
        super().__init__()
        on = {k: v for k, v in unsupported.items() if v not in (None, False, 0, 0.0, -1.0)
              and k not in ("if_cls_token", "rms_norm", "fused_add_norm", "residual_in_fp32", "if_abs_pos_embed", "final_pool_type",
                            "imagenet_load_middle_cls_token", "use_PI_for_patch_embed", "imagenet_pretrain_modelkey")}
        if on:
            raise NotImplementedError(f"AudioMamba options outside the accelerated path: {sorted(on)}")
        if tuple(patch_size) != tuple(strides):
            raise NotImplementedError("overlapping patches are off the default path")
        self.embed_dim = self.d_model = embed_dim
        self.num_classes = num_classes
        self.use_middle_cls_token = use_middle_cls_token
        self.use_end_cls_token = use_end_cls_token
        self.transpose_token_sequence = transpose_token_sequence
        self.if_bidirectional = if_bidirectional
        if if_bidirectional and depth % 2:
            raise ValueError("if_bidirectional pairs the layers: depth must be even")
        fdim = (spectrogram_size[0] - patch_size[0]) // strides[0] + 1
        tdim = (spectrogram_size[1] - patch_size[1]) // strides[1] + 1
        self.patch_grid_size = (fdim, tdim)
        self.num_patches = fdim * tdim
        self.num_tokens = 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.layers = nn.ModuleList([
            Block(embed_dim, Mamba(embed_dim, layer_idx=i, bimamba_type=bimamba_type, if_devide_out=if_devide_out,
                                   **(ssm_cfg or {})), norm_epsilon)
            for i in range(depth)])
        self.norm_f = RMSNorm(embed_dim, eps=norm_epsilon)
        if isinstance(self.head, nn.Linear):                       # segm_init_weights, MM:179-184
            nn.init.trunc_normal_(self.head.weight, std=0.02)
            nn.init.zeros_(self.head.bias)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(lambda m: _init_weights(m, depth))              # MM:146-176
        self.patch_embed = PatchEmbed(tuple(patch_size), tuple(strides), channels, embed_dim)
        self.pos_embed = PosEmbed(self.num_patches + self.num_tokens, embed_dim)
        if device is not None or dtype is not None:
            self.to(device=device, dtype=dtype)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def tokens(self, x):
        """(B, T, F) spectrogram -> (B, N+1, Dm) token sequence with the cls token in the middle (MM:509-541)."""
        x = x.unsqueeze(1).transpose(2, 3)                         # B x 1 x F x T
        x = self.patch_embed(x)                                    # token index = f * n_t + t
        Bsz, Np, _ = x.shape
        pe = self.pos_embed.pos_embed
        pos = Np // 2 if self.use_middle_cls_token else (Np if self.use_end_cls_token else 0)      # MM:528-535
        cls = (self.cls_token + pe[:, :1]).expand(Bsz, -1, -1)
        x = x + pe[:, 1:]                                          # position embedding belongs to the (f, t) cell
        if self.transpose_token_sequence:                          # MM:545-566: patches in time-major order, cls stays put
            nf, nt = self.patch_grid_size
            x = x.reshape(Bsz, nf, nt, -1).transpose(1, 2).reshape(Bsz, nt * nf, -1)
        return torch.cat((x[:, :pos], cls.to(x.dtype), x[:, pos:]), dim=1), pos

    def tokens_from_wave(self, wave, fe):
        """(B, n_samples) mean-removed waveform + aum.frontend.WaveInput -> the same token sequence as tokens(spectrogram), in one
        launch when the configuration is the 16 kHz / 128-mel / 16 x 16 autocast one; otherwise log-mel kernel + tokens()."""
        import aum_hip
        dtype = torch.get_autocast_dtype(wave.device.type) if torch.is_autocast_enabled(wave.device.type) else None
        ps = self.patch_embed.proj
        if (dtype is None or tuple(ps.kernel_size) != (16, 16) or ps.in_channels != 1 or self.patch_grid_size[0] != 8
                or self.patch_grid_size[1] * 16 != fe.target_length
                or not aum_hip.frontend_tokens_supported(fe.tables.tables, fe.target_length, self.embed_dim, dtype)):
            return self.tokens(fe.spectrogram(wave))
        Np = self.num_patches
        pos = Np // 2 if self.use_middle_cls_token else (Np if self.use_end_cls_token else 0)
        tok = FrontendTokensFn.apply(wave, ps.weight, ps.bias, self.pos_embed.pos_embed, self.cls_token, fe, dtype, pos,
                                     self.transpose_token_sequence)
        return tok, pos

    def forward_features(self, x, frontend=None):
        hidden, pos = self.tokens(x) if frontend is None else self.tokens_from_wave(x, frontend)
        with step_cache([layer.mixer for layer in self.layers], _autocast_dtype()):     # all blocks' 16-bit weights and A in a few launches
            hidden, residual = self._run_layers(hidden)
        # MM:646-657, then the cls row (MM:660-680).  add + RMSNorm is row-wise and only the cls row is read: the final norm runs on
        # those `batch` rows instead of on all 513 per clip (same values, same gradients; 0.1 ms per step at the bench shape)
        return rms_norm_fn(hidden[:, pos], self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps,
                           residual=None if residual is None else residual[:, pos], prenorm=False, residual_in_fp32=True)

    def _run_layers(self, hidden):
        residual = None
        if not self.if_bidirectional:
            for layer in self.layers:
                hidden, residual = layer(hidden, residual)
        else:                                                      # MM:623-638: layer 2i forward, layer 2i+1 on the flipped sequence
            for i in range(len(self.layers) // 2):
                # flip(layer(flip(h), flip(r))) == layer(h, r, time_reversed=True): the four flipped copies per pair are direction
                # flags on the conv and scan kernels
                hf, rf = self.layers[2 * i](hidden, residual)
                hb, rb = self.layers[2 * i + 1](hidden, residual, time_reversed=True)
                hidden, residual = hf + hb, rf + rb
        return hidden, residual

    def forward(self, x, return_features=False, frontend=None):
        """x: (B, T, F) normalised log-mel spectrogram, or -- with frontend=aum.frontend.WaveInput -- (B, n_samples) waveform"""
        f = self.forward_features(x, frontend)
        return f if return_features else self.head(f)


def _init_weights(module, n_layer):
    """MM:146-176: zero Linear biases (unless _no_reinit), rescale out_proj by 1/sqrt(n_layer)."""
    if isinstance(module, nn.Linear) and module.bias is not None and not getattr(module.bias, "_no_reinit", False):
        nn.init.zeros_(module.bias)
    if isinstance(module, Mamba):
        nn.init.kaiming_uniform_(module.out_proj.weight, a=math.sqrt(5))
        with torch.no_grad():
            module.out_proj.weight /= math.sqrt(n_layer)


def build_aum(size="base", **kw):
    return AudioMamba(embed_dim=AUM_SIZES[size], **kw)
