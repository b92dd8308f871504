"""Loading the reference's published AuM checkpoints (`--aum_pretrain`, /root/reference/src/models/mamba_models.py
= "MM":397-446) into aum.model.AudioMamba:

  * DDP/accelerate `module.` prefixes are stripped (MM:400);
  * the absolute position embedding is re-gridded when the clip length differs from the checkpoint's: the cls row is
    kept, the patch rows are bilinearly resampled with antialiasing on the (freq, time) grid
    (/root/reference/src/utilities/tokenization.py:26-66, 357-369); the old grid is recovered from the row count
    assuming 128 mel bins and a power-of-two clip length, as the reference does (MM:419-429);
  * a head with a different class count is dropped (MM:440-444);
  * the patch projection must have the model's patch size (the pseudo-inverse patch resize belongs to the
    flexible-patch path, which is out of scope).
"""
import torch
import torch.nn.functional as F


def _grid(fstride, tstride, patch, fdim, tdim):
    return (fdim - patch[0]) // fstride + 1, (tdim - patch[1]) // tstride + 1


def resample_pos_embed(pos_embed, old_grid, new_grid, n_prefix=1):
    if tuple(old_grid) == tuple(new_grid):
        return pos_embed
    prefix, grid = pos_embed[:, :n_prefix], pos_embed[:, n_prefix:]
    grid = grid.reshape(1, old_grid[0], old_grid[1], -1).permute(0, 3, 1, 2).float()
    grid = F.interpolate(grid, size=tuple(new_grid), mode="bilinear", antialias=True)
    grid = grid.permute(0, 2, 3, 1).reshape(1, new_grid[0] * new_grid[1], -1).to(pos_embed.dtype)
    return torch.cat([prefix, grid], dim=1)


def load_aum_checkpoint(model, weights, pretrain_fstride=None, pretrain_tstride=None, strict_backbone=True):
    """weights: path or state dict.  Returns torch's load_state_dict result (missing / unexpected keys)."""
    if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
        weights = torch.load(weights, map_location="cpu")
    weights = {k.replace("module.", ""): v for k, v in weights.items()}

    proj_w = weights["patch_embed.proj.weight"]
    patch_load = tuple(proj_w.shape[-2:])
    if patch_load != tuple(model.patch_embed.proj.kernel_size):
        raise NotImplementedError(f"checkpoint patch size {patch_load} != model {model.patch_embed.proj.kernel_size}")
    strides_load = (pretrain_fstride or patch_load[0], pretrain_tstride or patch_load[1])

    pe = weights["pos_embed.pos_embed"]
    n_prefix = model.num_tokens
    if pe.shape[1] != model.pos_embed.pos_embed.shape[1]:
        old = None
        for log_len in range(6, 20):
            g = _grid(*strides_load, patch_load, 128, 2 ** log_len)
            if g[0] * g[1] == pe.shape[1] - n_prefix:
                old = g
                break
        if old is None:
            raise ValueError("Could not find matching audio length")
        weights["pos_embed.pos_embed"] = resample_pos_embed(pe, old, model.patch_grid_size, n_prefix)

    if "head.weight" in weights and weights["head.weight"].shape[0] != model.num_classes:
        print("Num classes differ! Can only load the backbone weights.")
        del weights["head.weight"], weights["head.bias"]

    result = model.load_state_dict(weights, strict=False)
    if strict_backbone:
        bad = [k for k in result.missing_keys if not k.startswith("head.")] + list(result.unexpected_keys)
        if bad:
            raise RuntimeError(f"checkpoint does not match the AuM backbone: {bad[:8]}")
    return result
