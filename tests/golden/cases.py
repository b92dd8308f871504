"""Seeded input generators shared by make_golden.py (which runs the reference on them, in the build
container only) and the parity tests (which regenerate the same inputs on the GPU box).

Fixtures store the reference's OUTPUTS plus a checksum of the inputs, never reference code.
Input distributions follow SURVEY.md 8(d): u ~ N(0,1), delta_raw ~ N(0,0.5^2), A = -(1..N) times a
"trained-like" jitter, B,C,z ~ N(0,1), D ~ 1, dt_bias = softplus^-1(logU[1e-3,1e-1]) (MS:104-109).
"""
import numpy as np

# (name, batch, dim, len, dstate, has_z, has_D, has_bias, softplus)
SCAN_CASES = [
    ("l1", 2, 8, 1, 16, True, True, True, True),
    ("l64", 2, 16, 64, 16, True, True, True, True),
    ("l65", 2, 16, 65, 16, True, True, True, True),
    ("l65_plain", 2, 16, 65, 16, False, False, False, False),
    ("l65_noz", 2, 16, 65, 16, False, True, True, True),
    ("l65_nod", 2, 16, 65, 16, True, False, False, True),
    ("l513", 1, 8, 513, 16, True, True, True, True),
    ("l2049", 1, 4, 2049, 16, True, True, True, True),
    ("l130_n4", 2, 8, 130, 4, True, True, True, True),
    ("l577_n8", 1, 6, 577, 8, True, True, True, True),
]

# checked against the live oracle only (no fixture): several row groups per batch entry, a ragged last group, idle waves
SCAN_WIDE_CASES = [
    ("l513_d100", 2, 100, 513, 16, True, True, True, True),
    ("l513_d70_n8_noz", 1, 70, 513, 8, False, True, False, True),
]

# L = 513 rows for the row kernels with the lane-entry checkpoint (scan_row_kernels.h): the wide cases above plus a row with no
# gate / D / bias / softplus and a single ragged group; live oracle only
SCAN_ROW_CASES = SCAN_WIDE_CASES + [
    ("l513_d5_plain", 2, 5, 513, 16, False, False, False, False),
    ("l513_d3_n5_nod", 1, 3, 513, 5, True, False, True, True),
]

# long rows of 512*m (+1) steps: the backward is the chunked one-row kernel (one launch per direction, carries between the
# 512-step chunks, the tail step owned by the last chunk); live oracle only.  Two row groups with idle waves / fewer states /
# no gate, no D, no bias, no softplus / three chunks
SCAN_LONG_CASES = [
    ("l1024_d70_n8", 1, 70, 1024, 8, True, True, True, True),
    ("l1025_d5_plain", 2, 5, 1025, 16, False, False, False, False),
    ("l1537_d3", 1, 3, 1537, 16, True, True, True, True),
]

# time-serial token-major scan (scan_tm_kernels.h: lanes = channels, dim % 64 == 0, dstate 16): lengths around the 8-step block /
# 4-step prefetch group boundaries, an even and an odd meeting point of the direction pair, two channel groups; live oracle only
SCAN_TM_CASES = [
    ("tm_l1", 2, 64, 1, 16, True, True, True, True),
    ("tm_l2", 1, 64, 2, 16, True, True, True, True),
    ("tm_l7", 1, 64, 7, 16, True, True, True, True),
    ("tm_l8", 2, 64, 8, 16, True, True, True, True),
    ("tm_l9", 1, 64, 9, 16, True, True, True, True),
    ("tm_l25", 1, 128, 25, 16, True, True, True, True),
    ("tm_l40_plain", 2, 64, 40, 16, False, False, False, False),
    ("tm_l65_noz", 1, 64, 65, 16, False, True, True, True),
    ("tm_l66_nod", 1, 64, 66, 16, True, False, False, True),
    ("tm_l75", 1, 64, 75, 16, True, True, True, True),            # meeting point 37 | 38: the directions' iterations numbered from 3 and 2 (scant_grid_shift)
    ("tm_l100", 1, 64, 100, 16, True, True, True, True),          # 50 | 50: both from 6, one block more than ceil(100 / 8)
    ("tm_l130", 2, 128, 130, 16, True, True, True, True),
    ("tm_l513", 1, 64, 513, 16, True, True, True, True),
]

# (name, batch, dim, len, width, has_bias)
CONV_CASES = [
    ("l1", 2, 8, 1, 4, True),
    ("l3", 2, 8, 3, 4, True),
    ("l65", 2, 8, 65, 4, True),
    ("l513", 1, 16, 513, 4, True),
    ("l65_nobias", 2, 8, 65, 4, False),
    ("l70_w3", 1, 4, 70, 3, True),
    # the AuM row shape 512 + tail (several rows of one channel per wave, tail steps wave-uniform): batch not a multiple of 8
    ("l513_b11", 11, 3, 513, 4, True),
    ("l520_b3", 3, 4, 520, 4, False),
    ("l512_b9", 9, 2, 512, 4, True),
]

# token-major conv (batch, len, dim): dim a multiple of 8; chunks of 64 steps per wave, 512 channels (16-bit) / 256 (fp32) per wave
CONV_TM_CASES = [
    ("tm_l1", 2, 8, 1, 4, True),
    ("tm_l3", 2, 16, 3, 4, True),
    ("tm_l65", 2, 24, 65, 4, True),
    ("tm_l130_nobias", 2, 8, 130, 4, False),
    ("tm_l70_w3", 1, 8, 70, 3, True),
    ("tm_l64", 1, 8, 64, 4, True),
    ("tm_l513_d520", 1, 520, 513, 4, True),
]

# (name, rows(list shape), cols, has_residual, prenorm)
NORM_CASES = [
    ("r130_c192_res_pre", (2, 65), 192, True, True),
    ("r130_c192_nores_pre", (2, 65), 192, False, True),
    ("r33_c768_res", (1, 33), 768, True, False),
    ("r7_c100_res_pre", (7,), 100, True, True),
]

# (name, mode, batch, d_model, len)      mode: v1 (Fo-Bi), v2 (Bi-Bi), none (Fo-Fo)
INNER_CASES = [
    ("v1_d16_l65", "v1", 2, 16, 65),
    ("v1_d16_l513", "v1", 2, 16, 513),
    ("v1_d64_l65", "v1", 2, 64, 65),
    ("none_d16_l65", "none", 2, 16, 65),
    ("v2_d16_l65", "v2", 2, 16, 65),
    ("v2_d64_l130", "v2", 1, 64, 130),
    # long-form row (512*2 + cls): the host composes one launch per direction, kernels = the 512-step chunked ones
    ("v1_d16_l1025", "v1", 1, 16, 1025),
]


def _rng(tag):
    seed = int.from_bytes(tag.encode(), "little") % (2 ** 31 - 1)
    return np.random.default_rng(seed)


def checksum(arrs):
    """Order-dependent fp64 checksum of a dict of arrays (guards against generator drift)."""
    tot = 0.0
    for i, k in enumerate(sorted(arrs)):
        a = arrs[k]
        if a is None:
            continue
        a = np.asarray(a, np.float64).ravel()
        tot += float((a * np.cos(np.arange(a.size) * 0.37 + i)).sum())
    return np.float64(tot)


def dt_bias_init(rng, dim, dt_min=1e-3, dt_max=1e-1):
    dt = np.exp(rng.random(dim) * (np.log(dt_max) - np.log(dt_min)) + np.log(dt_min)).clip(min=1e-4)
    return (dt + np.log(-np.expm1(-dt))).astype(np.float32)


def scan_inputs(name, batch, dim, length, dstate, has_z, has_D, has_bias, softplus):
    r = _rng("scan_" + name)
    f = np.float32
    A = -(np.arange(1, dstate + 1, dtype=f)[None, :] * np.exp(r.normal(0, 0.1, (dim, dstate)))).astype(f)
    d = dict(
        u=r.normal(0, 1, (batch, dim, length)).astype(f),
        delta=(r.normal(0, 0.5, (batch, dim, length)) if softplus
               else np.abs(r.normal(0, 0.05, (batch, dim, length))) + 1e-3).astype(f),
        A=A,
        B=r.normal(0, 1, (batch, dstate, length)).astype(f),
        C=r.normal(0, 1, (batch, dstate, length)).astype(f),
        D=(1.0 + r.normal(0, 0.1, dim)).astype(f) if has_D else None,
        z=r.normal(0, 1, (batch, dim, length)).astype(f) if has_z else None,
        delta_bias=dt_bias_init(r, dim) if has_bias else None,
        dout=r.normal(0, 1, (batch, dim, length)).astype(f),
    )
    if softplus and has_bias:
        # exercise the softplus linear branch (x > 20) and a very negative input
        d["delta"][0, 0, 0] = 25.0
        if length > 2:
            d["delta"][-1, -1, 2] = -12.0
    return d


def conv_inputs(name, batch, dim, length, width, has_bias):
    r = _rng("conv_" + name)
    f = np.float32
    return dict(
        x=r.normal(0, 1, (batch, dim, length)).astype(f),
        weight=r.normal(0, 0.5, (dim, width)).astype(f),
        bias=r.normal(0, 0.2, dim).astype(f) if has_bias else None,
        dout=r.normal(0, 1, (batch, dim, length)).astype(f),
    )


def norm_inputs(name, lead, cols, has_res, prenorm):
    r = _rng("norm_" + name)
    f = np.float32
    return dict(
        x=r.normal(0, 1, lead + (cols,)).astype(f),
        residual=r.normal(0, 2, lead + (cols,)).astype(f) if has_res else None,
        weight=(1 + r.normal(0, 0.2, cols)).astype(f),
        dy=r.normal(0, 1, lead + (cols,)).astype(f),
        dres=r.normal(0, 1, lead + (cols,)).astype(f) if prenorm else None,
    )


def inner_params(r, d_model, d_state=16, d_conv=4, expand=2, v2=False):
    """Parameter tensors with the shapes of Mamba.__init__ (MS:74-167), trained-like values."""
    f = np.float32
    E = expand * d_model
    R = -(-d_model // 16)
    p = dict(
        conv_w=r.normal(0, 0.4, (E, 1, d_conv)).astype(f),
        conv_b=r.normal(0, 0.2, E).astype(f),
        x_proj_w=(r.normal(0, 1, (R + 2 * d_state, E)) / np.sqrt(E)).astype(f),
        dt_proj_w=r.uniform(-R ** -0.5, R ** -0.5, (E, R)).astype(f),
        dt_bias=dt_bias_init(r, E),
        A=-(np.arange(1, d_state + 1, dtype=f)[None, :] * np.exp(r.normal(0, 0.1, (E, d_state)))).astype(f),
        D=(1.0 + r.normal(0, 0.1, E)).astype(f),
        out_proj_w=(r.normal(0, 1, (d_model, E)) / np.sqrt(E)).astype(f),
    )
    p["A_b"] = -(np.arange(1, d_state + 1, dtype=f)[None, :] * np.exp(r.normal(0, 0.1, (E, d_state)))).astype(f)
    if v2:
        p["conv_w_b"] = r.normal(0, 0.4, (E, 1, d_conv)).astype(f)
        p["conv_b_b"] = r.normal(0, 0.2, E).astype(f)
        p["x_proj_w_b"] = (r.normal(0, 1, (R + 2 * d_state, E)) / np.sqrt(E)).astype(f)
        p["dt_proj_w_b"] = r.uniform(-R ** -0.5, R ** -0.5, (E, R)).astype(f)
        p["dt_bias_b"] = dt_bias_init(r, E)
        p["D_b"] = (1.0 + r.normal(0, 0.1, E)).astype(f)
    return p


def inner_inputs(name, mode, batch, d_model, length):
    r = _rng("inner_" + name)
    f = np.float32
    p = inner_params(r, d_model, v2=(mode == "v2"))
    E = 2 * d_model
    p["xz"] = r.normal(0, 1, (batch, 2 * E, length)).astype(f)
    p["dout"] = r.normal(0, 1, (batch, length, d_model)).astype(f)
    return p


# (name, bimamba_type, depth, embed_dim, spectrogram (F, T), num_classes, batch)
MODEL_CASES = [
    ("tiny_v1_d2", "v1", 2, 192, (128, 128), 35, 4),
    ("tiny_v2_d2", "v2", 2, 192, (128, 128), 35, 2),
    ("tiny_none_d2", "none", 2, 192, (128, 128), 35, 2),
    ("d64_v1_d3_t256", "v1", 3, 64, (128, 256), 10, 2),
    # off-default flags of run.py: layer pairing on flipped sequences, end cls token, time-major token order
    ("d64_none_pairs_d4", "none", 4, 64, (128, 128), 10, 2, {"if_bidirectional": True}),
    ("d64_v1_endcls_tr", "v1", 2, 64, (128, 128), 10, 2, {"use_middle_cls_token": False, "use_end_cls_token": True,
                                                           "transpose_token_sequence": True}),
    ("d64_v2_headcls", "v2", 2, 64, (128, 128), 10, 2, {"use_middle_cls_token": False}),
    # BASELINE config 1 at its full depth: AuM-Tiny, 12 Fo-Bi blocks, SpeechCommands shape (L = 65), batch 4
    ("tiny_v1_d12", "v1", 12, 192, (128, 128), 35, 4),
]


# BASELINE configs 3 and 2 at their full size (MM:678-685 + RUN:227-237): AuM-Base (d_model 768, 24 Fo-Bi blocks, 128 x 1024 frames ->
# L = 513 tokens, 527 classes) forward + backward on one clip, AuM-Small (d_model 384) forward only on two clips.  The state comes
# from model_state (seeded), so the fixture (headline.npz) holds logits, gradient norms and the small gradients only.
# (name, bimamba_type, depth, embed_dim, spectrogram (F, T), num_classes, batch, backward)
HEADLINE_CASES = [
    ("base_v1_d24_l513", "v1", 24, 768, (128, 1024), 527, 1, True),
    ("small_v1_d24_l513", "v1", 24, 384, (128, 1024), 527, 2, False),
]


def model_kwargs(case):
    return case[7] if len(case) > 7 else {}


def model_state(shapes, tag):
    """Seeded 'trained-like' values for every entry of an AudioMamba state dict ({key: shape})."""
    r = _rng("model_" + tag)
    f = np.float32
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        leaf = k.split(".")[-2] + "." + k.split(".")[-1] if "." in k else k
        if k.endswith("A_log") or k.endswith("A_b_log"):
            v = np.log(np.arange(1, shp[1] + 1, dtype=f)[None, :] * np.exp(r.normal(0, 0.1, shp)))
        elif k.endswith(".D") or k.endswith(".D_b") or k.endswith("norm.weight") or k.endswith("norm_f.weight"):
            v = 1.0 + r.normal(0, 0.1, shp)
        elif "dt_proj" in k and k.endswith("bias"):
            v = dt_bias_init(r, shp[0])
        elif "conv1d" in k and k.endswith("weight"):
            v = r.normal(0, 0.4, shp)
        elif k.endswith("bias"):
            v = r.normal(0, 0.1, shp)
        elif k in ("cls_token", "pos_embed.pos_embed"):
            v = r.normal(0, 0.5, shp)
        else:   # dense weights: unit-variance outputs
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            v = r.normal(0, 1, shp) / np.sqrt(fan_in)
        out[k] = np.asarray(v, f)
    return out


def model_inputs(name, bimamba_type, depth, embed_dim, spec, num_classes, batch, extra=None):
    r = _rng("modelin_" + name)
    f = np.float32
    return dict(x=(0.5 * r.normal(0, 1, (batch, spec[1], spec[0]))).astype(f),
                dlogits=r.normal(0, 1, (batch, num_classes)).astype(f))

# (name, d_inner, dt_rank, d_state, batch, len): the MFMA projection kernels (dim % 64 == 0, dt_rank + 2 d_state <= 80)
PROJ_CASES = [
    ("base_ranks", 128, 48, 16, 3, 70),      # R + 2N = 80: all 5 column blocks, 2 dt k-chunks, ragged last token tile
    ("small_ranks", 192, 24, 16, 1, 129),    # AuM-Small: R = 24 (dt rows share a 16-block with B rows)
    ("tiny_ranks", 64, 12, 16, 2, 33),       # AuM-Tiny: R = 12 (k-groups straddle R), one staging step pair
    ("n8", 64, 8, 8, 5, 13),                 # d_state 8, ntok = 65 (one full tile + 1 token)
]


# ---- launcher and checkpoint (8f2) fixtures: the reference's own traintest.train / AudioMamba(aum_pretrain=True) ----
# tiny Fo-Bi model on 64-frame clips, 16 training clips in batches of 4 (4 steps per epoch), 2 epochs; bs_scale_factor 16 makes
# the 50-step warm-up stairs of TT:118-124 three steps wide (lr changes at steps 0, 3, 6), scales Adam's betas to (0.2, 0.984)
# and eps to 2.5e-9 (TT:25-33), and the MultiStepLR milestone falls on epoch 2
TRAIN_CASE = dict(depth=1, embed_dim=32, spec=(128, 64), n_class=4, batch=4, n_train=16, n_val=6, n_epochs=2, lr=2e-3,
                  bs_scale_factor=16, weight_decay=5e-7, lrscheduler_start=1, lrscheduler_step=1, lrscheduler_decay=0.5)


def train_inputs():
    r = _rng("train_data")
    c = TRAIN_CASE
    f = np.float32
    mk = lambda n: dict(x=(0.5 * r.normal(0, 1, (n, c["spec"][1], c["spec"][0]))).astype(f),
                        y=(r.random((n, c["n_class"])) < 0.35).astype(f))
    tr, va = mk(c["n_train"]), mk(c["n_val"])
    va["y"][:c["n_class"]] = np.eye(c["n_class"], dtype=f)        # every class has a positive in the validation set (AP defined)
    va["y"][c["n_class"]:] = 0
    va["y"][c["n_class"]:, 0] = 1
    return tr, va


# checkpoint re-grid: a (128 x 256)-frame model with 7 classes loaded into a (128 x 512)-frame model with 3 classes
CKPT_CASE = dict(depth=1, embed_dim=32, src_spec=(128, 256), src_classes=7, dst_spec=(128, 512), dst_classes=3, batch=2)


def ckpt_inputs():
    r = _rng("ckpt_in")
    c = CKPT_CASE
    return dict(x=(0.5 * r.normal(0, 1, (c["batch"], c["dst_spec"][1], c["dst_spec"][0]))).astype(np.float32))


# aum_gemm_tn (ABI 9): (name, m, n, k, row-pitch padding of a / c in elements).  One ragged row block (m < 256, m = 256 q + r), a single row,
# more than one tile in both directions, every K the model has (768 / 1536 / 3072 divide by 64; 64 is one K-step: no pipelining at all),
# operands / results that are column blocks of wider tensors.
GEMM_CASES = [
    ("m1_k64", 1, 256, 64, 0, 0),
    ("m255_k128", 255, 256, 128, 0, 0),
    ("m257_n512_k192", 257, 512, 192, 0, 0),
    ("m513_n768_k768", 513, 768, 768, 0, 0),
    ("m700_n256_k1536_pitch", 700, 256, 1536, 64, 256),
    ("m300_n1536_k3072", 300, 1536, 3072, 0, 0),
    # the persistent kernel's item list: a 128-row remainder exactly, 129 rows (full item with rows out of range), 64 rows behind full blocks
    ("m384_k128", 384, 256, 128, 0, 0),
    ("m385_n512_k128", 385, 512, 128, 0, 0),
    ("m576_n512_k192", 576, 512, 192, 0, 0),
]
# aum_gemm_wgrad: (tokens, n, k, splits, pad_y, pad_x) -- one K-step, ragged last step, empty splits, both operands slices of wider rows
GEMM_WGRAD_CASES = [(64, 256, 256, 1, 0, 0), (65, 256, 256, 1, 0, 0), (200, 256, 512, 2, 0, 0), (513, 512, 256, 3, 8, 16), (130, 256, 256, 7, 0, 8),
                    (1026, 768, 256, 4, 768, 0),
                    # the skinny second operand (k = 48 / 80: dt_proj's and x_proj's weight gradients; x = the first 48 of 80-column rows)
                    (64, 256, 48, 1, 0, 32), (65, 256, 80, 1, 0, 0), (513, 512, 48, 3, 8, 32), (130, 256, 80, 7, 0, 0), (1026, 768, 80, 4, 768, 16)]
# (the GPU-only sizes below also cover more items than CUs: several tiles per workgroup, odd and even step counts)
# on the GPU only (the host build's triple loop would take minutes): the bench's own GEMMs, (m, n, k) of in_proj / out_proj forward and
# data gradient at 64 x 513 tokens and at 3 x 513 tokens
GEMM_FULL_CASES = [(70000, 512, 192), (64 * 513, 3072, 768), (64 * 513, 768, 1536), (64 * 513, 1536, 768), (64 * 513, 768, 3072), (3 * 513, 3072, 768)]

# aum_dtproj_tm_fwd (ABI 9): (ntok, dim, dt_rank, columns of the x_dbl rows).  One K-step (rank <= 32) and two, rank not a multiple of 32,
# ragged token tiles (32 tokens per wave, 4 waves per workgroup), AuM-Base / Small / a 64-rank row
DTPROJ_CASES = [(1, 64, 8, 40), (33, 96, 24, 56), (129, 256, 48, 80), (513, 1536, 48, 80), (200, 768, 24, 56), (70, 128, 64, 96), (31, 32, 16, 48)]

# aum_xdt_tm_fwd (ABI 9): (ntok, dim, dt_rank) -- ragged token tiles (128 per workgroup, 32 per wave), one and two dt K-steps, every dim class
# (ntok, dim, pad columns behind the ddelta / du rows): aum_xdt_tm_bwd; 2052 and 2305 tokens: 9 waves per workgroup on 256 CUs, ragged last workgroup
XDT_BWD_CASES = [(1, 256, 0), (33, 256, 8), (127, 512, 0), (145, 768, 16), (513, 1536, 0), (300, 1024, 8), (2305, 1536, 0)]
XDT_CASES = [(1, 256, 8), (33, 256, 24), (127, 512, 48), (129, 768, 32), (513, 1536, 48), (300, 1024, 64),
             # 56-column x_dbl rows (AuM-Small: dt rank 24 + 2 x 16): (ntok, dim, rank, ncols)
             (1, 256, 24, 56), (145, 768, 24, 56), (2305, 768, 24, 56), (300, 512, 16, 56)]
