#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own pure-PyTorch functions on CPU.

Runs only in the build container (needs /root/reference; no-op elsewhere).  Nothing from the
reference is copied: the fixtures hold the outputs of
  selective_scan_ref      (SSI:86-152)  + autograd grads
  MS:272 conv expression  (act(conv1d(x)[..., :L])) + autograd grads
  rms_norm_ref            (LN:35-48, upcast=True) + autograd grads
  mamba_inner_ref / bimamba_inner_ref (SSI:636-709) and the v2 composition of MS:214-246 + grads
for the seeded inputs defined in cases.py, plus an input checksum.

Import recipe: the CUDA extension modules the reference imports at module scope
(causal_conv1d, causal_conv1d_cuda, selective_scan_cuda) are absent here, so empty stand-in MODULE
OBJECTS are registered in sys.modules for the duration of this script -- they provide no arithmetic;
every number written below comes from the reference's own *_ref code paths.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import cases  # noqa: E402


def import_reference():
    import importlib.machinery
    import torch
    import torch.nn.functional as F

    for name in ("causal_conv1d_cuda", "selective_scan_cuda"):
        sys.modules[name] = types.ModuleType(name)
    cc = types.ModuleType("causal_conv1d")
    cc.causal_conv1d_fn = None
    cc.causal_conv1d_update = None
    sys.modules["causal_conv1d"] = cc
    # namespace stub so mamba_ssm/__init__.py (which pulls transformers-era names) is skipped
    pkg = types.ModuleType("mamba_ssm")
    pkg.__path__ = [os.path.join(REF, "vim-mamba_ssm", "mamba_ssm")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("mamba_ssm", None, is_package=True)
    sys.modules["mamba_ssm"] = pkg
    import mamba_ssm.ops.selective_scan_interface as ssi
    import mamba_ssm.ops.triton.layernorm as ln

    def conv_fn(x, weight, bias=None, activation=None):
        # the reference's own non-fused expression, MS:272: act(conv1d(x)[..., :seqlen])
        w = weight.shape[-1]
        y = F.conv1d(x, weight.unsqueeze(1), bias, padding=w - 1, groups=x.shape[1])[..., : x.shape[-1]]
        return F.silu(y) if activation in ("silu", "swish") else y

    ssi.causal_conv1d_fn = conv_fn
    ssi.selective_scan_fn = ssi.selective_scan_ref
    return torch, ssi, ln, conv_fn


def T(torch, a, grad=True, dtype=None):
    if a is None:
        return None
    t = torch.tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(grad)


def npy(t):
    return None if t is None else t.detach().float().cpu().numpy()


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    torch, ssi, ln, conv_fn = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- selective_scan_ref
    out = {}
    for case in cases.SCAN_CASES:
        name = case[0]
        has_z, has_D, has_bias, softplus = case[5:]
        d = cases.scan_inputs(*case)
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            if dt_name == "bf16" and name not in ("l65", "l513"):
                continue
            act = lambda a: T(torch, a, True, dt)
            u, delta, z = act(d["u"]), act(d["delta"]), act(d["z"])
            Bm, Cm = act(d["B"][:, None]), act(d["C"][:, None])       # (B,1,N,L) as the fused path passes them
            A, D, bias = T(torch, d["A"]), T(torch, d["D"]), T(torch, d["delta_bias"])
            o, last = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z, bias, softplus, True)
            (o.float() * torch.tensor(d["dout"])).sum().backward()
            pre = f"{name}.{dt_name}."
            out[pre + "out"] = npy(o)
            out[pre + "last_state"] = npy(last)
            for k, t in (("du", u), ("ddelta", delta), ("dA", A), ("dB", Bm), ("dC", Cm), ("dD", D),
                         ("dz", z), ("ddelta_bias", bias)):
                if t is not None:
                    g = npy(t.grad)
                    out[pre + k] = g[:, 0] if k in ("dB", "dC") else g
            if dt_name == "f32":
                out[name + ".checksum"] = cases.checksum(d)
                # also pin the 3-D B/C entry (SSI:125-126) and the flipped form used for the reverse
                # direction (SSI:707-708) so the oracle's `reverse` flag is pinned to reference output
                with torch.no_grad():
                    fl = lambda a: None if a is None else torch.tensor(a).flip([-1])
                    ob = ssi.selective_scan_ref(fl(d["u"]), fl(d["delta"]), torch.tensor(d["A"]), fl(d["B"]),
                                                fl(d["C"]), T(torch, d["D"], False), fl(d["z"]),
                                                T(torch, d["delta_bias"], False), softplus).flip([-1])
                out[name + ".f32.out_reverse"] = npy(ob)
    np.savez_compressed(os.path.join(HERE, "scan.npz"), **out)
    print("scan.npz", len(out))

    # ---------------------------------------------------------------- conv (MS:272)
    out = {}
    for case in cases.CONV_CASES:
        name = case[0]
        d = cases.conv_inputs(*case)
        x, w, b = T(torch, d["x"]), T(torch, d["weight"]), T(torch, d["bias"])
        y = conv_fn(x, w, b, "silu")
        (y * torch.tensor(d["dout"])).sum().backward()
        out[name + ".y"] = npy(y)
        out[name + ".dx"], out[name + ".dweight"] = npy(x.grad), npy(w.grad)
        if b is not None:
            out[name + ".dbias"] = npy(b.grad)
        with torch.no_grad():
            out[name + ".y_nosilu"] = npy(conv_fn(x, w, b, None))
            # anti-causal form = conv on the flipped sequence, flipped back (MS:229-246)
            out[name + ".y_reverse"] = npy(conv_fn(x.flip([-1]), w, b, "silu").flip([-1]))
        out[name + ".checksum"] = cases.checksum(d)
    np.savez_compressed(os.path.join(HERE, "conv.npz"), **out)
    print("conv.npz", len(out))

    # ---------------------------------------------------------------- rms_norm_ref (LN:35-48)
    out = {}
    for case in cases.NORM_CASES:
        name, lead, cols, has_res, prenorm = case
        d = cases.norm_inputs(*case)
        x, res, w = T(torch, d["x"]), T(torch, d["residual"]), T(torch, d["weight"])
        r = ln.rms_norm_ref(x, w, None, residual=res, eps=1e-5, prenorm=prenorm, upcast=True)
        y, res_out = (r if prenorm else (r, None))
        loss = (y * torch.tensor(d["dy"])).sum()
        if prenorm:
            loss = loss + (res_out * torch.tensor(d["dres"])).sum()
        loss.backward()
        out[name + ".y"] = npy(y)
        if prenorm:
            out[name + ".residual_out"] = npy(res_out)
        out[name + ".dx"], out[name + ".dweight"] = npy(x.grad), npy(w.grad)
        if has_res:
            out[name + ".dresidual"] = npy(res.grad)
        out[name + ".checksum"] = cases.checksum(d)
    np.savez_compressed(os.path.join(HERE, "norm.npz"), **out)
    print("norm.npz", len(out))

    # ---------------------------------------------------------------- inner blocks
    out = {}
    for case in cases.INNER_CASES:
        name, mode, batch, d_model, length = case
        p = cases.inner_inputs(*case)
        t = {k: T(torch, v) for k, v in p.items() if k != "dout"}
        if mode == "v1":
            o = ssi.bimamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"],
                                      t["out_proj_w"], None, t["A"], t["A_b"], None, None, t["D"],
                                      delta_bias=t["dt_bias"], delta_softplus=True)
        elif mode == "none":
            o = ssi.mamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"],
                                    t["out_proj_w"], None, t["A"], None, None, t["D"],
                                    delta_bias=t["dt_bias"], delta_softplus=True)
        else:  # v2 = MS:214-246 with if_devide_out=True, built from the reference's mamba_inner_ref
            E = 2 * d_model
            eye = torch.eye(E)     # out_proj = identity turns mamba_inner_ref into its no-out-proj form
            of = ssi.mamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"],
                                     eye, None, t["A"], None, None, t["D"], delta_bias=t["dt_bias"],
                                     delta_softplus=True)          # (B, L, E)
            ob = ssi.mamba_inner_ref(t["xz"].flip([-1]), t["conv_w_b"], t["conv_b_b"], t["x_proj_w_b"],
                                     t["dt_proj_w_b"], eye, None, t["A_b"], None, None, t["D_b"],
                                     delta_bias=t["dt_bias_b"], delta_softplus=True)
            o = torch.nn.functional.linear((of + ob.flip([1])) / 2, t["out_proj_w"], None)
        (o * torch.tensor(p["dout"])).sum().backward()
        out[name + ".out"] = npy(o)
        for k, v in t.items():
            if v.grad is not None:
                out[name + ".d_" + k] = npy(v.grad)
        out[name + ".checksum"] = cases.checksum(p)
    np.savez_compressed(os.path.join(HERE, "inner.npz"), **out)
    print("inner.npz", len(out))


def import_reference_model(torch, ssi, ln):
    """SURVEY 8c 'whole model on CPU': init-only helper stubs for timm/wget, reference model loaded from its file,
    fused entry points rebound to the reference's own *_ref functions."""
    import importlib.machinery
    import importlib.util
    import torch.nn as nn
    timm, tm, tl = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.layers"))

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.DropPath = DropPath
    tl.trunc_normal_ = lambda t, std=0.02, **kw: nn.init.trunc_normal_(t, std=std)
    tl.lecun_normal_ = lambda t: nn.init.trunc_normal_(t, std=(1.0 / t[0].numel()) ** 0.5)
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "wget": types.ModuleType("wget")})
    for name, path in (("src", REF + "/src"), ("src.models", REF + "/src/models"), ("src.utilities", REF + "/src/utilities")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
    import mamba_ssm.modules.mamba_simple as ms
    ms.bimamba_inner_fn = ssi.bimamba_inner_ref
    ms.mamba_inner_fn = ssi.mamba_inner_ref

    def no_out_proj_ref(xz, cw, cb, xw, dw, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                        C_proj_bias=None, delta_softplus=True):
        eye = torch.eye(xz.shape[1] // 2)
        return ssi.mamba_inner_ref(xz, cw, cb, xw, dw, eye, None, A, B, C, D, delta_bias=delta_bias,
                                   delta_softplus=True).transpose(1, 2)
    ms.mamba_inner_fn_no_out_proj = no_out_proj_ref

    def rms_ref(x, w, b, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
        return ln.rms_norm_ref(x, w, b, residual=residual, eps=eps, prenorm=prenorm, upcast=True)
    ln.rms_norm_fn = rms_ref
    spec = importlib.util.spec_from_file_location("src.models.mamba_models", REF + "/src/models/mamba_models.py")
    mm = importlib.util.module_from_spec(spec)
    sys.modules["src.models.mamba_models"] = mm
    spec.loader.exec_module(mm)
    mm.rms_norm_fn = rms_ref
    return mm


def model_goldens(torch, ssi, ln):
    import contextlib
    import io
    mm = import_reference_model(torch, ssi, ln)
    out = {}
    for case in cases.MODEL_CASES:
        name, btype, depth, dim, spec, ncls, batch = case[:7]
        with contextlib.redirect_stdout(io.StringIO()):
            model = mm.AudioMamba(spectrogram_size=spec, depth=depth, embed_dim=dim, num_classes=ncls,
                                  bimamba_type=btype, **cases.model_kwargs(case))
        sd = model.state_dict()
        vals = cases.model_state({k: tuple(v.shape) for k, v in sd.items()}, name)
        model.load_state_dict({k: torch.tensor(v) for k, v in vals.items()})
        d = cases.model_inputs(*case)
        logits = model(torch.tensor(d["x"]))
        (logits * torch.tensor(d["dlogits"])).sum().backward()
        out[name + ".logits"] = npy(logits)
        out[name + ".keys"] = np.array(sorted(sd.keys()))
        for k, p in model.named_parameters():
            g = npy(p.grad)
            out[name + ".gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            if p.numel() <= 4096:
                out[name + ".grad." + k] = g
        out[name + ".checksum"] = cases.checksum(dict(vals, **d))
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)
    print("model.npz", len(out))


def launcher_goldens(torch, ssi, ln):
    """the reference's own training loop (src/traintest.py `train`, run unmodified on CPU through
    accelerate) and its own checkpoint loading (`AudioMamba(aum_pretrain=True)`, MM:397-446 + TOK:26-66) on the seeded toy
    problems of cases.py.  Stored: learning rate at every optimizer step, Adam's hyper-parameters, result.csv (train / valid loss,
    metrics, lr per epoch), the parameters after training; the re-gridded position embedding and the logits of the loaded model."""
    import contextlib
    import importlib.util
    import io
    import tempfile
    from argparse import Namespace
    mm = import_reference_model(torch, ssi, ln)
    out = {}
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())

    # ---------------- f2: checkpoint with a different clip length and class count ----------------
    c = cases.CKPT_CASE
    kw = dict(depth=c["depth"], embed_dim=c["embed_dim"], bimamba_type="v1")
    with quiet():
        src = mm.AudioMamba(spectrogram_size=c["src_spec"], num_classes=c["src_classes"], **kw)
    vals = cases.model_state({k: tuple(v.shape) for k, v in src.state_dict().items()}, "ckpt_src")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "src.pth")
        torch.save({"module." + k: torch.tensor(v) for k, v in vals.items()}, path)       # as a DDP run saves it
        with quiet():
            dst = mm.AudioMamba(spectrogram_size=c["dst_spec"], num_classes=c["dst_classes"], aum_pretrain=True,
                                aum_pretrain_path=path, **kw)
    head = cases.model_state({k: tuple(v.shape) for k, v in dst.state_dict().items() if k.startswith("head.")}, "ckpt_dst_head")
    dst.load_state_dict({k: torch.tensor(v) for k, v in head.items()}, strict=False)       # the head is not in the checkpoint's shape
    with torch.no_grad():
        logits = dst(torch.tensor(cases.ckpt_inputs()["x"]))
    out["ckpt.pos_embed"] = npy(dst.pos_embed.pos_embed)
    out["ckpt.logits"] = npy(logits)
    out["ckpt.checksum"] = cases.checksum(dict(vals, **cases.ckpt_inputs()))

    # ---------------- f1: traintest.train ----------------
    import accelerate
    sys.path.insert(0, REF + "/src")
    sys.modules.pop("utilities", None)
    spec = importlib.util.spec_from_file_location("ref_traintest", REF + "/src/traintest.py")
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    t = cases.TRAIN_CASE
    tr, va = cases.train_inputs()

    class DS(torch.utils.data.Dataset):
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return len(self.d["x"])

        def __getitem__(self, i):
            return torch.tensor(self.d["x"][i]), torch.tensor(self.d["y"][i]), f"clip{i}"

    with quiet():
        model = mm.AudioMamba(spectrogram_size=t["spec"], depth=t["depth"], embed_dim=t["embed_dim"], num_classes=t["n_class"],
                              bimamba_type="v1")
    init = cases.model_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, "train_init")
    model.load_state_dict({k: torch.tensor(v) for k, v in init.items()})
    lrs, hyper = [], {}
    RealAdam = torch.optim.Adam

    class SpyAdam(RealAdam):
        def __init__(self, params, lr, **kwargs):
            hyper.update(dict(lr=lr, **kwargs))
            super().__init__(params, lr, **kwargs)

        def step(self, *a, **k):
            lrs.append(self.param_groups[0]["lr"])
            return super().step(*a, **k)

    with tempfile.TemporaryDirectory() as exp:
        os.makedirs(exp + "/models")
        args = Namespace(accelerator=accelerate.Accelerator(cpu=True), lr=t["lr"], weight_decay=t["weight_decay"],
                         bs_scale_factor=t["bs_scale_factor"], optim_path=None, exp_dir=exp, n_epochs=t["n_epochs"], metrics="mAP",
                         loss="BCE", warmup=True, dataset="audioset", lrscheduler_start=t["lrscheduler_start"],
                         lrscheduler_step=t["lrscheduler_step"], lrscheduler_decay=t["lrscheduler_decay"], model="aum",
                         flexible_training=False, if_random_cls_token_position=False, if_nan2num=True, if_continue_inf=False,
                         save_model=True)
        torch.optim.Adam = SpyAdam
        try:
            with quiet(), contextlib.redirect_stderr(io.StringIO()):
                tt.train(model, torch.utils.data.DataLoader(DS(tr), batch_size=t["batch"], shuffle=False),
                         torch.utils.data.DataLoader(DS(va), batch_size=2 * t["batch"], shuffle=False), args)
        finally:
            torch.optim.Adam = RealAdam
        out["train.result"] = np.loadtxt(exp + "/result.csv", delimiter=",")
        out["train.predictions"] = np.loadtxt(exp + f"/predictions/predictions_{t['n_epochs']}.csv", delimiter=",")
    out["train.lr_per_step"] = np.array(lrs, np.float64)
    out["train.adam"] = np.array([hyper["betas"][0], hyper["betas"][1], hyper["eps"], hyper["weight_decay"], hyper["lr"]], np.float64)
    for k, p_ in model.state_dict().items():
        out["train.final." + k] = npy(p_)
    out["train.checksum"] = cases.checksum(dict(init, tx=tr["x"], ty=tr["y"], vx=va["x"], vy=va["y"]))
    np.savez_compressed(os.path.join(HERE, "launcher.npz"), **out)
    print("launcher.npz", len(out))


def wav_excerpts():
    """Real audio for the log-mel frontend tests: 3-second int16 excerpts of the five example clips the reference ships for its
    inference notebook (examples/inference/data/sample{0..4}.wav, 16 kHz mono; data files, not source).  They hold what synthetic
    noise + sines do not: near-silent stretches (the log floor), onsets, a real spectral tilt.  No expected outputs are stored --
    torchaudio is absent, so the frontend's oracle stays parity-unpinned; the tests compare kernel and oracle on these inputs."""
    import wave
    out = {}
    for i in range(5):
        w = wave.open(os.path.join(REF, "examples", "inference", "data", f"sample{i}.wav"))
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        start = (1 + 2 * i) * 16000 - 37 * i                   # different offsets, not frame-aligned
        out[f"sample{i}"] = pcm[start:start + 3 * 16000].copy()
    np.savez_compressed(os.path.join(HERE, "wav_excerpts.npz"), **out)
    print("wav_excerpts.npz", {k: (v.shape, int(np.abs(v).max())) for k, v in out.items()})

def headline_goldens(torch, ssi, ln):
    """BASELINE configs 2 and 3 at full size through the reference's AudioMamba on CPU (selective_scan_ref loop; about ten minutes
    and 10 GB for the Base backward).  Written to headline.npz; run with --headline (not part of the default regeneration)."""
    import contextlib
    import io
    import time
    mm = import_reference_model(torch, ssi, ln)
    out = {}
    for case in cases.HEADLINE_CASES:
        name, btype, depth, dim, spec, ncls, batch, bwd = case
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            model = mm.AudioMamba(spectrogram_size=spec, depth=depth, embed_dim=dim, num_classes=ncls, bimamba_type=btype)
        sd = model.state_dict()
        vals = cases.model_state({k: tuple(v.shape) for k, v in sd.items()}, name)
        model.load_state_dict({k: torch.tensor(v) for k, v in vals.items()})
        d = cases.model_inputs(*case[:7])
        with (contextlib.nullcontext() if bwd else torch.no_grad()):
            logits = model(torch.tensor(d["x"]))
        out[name + ".logits"] = npy(logits)
        out[name + ".keys"] = np.array(sorted(sd.keys()))
        if bwd:
            (logits * torch.tensor(d["dlogits"])).sum().backward()
            for k, p in model.named_parameters():
                g = npy(p.grad)
                out[name + ".gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                if p.numel() <= 1024:
                    out[name + ".grad." + k] = g
        out[name + ".checksum"] = cases.checksum(dict(vals, **d))
        print(name, "done in %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "headline.npz"), **out)
    print("headline.npz", len(out))



def headline_bf16_goldens(torch, ssi, ln, dtype_name="bf16"):
    """(dtype_name "fp16": the same run under torch.autocast(float16) -> headline_fp16.npz -- the precision every exps/**/aum-*.sh of the
    reference launches with (`--mixed_precision=fp16`, exps/audioset/aum-base_scratch-audioset.sh:54); VERDICT r5 weak #1.)
    The SAME-PRECISION pin of the headline configuration: the reference's AudioMamba at BASELINE configs 3 and 2
    run under torch.autocast(bfloat16) on CPU -- the projections (F.linear / matmul) and the conv round to bf16 exactly where the
    reference's autocast run does (SSI:452-457 casts the projection weights, the activations between the ops are 16-bit), while the
    selective scan keeps its fp32 interior: on the GPU `selective_scan_cuda` computes in fp32 whatever the I/O type and
    MambaInnerFn.forward is `custom_fwd` (autocast disabled inside), so `selective_scan_ref` (which upcasts at SSI:101-107) is called
    with autocast switched off around it -- otherwise its contracting einsums would be autocast to bf16 bmm, which no GPU run of the
    reference does.  Written to headline_bf16.npz next to the fp32 pin (headline.npz) of the same seeded parameters and inputs."""
    import contextlib
    import io
    import time
    ref_scan = ssi.selective_scan_ref
    lowp = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype_name]

    def scan_fp32_interior(*a, **k):
        with torch.autocast("cpu", enabled=False):
            return ref_scan(*a, **k)

    ssi.selective_scan_fn = scan_fp32_interior
    mm = import_reference_model(torch, ssi, ln)
    out = {}
    for case in cases.HEADLINE_CASES:
        name, btype, depth, dim, spec, ncls, batch, bwd = case
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            model = mm.AudioMamba(spectrogram_size=spec, depth=depth, embed_dim=dim, num_classes=ncls, bimamba_type=btype)
        sd = model.state_dict()
        vals = cases.model_state({k: tuple(v.shape) for k, v in sd.items()}, name)
        model.load_state_dict({k: torch.tensor(v) for k, v in vals.items()})
        d = cases.model_inputs(*case[:7])
        with (contextlib.nullcontext() if bwd else torch.no_grad()):
            with torch.autocast("cpu", dtype=lowp):
                logits = model(torch.tensor(d["x"]))
        out[name + ".logits"] = npy(logits)
        out[name + ".logits_dtype"] = np.array(str(logits.dtype))
        if bwd:
            (logits.float() * torch.tensor(d["dlogits"])).sum().backward()
            for k, p in model.named_parameters():
                g = npy(p.grad)
                out[name + ".gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                if p.numel() <= 1024:
                    out[name + ".grad." + k] = g
        out[name + ".checksum"] = cases.checksum(dict(vals, **d))
        print(name, dtype_name, "autocast done in %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, f"headline_{dtype_name}.npz"), **out)
    print(f"headline_{dtype_name}.npz", len(out))


if __name__ == "__main__":
    if os.path.isdir(REF) and ("--headline-bf16" in sys.argv or "--headline-fp16" in sys.argv):
        torch_, ssi_, ln_, _ = import_reference()
        torch_.manual_seed(0)
        torch_.set_num_threads(8)
        headline_bf16_goldens(torch_, ssi_, ln_, "fp16" if "--headline-fp16" in sys.argv else "bf16")
        sys.exit(0)
    if os.path.isdir(REF) and "--headline" in sys.argv:
        torch_, ssi_, ln_, _ = import_reference()
        torch_.manual_seed(0)
        torch_.set_num_threads(8)
        headline_goldens(torch_, ssi_, ln_)
        sys.exit(0)
    main()
    if os.path.isdir(REF) and "--wav-only" in sys.argv:
        wav_excerpts()
        sys.exit(0)
    if os.path.isdir(REF) and "--no-model" not in sys.argv:
        torch_, ssi_, ln_, _ = import_reference()
        model_goldens(torch_, ssi_, ln_)
        launcher_goldens(torch_, ssi_, ln_)
        wav_excerpts()
