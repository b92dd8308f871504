"""Host logic of the drop-in package (mamba_ssm / causal_conv1d / aum) on CPU.

The product path has no CPU implementation, so for these tests the tests-only lane-array build of the kernel sources
(tests/emu) is injected where libaum_hip.so would be.  What is checked here is the Python side: autograd wiring, layouts
(channel-major tensors, strided dxz views), autocast rules, state-dict keys and dispatch -- against the golden vectors
produced by the reference's own code (tests/golden/*.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

import aum_hip
import cases
from conftest import load_golden, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module", autouse=True)
def emu_as_product():
    import build_emu
    old = aum_hip._product
    aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    yield
    aum_hip._product = old


def P(a, grad=True):
    return None if a is None else torch.tensor(np.asarray(a)).requires_grad_(grad)


@pytest.mark.parametrize("case", [c for c in cases.SCAN_CASES if c[0] in ("l65", "l65_plain", "l65_noz", "l130_n4", "l2049")],
                         ids=lambda c: c[0])
def test_selective_scan_fn_autograd_vs_reference(case):
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, selective_scan_ref
    g = load_golden("scan")
    name, softplus = case[0], case[8]
    d = cases.scan_inputs(*case)
    t = {k: P(d[k]) for k in ("u", "delta", "A", "D", "z", "delta_bias")}
    Bm, Cm = P(d["B"][:, None]), P(d["C"][:, None])
    out, last = selective_scan_fn(t["u"], t["delta"], t["A"], Bm, Cm, t["D"], t["z"], t["delta_bias"], softplus, True)
    (out * torch.tensor(d["dout"])).sum().backward()
    assert rel_err(out.detach().numpy(), g[name + ".f32.out"]) < 1e-4
    assert rel_err(last.numpy(), g[name + ".f32.last_state"]) < 1e-4
    for k, v in (("du", t["u"]), ("ddelta", t["delta"]), ("dA", t["A"]), ("dB", Bm), ("dC", Cm), ("dD", t["D"]),
                 ("dz", t["z"]), ("ddelta_bias", t["delta_bias"])):
        if v is not None:
            got = v.grad.numpy()
            got = got[:, 0] if k in ("dB", "dC") else got
            assert rel_err(got, g[f"{name}.f32.{k}"]) < 4e-4, k
    # the package's own pure-PyTorch statement agrees with the reference's output as well
    with torch.no_grad():
        ref = selective_scan_ref(t["u"], t["delta"], t["A"], Bm, Cm, t["D"], t["z"], t["delta_bias"], softplus)
    assert rel_err(ref.numpy(), g[name + ".f32.out"]) < 2e-5


@pytest.mark.parametrize("case", cases.NORM_CASES, ids=lambda c: c[0])
def test_rms_norm_fn_vs_reference(case):
    from mamba_ssm.ops.triton.layernorm import rms_norm_fn, rms_norm_ref
    g = load_golden("norm")
    name, lead, cols, has_res, prenorm = case
    d = cases.norm_inputs(*case)
    x, res, w = P(d["x"]), P(d["residual"]), P(d["weight"])
    r = rms_norm_fn(x, w, None, residual=res, prenorm=prenorm, residual_in_fp32=True, eps=1e-5)
    y, res_out = r if prenorm else (r, None)
    loss = (y * torch.tensor(d["dy"])).sum()
    if prenorm:
        loss = loss + (res_out * torch.tensor(d["dres"])).sum()
    loss.backward()
    assert rel_err(y.detach().numpy(), g[name + ".y"]) < 1e-5
    assert rel_err(x.grad.numpy(), g[name + ".dx"]) < 1e-5
    assert rel_err(w.grad.numpy(), g[name + ".dweight"]) < 1e-5
    if has_res:
        assert rel_err(res.grad.numpy(), g[name + ".dresidual"]) < 1e-5
    with torch.no_grad():
        assert rel_err(rms_norm_ref(x, w, None, residual=res, eps=1e-5, upcast=True).numpy(), g[name + ".y"]) < 1e-5


def _run_inner(case, dmajor_xz):
    import mamba_ssm.ops.selective_scan_interface as ssi
    name, mode, batch, d_model, length = case
    p = cases.inner_inputs(*case)
    t = {k: P(v) for k, v in p.items() if k != "dout"}
    xz_leaf = t["xz"]
    xz = xz_leaf
    if dmajor_xz == "token":      # token-major rows [x | z]: the layout of the time-serial kernels
        xz = xz_leaf.transpose(1, 2).contiguous().transpose(1, 2)
        assert ssi._is_tm(xz)
    elif dmajor_xz:     # the channel-major layout Mamba.forward produces for the row kernels (MS:185-189)
        xz = xz_leaf.permute(1, 0, 2).contiguous().permute(1, 0, 2)
    if mode == "v1":
        o = ssi.bimamba_inner_fn(xz, t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                                 t["A"], t["A_b"], None, None, t["D"], delta_bias=t["dt_bias"], delta_softplus=True)
    elif mode == "none":
        o = ssi.mamba_inner_fn(xz, t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                               t["A"], None, None, t["D"], delta_bias=t["dt_bias"], delta_softplus=True)
    else:
        of = ssi.mamba_inner_fn_no_out_proj(xz, t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["A"], None,
                                            None, t["D"], delta_bias=t["dt_bias"], delta_softplus=True)
        ob = ssi.mamba_inner_fn_no_out_proj(xz, t["conv_w_b"], t["conv_b_b"], t["x_proj_w_b"], t["dt_proj_w_b"],
                                            t["A_b"], None, None, t["D_b"], delta_bias=t["dt_bias_b"],
                                            delta_softplus=True, reverse=True)
        o = torch.nn.functional.linear(((of + ob) / 2).transpose(1, 2), t["out_proj_w"], None)
    (o * torch.tensor(p["dout"])).sum().backward()
    return o, t


@pytest.mark.parametrize("case", cases.INNER_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("dmajor_xz", [False, True, "token"])
def test_inner_fns_vs_reference(case, dmajor_xz):
    g = load_golden("inner")
    name = case[0]
    o, t = _run_inner(case, dmajor_xz)
    assert rel_err(o.detach().numpy(), g[name + ".out"]) < 1e-4
    keys = [k[len(name) + 3:] for k in g if k.startswith(name + ".d_")]
    assert len(keys) >= 9
    for k in keys:
        assert t[k].grad is not None, k
        assert rel_err(t[k].grad.numpy(), g[f"{name}.d_{k}"]) < 5e-4, k


def test_inner_ref_functions_match_reference():
    import mamba_ssm.ops.selective_scan_interface as ssi
    g = load_golden("inner")
    case = cases.INNER_CASES[0]
    p = cases.inner_inputs(*case)
    t = {k: P(v, False) for k, v in p.items()}
    o = ssi.bimamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                              t["A"], t["A_b"], None, None, t["D"], delta_bias=t["dt_bias"], delta_softplus=True)
    assert rel_err(o.numpy(), g[case[0] + ".out"]) < 2e-5
    case = cases.INNER_CASES[3]
    p = cases.inner_inputs(*case)
    t = {k: P(v, False) for k, v in p.items()}
    o = ssi.mamba_inner_ref(t["xz"], t["conv_w"], t["conv_b"], t["x_proj_w"], t["dt_proj_w"], t["out_proj_w"], None,
                            t["A"], None, None, t["D"], delta_bias=t["dt_bias"], delta_softplus=True)
    assert rel_err(o.numpy(), g[case[0] + ".out"]) < 2e-5


def test_mamba_module_contract():
    """Constructor defaults, parameter names/shapes (checkpoint contract), init attributes (MS:113,123,127)."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(192, bimamba_type="v2", if_devide_out=True)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sd == {
        "A_log": (384, 16), "D": (384,), "A_b_log": (384, 16), "D_b": (384,), "in_proj.weight": (768, 192),
        "conv1d.weight": (384, 1, 4), "conv1d.bias": (384,), "x_proj.weight": (44, 384),
        "dt_proj.weight": (384, 12), "dt_proj.bias": (384,), "conv1d_b.weight": (384, 1, 4),
        "conv1d_b.bias": (384,), "x_proj_b.weight": (44, 384), "dt_proj_b.weight": (384, 12),
        "dt_proj_b.bias": (384,), "out_proj.weight": (192, 384)}
    assert m.dt_proj.bias._no_reinit and m.A_log._no_weight_decay and m.D._no_weight_decay
    sp = torch.nn.functional.softplus(m.dt_proj.bias)
    assert sp.min() >= 1e-3 * 0.99 and sp.max() <= 0.1 * 1.01          # MS:104-111
    assert torch.allclose(m.A_log[7], torch.log(torch.arange(1, 17.0)))
    assert m.dt_proj.weight.abs().max() <= 12 ** -0.5
    y = m(torch.randn(2, 17, 192))
    assert y.shape == (2, 17, 192)
    with pytest.raises(NotImplementedError):
        m(torch.randn(2, 17, 192), inference_params=object())


@pytest.mark.parametrize("case", cases.MODEL_CASES, ids=lambda c: c[0])
def test_audio_mamba_vs_reference_model(case):
    """Whole-model parity with the reference's AudioMamba (run on CPU through its own *_ref functions when the
    fixture was generated): identical state-dict keys, logits and every parameter gradient."""
    from aum.model import AudioMamba
    g = load_golden("model")
    name, btype, depth, dim, spec, ncls, batch = case[:7]
    model = AudioMamba(spectrogram_size=spec, depth=depth, embed_dim=dim, num_classes=ncls, bimamba_type=btype,
                       **cases.model_kwargs(case))
    sd = model.state_dict()
    assert sorted(sd.keys()) == list(g[name + ".keys"])
    vals = cases.model_state({k: tuple(v.shape) for k, v in sd.items()}, name)
    d = cases.model_inputs(*case)
    assert np.isclose(cases.checksum(dict(vals, **d)), g[name + ".checksum"], rtol=1e-12)
    model.load_state_dict({k: torch.tensor(v) for k, v in vals.items()})
    logits = model(torch.tensor(d["x"]))
    (logits * torch.tensor(d["dlogits"])).sum().backward()
    assert rel_err(logits.detach().numpy(), g[name + ".logits"]) < 1e-3          # north_star fp32 bar
    for k, p_ in model.named_parameters():
        gn = float(np.sqrt((p_.grad.double().numpy() ** 2).sum()))
        ref = float(g[f"{name}.gnorm.{k}"])
        assert abs(gn - ref) <= 2e-3 * max(ref, 1e-6), (k, gn, ref)
        if f"{name}.grad.{k}" in g:
            assert rel_err(p_.grad.numpy(), g[f"{name}.grad.{k}"]) < 2e-3, k


def test_unfused_path_matches_fused():
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(1)
    m = Mamba(32, bimamba_type="none")
    x = torch.randn(2, 70, 32)
    y1 = m(x)
    m.use_fast_path = False
    y2 = m(x)
    assert rel_err(y2.detach().numpy(), y1.detach().numpy()) < 1e-5


def test_bf16_module_runs_and_matches_fp32():
    """A module converted to bf16 end to end (weights included) runs through the same kernels (fp32 parameter
    arguments are converted at the ABI boundary) and stays within the bf16 bar of the fp32 run.  The autocast rules
    (SSI:452-457) need a CUDA autocast context and are tested in test_gpu_model.py."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(2)
    m = Mamba(32, bimamba_type="v1")
    x = torch.randn(2, 65, 32)
    y32 = m(x).detach()
    mb = Mamba(32, bimamba_type="v1")
    mb.load_state_dict(m.state_dict())
    mb = mb.to(torch.bfloat16)
    yb = mb(x.to(torch.bfloat16))
    yb.float().sum().backward()
    for n, p_ in mb.named_parameters():
        assert p_.grad is not None and p_.grad.dtype == torch.bfloat16, n
    assert rel_err(yb.float().detach().numpy(), y32.numpy()) < 5e-2


@pytest.mark.parametrize("btype", ["v1", "v2", "none"])
def test_mfma_projection_path_matches_library_gemm_path(btype, monkeypatch):
    """16-bit activations take the aum_proj_* kernels (x_dbl channel-major, B/C as strided views of it); the same
    module with that path switched off takes torch.matmul + explicit B/C transposes.  Outputs and every gradient
    must agree to bf16 rounding -- this pins the layout plumbing (strided B/C into the scan, dB/dC back)."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(5)
    m = Mamba(64, bimamba_type=btype).to(torch.bfloat16)          # d_inner 128, dt_rank 4, R + 2N = 36
    x = torch.randn(3, 37, 64).to(torch.bfloat16)
    calls = {"n": 0}
    real = aum_hip.proj_fwd

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(aum_hip, "proj_fwd", counting)
    y1 = m(x)
    y1.float().square().sum().backward()
    assert calls["n"] == (2 if btype == "v2" else 1)
    g1 = {n: p.grad.float().clone() for n, p in m.named_parameters()}
    m.zero_grad()
    monkeypatch.setattr(aum_hip, "proj_supported", lambda *a: False)
    y2 = m(x)
    y2.float().square().sum().backward()
    assert calls["n"] == (2 if btype == "v2" else 1)
    assert rel_err(y1.float().detach().numpy(), y2.float().detach().numpy()) < 2e-2
    for n, p in m.named_parameters():
        assert rel_err(g1[n].numpy(), p.grad.float().numpy()) < 4e-2, n


def test_streaming_inference_matches_full_sequence():
    """MS:176-182, 313-400: prefill with an inference cache, then token-by-token step() -- must reproduce the causal
    block's output on the whole sequence (conv window and SSM state carried in place)."""
    from types import SimpleNamespace
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(7)
    m = Mamba(32, layer_idx=0, bimamba_type="none").eval()
    x = torch.randn(2, 70, 32)
    with torch.no_grad():
        full = m(x)
        params = SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0)
        outs = [m(x[:, :65], inference_params=params)]
        conv_state, ssm_state = params.key_value_memory_dict[0]
        assert conv_state.shape == (2, 64, 4) and ssm_state.shape == (2, 64, 16)
        for t in range(65, 70):
            params.seqlen_offset = t
            outs.append(m(x[:, t:t + 1], inference_params=params))
        got = torch.cat(outs, dim=1)
    assert rel_err(got.numpy(), full.numpy()) < 1e-4
    c2, s2 = m.allocate_inference_cache(3, 128)
    assert c2.shape == (3, 64, 4) and s2.shape == (3, 64, 16) and float(c2.abs().sum()) == 0
    with pytest.raises(NotImplementedError):
        Mamba(32, layer_idx=0, bimamba_type="v1")(x, inference_params=params)


def test_step_cache_is_the_per_call_arithmetic(monkeypatch):
    """The per-forward weight cache (16-bit projection weights, their transposes, A = -exp(A_log) for all blocks in a few launches)
    holds exactly what each block would compute per call, lives for one forward, and leaves logits and gradients unchanged."""
    import contextlib
    from aum.model import AudioMamba
    from aum import model as M
    from mamba_ssm.ops import selective_scan_interface as S
    torch.manual_seed(5)
    model = AudioMamba(spectrogram_size=(128, 64), depth=2, embed_dim=32, num_classes=3, bimamba_type="v1")
    mixers = [l.mixer for l in model.layers]
    with S.step_cache(mixers, torch.bfloat16):
        for m in mixers:
            for lin in (m.in_proj, m.x_proj, m.dt_proj, m.out_proj):
                assert torch.equal(S._cast(lin.weight, torch.bfloat16), lin.weight.to(torch.bfloat16))
            for lin in (m.x_proj, m.dt_proj):
                assert torch.equal(S._cast_t(lin.weight, torch.bfloat16), lin.weight.to(torch.bfloat16).t().contiguous())
                assert S._cast_t(lin.weight, torch.bfloat16).is_contiguous()
            assert torch.equal(S.neg_exp(m.A_log), -torch.exp(m.A_log.float()))
            assert torch.equal(S.neg_exp(m.A_b_log), -torch.exp(m.A_b_log.float()))
        assert S._cast(mixers[0].in_proj.weight, torch.float16).dtype == torch.float16      # another dtype: plain cast
    assert not S._STEP_CACHE
    x = torch.randn(2, 64, 128)
    outs = []
    for cached in (True, False):
        if not cached:
            monkeypatch.setattr(M, "step_cache", lambda *a, **k: contextlib.nullcontext())
        model.zero_grad()
        y = model(x)
        y.square().sum().backward()
        outs.append((y.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_split_counts_and_token_split_rules():
    """Host-side launch rules of the library GEMMs: split-K counts are the first preference that divides the token count; the token
    split keeps a tile-aligned first GEMM and only applies to device tensors of a few thousand tokens or more."""
    from mamba_ssm.ops import selective_scan_interface as S
    assert S._pick_splits(64 * 513, (6, 4, 8, 2)) == 6 and S._pick_splits(64 * 513, (9, 8, 4, 2)) == 9
    assert S._pick_splits(8 * 4097, (6, 4, 8, 2)) == 4 and S._pick_splits(8 * 4097, (9, 8, 4, 2)) == 8      # long-form: 2^3 * 17 * 241
    assert S._pick_splits(2 * 513, (6, 4, 8, 2)) == 1                                                          # too short to split
    cpu = torch.zeros(1)
    assert S._tok_n0(64 * 513, 1, cpu) == 0                                                                   # host tensors: one GEMM
    a, bt = torch.randn(5, 7), torch.randn(9000, 7)
    assert torch.equal(S._mm_tokens_cols(a, bt, 1), a @ bt.t())
    at, b = torch.randn(7, 9000), torch.randn(7, 3)
    assert torch.equal(S._mm_tokens_rows(at, b, 8), at.t() @ b)


def test_token_major_module_matches_channel_major(monkeypatch):
    """Mamba.forward in the two activation layouts (token-major rows through InProjTmFn, the register-window conv and the time-serial
    scan; channel-major through the row kernels): same output, same parameter and input gradients; and the dispatch rules -- token-major
    only where the kernels' limits hold and the batch fills the chip."""
    import mamba_ssm.ops.selective_scan_interface as ssi
    from mamba_ssm.modules.mamba_simple import Mamba
    assert ssi.token_major_ok(1536, 16, 4, 48, torch.bfloat16) and ssi.token_major_ok(768, 16, 4, 24, torch.bfloat16)
    assert not ssi.token_major_ok(384, 16, 4, 12, torch.bfloat16)       # B / C rows of x_dbl not 16-byte aligned behind 12 bf16 dt columns
    assert not ssi.token_major_ok(1536, 8, 4, 48, torch.bfloat16) and not ssi.token_major_ok(96, 16, 4, 4, torch.float32)
    assert ssi.token_major_preferred(64, 1536, True) and not ssi.token_major_preferred(8, 1536, True)      # batch 8, short rows: chunk kernels
    # long-form clips (B = 8, L = 4097): token-major on time segments -- as many ranges as give every SIMD its resident waves
    assert ssi.token_major_preferred(8, 1536, True, training=True, seqlen=4097) and not ssi.token_major_preferred(8, 1536, True, seqlen=513)
    assert ssi.tm_segments(8, 1536, 4097, True, False) == 16 and ssi.tm_segments(8, 1536, 4097, True, True) == 16
    assert ssi.tm_segments(64, 1536, 4097, True, True) == 1 and ssi.tm_segments(1, 1536, 513, True, False) == 1
    # AuM-Small at batch 64 is 1536 waves: enough for the forward alone, not when a backward follows
    assert ssi.token_major_preferred(64, 768, True, training=False) and not ssi.token_major_preferred(64, 768, True, training=True)
    assert ssi.token_major_preferred(64, 1536, False, training=True)          # one direction (Fo-Fo): 1536 waves train token-major
    torch.manual_seed(3)
    for btype in ("v1", "none", "v2"):
        m = Mamba(64, bimamba_type=btype, if_devide_out=btype == "v2")
        x = torch.randn(2, 70, 64)
        w = torch.randn(2, 70, 64)
        res = []
        for min_waves in (0, 10 ** 9):
            monkeypatch.setattr(ssi, "_TM_MIN_WAVES", min_waves)
            m.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            (y * w).sum().backward()
            res.append((y.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        assert rel_err(res[0][0].numpy(), res[1][0].numpy()) < 1e-5 and rel_err(res[0][1].numpy(), res[1][1].numpy()) < 1e-4
        for k in res[0][2]:
            assert rel_err(res[0][2][k].numpy(), res[1][2][k].numpy()) < 2e-4, (btype, k)


def test_token_major_segments_module(monkeypatch):
    """the block on time-segmented token-major launches (what long-form clips dispatch to; forced here on a short row with
    AUM_TM_SEGMENTS' module switch) against the channel-major block: output, input gradient, every parameter gradient"""
    import mamba_ssm.ops.selective_scan_interface as ssi
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(4)
    for btype in ("v1", "none", "v2"):
        m = Mamba(64, bimamba_type=btype, if_devide_out=btype == "v2")
        x, w = torch.randn(2, 70, 64), torch.randn(2, 70, 64)
        res = []
        for seg in (3, 0):
            monkeypatch.setattr(ssi, "_TM_SEGMENTS", seg)
            monkeypatch.setattr(ssi, "_TM_MIN_WAVES", 10 ** 9)
            calls = []
            real = aum_hip.scan_tm_fwd
            monkeypatch.setattr(aum_hip, "scan_tm_fwd", lambda *a, **k: (calls.append(k.get("segments")), real(*a, **k))[1])
            m.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            (y * w).sum().backward()
            monkeypatch.setattr(aum_hip, "scan_tm_fwd", real)
            assert calls == (([3, 3] if btype == "v2" else [3]) if seg else []), (btype, seg, calls)
            res.append((y.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        assert rel_err(res[0][0].numpy(), res[1][0].numpy()) < 1e-5 and rel_err(res[0][1].numpy(), res[1][1].numpy()) < 1e-4
        for k in res[0][2]:
            assert rel_err(res[0][2][k].numpy(), res[1][2][k].numpy()) < 2e-4, (btype, k)


def test_dA_log_product_comes_from_the_scan_backward(monkeypatch):
    """A = -exp(A_log) from the forward's cache: the token-major scan backward writes d A .* A next to d A (aum_scan_tm_bwd: dA_xA) and
    _NegExpFn.backward hands it on instead of multiplying -- every A / A_b of the model, same A_log gradients as the multiplying path,
    nothing left in the hand-over table; a block used on its own (no cache) multiplies as before."""
    import mamba_ssm.ops.selective_scan_interface as ssi
    from aum.model import build_aum
    hits = {"product": 0, "multiply": 0}
    orig = ssi._NegExpFn.backward

    def counting(ctx, g):
        before = len(ssi._DA_XA)
        r = orig(ctx, g)
        hits["product" if len(ssi._DA_XA) < before else "multiply"] += 1
        return r
    monkeypatch.setattr(ssi._NegExpFn, "backward", staticmethod(counting))
    monkeypatch.setattr(ssi, "_TM_MIN_WAVES", 0)
    torch.manual_seed(0)
    m = build_aum("small", depth=2, num_classes=5, bimamba_type="v1", spectrogram_size=(128, 64))
    x = torch.randn(2, 64, 128)
    m(x).sum().backward()
    assert hits == {"product": 4, "multiply": 0} and not ssi._DA_XA, (hits, len(ssi._DA_XA))
    fused = {k: p.grad.clone() for k, p in m.named_parameters() if "A_" in k}
    m.zero_grad()
    monkeypatch.setattr(ssi, "_TM_MIN_WAVES", 10 ** 9)          # channel-major block: no product from the kernel
    hits.update(product=0, multiply=0)
    m(x).sum().backward()
    assert hits == {"product": 0, "multiply": 4}
    for k, p in m.named_parameters():
        if "A_" in k:
            assert rel_err(fused[k].numpy(), p.grad.numpy()) < 2e-4, k


def test_time_reversed_block_equals_flip_sandwich(monkeypatch):
    """Mamba.forward(h, time_reversed=True) == flip(forward(flip(h))) for the three block types in both activation layouts -- outputs,
    input gradient and every parameter gradient (the odd layers of an `if_bidirectional` model, MM:623-638, run without the four flipped
    copies per pair: the conv and the scan take direction flags, Fo-Bi's two directions trade their A matrices)."""
    import mamba_ssm.ops.selective_scan_interface as ssi
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(5)
    for btype in ("v1", "none", "v2"):
        m = Mamba(64, bimamba_type=btype, if_devide_out=btype == "v2")
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if n_.startswith("A_b_log"):
                    p_.add_(0.3 * torch.randn_like(p_))          # the two directions must be distinguishable
        x, w = torch.randn(2, 41, 64), torch.randn(2, 41, 64)
        for min_waves in (0, 10 ** 9):
            monkeypatch.setattr(ssi, "_TM_MIN_WAVES", min_waves)
            res = []
            for mode in ("flag", "flips"):
                m.zero_grad()
                xi = x.clone().requires_grad_(True)
                y = m(xi, time_reversed=True) if mode == "flag" else m(xi.flip([1])).flip([1])
                (y * w).sum().backward()
                res.append((y.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
            assert rel_err(res[0][0].numpy(), res[1][0].numpy()) < 1e-5, (btype, min_waves)
            assert rel_err(res[0][1].numpy(), res[1][1].numpy()) < 1e-4, (btype, min_waves)
            for k in res[0][2]:
                assert rel_err(res[0][2][k].numpy(), res[1][2][k].numpy()) < 2e-4, (btype, min_waves, k)


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus N` without a launcher starts N ranks under torch.distributed.run on 127.0.0.1 (the driver's own
    command line); with WORLD_SIZE set it is a rank and does not launch again"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AUM_BENCH_PRINT_LAUNCH"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
