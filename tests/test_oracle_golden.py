"""Pin the oracle (oracle/oracle.py + C) to the reference's own outputs (tests/golden/*.npz, produced
by tests/golden/make_golden.py importing /root/reference).  CPU only."""
import numpy as np
import pytest

import cases
from conftest import load_golden, rel_err
from oracle import oracle as O

TOL = {"f32": 2e-5, "f64": 2e-5}   # golden itself is fp32 arithmetic (einsum order differs): ~1e-6 typical


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("case", cases.SCAN_CASES, ids=lambda c: c[0])
def test_scan_vs_reference(case, prec):
    g = load_golden("scan")
    name = case[0]
    softplus = case[8]
    d = cases.scan_inputs(*case)
    assert np.isclose(cases.checksum(d), g[name + ".checksum"], rtol=1e-12), "input generator drifted"
    r = O.scan_fwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], softplus,
                   False, prec)
    assert rel_err(r["out"], g[name + ".f32.out"]) < TOL[prec]
    assert rel_err(r["last_state"], g[name + ".f32.last_state"]) < TOL[prec]
    rr = O.scan_fwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], softplus,
                    True, prec)
    assert rel_err(rr["out"], g[name + ".f32.out_reverse"]) < TOL[prec]
    gr = O.scan_bwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["dout"],
                    softplus, False, prec)
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"):
        key = f"{name}.f32.{k}"
        if key in g:
            assert rel_err(gr[k], g[key]) < 20 * TOL[prec], k
        else:
            assert gr[k] is None or k in ("dD", "dz", "ddelta_bias")


def test_scan_reverse_bwd_is_flip_of_forward_bwd():
    """reverse=True adjoint == flip / forward adjoint / flip (SSI:548-561)."""
    case = cases.SCAN_CASES[2]
    d = cases.scan_inputs(*case)
    fl = lambda a: None if a is None else np.ascontiguousarray(a[..., ::-1])
    g1 = O.scan_bwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["dout"],
                    True, True, "f64")
    g2 = O.scan_bwd(fl(d["u"]), fl(d["delta"]), d["A"], fl(d["B"]), fl(d["C"]), d["D"], fl(d["z"]),
                    d["delta_bias"], fl(d["dout"]), True, False, "f64")
    for k in ("du", "ddelta", "dB", "dC", "dz"):
        assert rel_err(g1[k], fl(g2[k])) < 1e-12
    for k in ("dA", "dD", "ddelta_bias"):
        assert rel_err(g1[k], g2[k]) < 1e-12


@pytest.mark.parametrize("case", ["l65", "l513"])
def test_scan_bf16_io_vs_reference(case):
    """bf16 I/O: fp32 internal math, output rounded to bf16 (SSI:101-103,151): 1e-2 bar of north_star."""
    import torch
    g = load_golden("scan")
    c = [x for x in cases.SCAN_CASES if x[0] == case][0]
    d = cases.scan_inputs(*c)
    bf = lambda a: None if a is None else torch.tensor(a).bfloat16().float().numpy()
    r = O.scan_fwd(bf(d["u"]), bf(d["delta"]), d["A"], bf(d["B"]), bf(d["C"]), d["D"], bf(d["z"]),
                   d["delta_bias"], True, False, "f32")
    out = torch.tensor(r["out"]).bfloat16().float().numpy()
    assert rel_err(out, g[case + ".bf16.out"]) < 1e-2


@pytest.mark.parametrize("case", cases.CONV_CASES, ids=lambda c: c[0])
def test_conv_vs_reference(case):
    g = load_golden("conv")
    name = case[0]
    d = cases.conv_inputs(*case)
    assert np.isclose(cases.checksum(d), g[name + ".checksum"], rtol=1e-12)
    for prec in ("f32", "f64"):
        assert rel_err(O.conv1d_fwd(d["x"], d["weight"], d["bias"], True, False, prec), g[name + ".y"]) < 1e-5
        assert rel_err(O.conv1d_fwd(d["x"], d["weight"], d["bias"], False, False, prec), g[name + ".y_nosilu"]) < 1e-5
        assert rel_err(O.conv1d_fwd(d["x"], d["weight"], d["bias"], True, True, prec), g[name + ".y_reverse"]) < 1e-5
        gr = O.conv1d_bwd(d["x"], d["weight"], d["bias"], d["dout"], True, False, prec)
        assert rel_err(gr["dx"], g[name + ".dx"]) < 1e-5
        assert rel_err(gr["dweight"], g[name + ".dweight"]) < 1e-5
        if d["bias"] is not None:
            assert rel_err(gr["dbias"], g[name + ".dbias"]) < 1e-5


@pytest.mark.parametrize("case", cases.NORM_CASES, ids=lambda c: c[0])
def test_rmsnorm_vs_reference(case):
    g = load_golden("norm")
    name, lead, cols, has_res, prenorm = case
    d = cases.norm_inputs(*case)
    assert np.isclose(cases.checksum(d), g[name + ".checksum"], rtol=1e-12)
    for prec in ("f32", "f64"):
        r = O.rmsnorm_fwd(d["x"], d["weight"], None, d["residual"], 1e-5, prec)
        assert rel_err(r["y"], g[name + ".y"]) < 1e-5
        if prenorm:
            assert rel_err(r["residual_out"], g[name + ".residual_out"]) < 1e-6
        b = O.rmsnorm_bwd(d["dy"], r["residual_out"], d["weight"], r["rstd"], d["dres"], False, prec)
        assert rel_err(b["dx"], g[name + ".dx"]) < 1e-5
        assert rel_err(b["dweight"], g[name + ".dweight"]) < 1e-5
        if has_res:
            assert rel_err(b["dx"], g[name + ".dresidual"]) < 1e-5


def _run_inner(case, prec):
    name, mode, batch, d_model, length = case
    p = cases.inner_inputs(*case)
    if mode in ("v1", "none"):
        A_b = p["A_b"] if mode == "v1" else None
        st = O.inner_fwd(p["xz"], p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"],
                         None, p["A"], p["D"], p["dt_bias"], A_b, prec)
        gr = O.inner_full_bwd(st, p["dout"], p["xz"], p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"],
                              p["out_proj_w"], None, p["A"], p["D"], p["dt_bias"], A_b, prec)
        res = {"out": st["out"], "d_xz": gr["dxz"], "d_conv_w": gr["dconv_w"], "d_conv_b": gr["dconv_b"],
               "d_x_proj_w": gr["dx_proj_w"], "d_dt_proj_w": gr["ddt_proj_w"], "d_A": gr["dA"],
               "d_D": gr["dD"], "d_dt_bias": gr["ddelta_bias"], "d_out_proj_w": gr["dout_proj_w"]}
        if mode == "v1":
            res["d_A_b"] = gr["dA_b"]
        return p, res
    # v2: MS:214-246, if_devide_out=True
    sf = O.inner_no_out_proj_fwd(p["xz"], p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["A"],
                                 p["D"], p["dt_bias"], False, prec)
    sb = O.inner_no_out_proj_fwd(p["xz"], p["conv_w_b"], p["conv_b_b"], p["x_proj_w_b"], p["dt_proj_w_b"],
                                 p["A_b"], p["D_b"], p["dt_bias_b"], True, prec)
    y = (sf["out_z"] + sb["out_z"]) / 2
    out = y.transpose(0, 2, 1) @ p["out_proj_w"].T.astype(y.dtype)
    dout2 = p["dout"].reshape(-1, d_model).astype(y.dtype)
    dy = ((dout2 @ p["out_proj_w"].astype(y.dtype)).reshape(batch, length, -1).transpose(0, 2, 1)) / 2
    dy = np.ascontiguousarray(dy)
    gf = O.inner_bwd(sf, dy, p["xz"], p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["A"], p["D"],
                     p["dt_bias"], None, False, prec)
    gb = O.inner_bwd(sb, dy, p["xz"], p["conv_w_b"], p["conv_b_b"], p["x_proj_w_b"], p["dt_proj_w_b"],
                     p["A_b"], p["D_b"], p["dt_bias_b"], None, True, prec)
    res = {"out": out, "d_xz": gf["dxz"] + gb["dxz"],
           "d_out_proj_w": dout2.T @ y.transpose(0, 2, 1).reshape(batch * length, -1),
           "d_conv_w": gf["dconv_w"], "d_conv_b": gf["dconv_b"], "d_x_proj_w": gf["dx_proj_w"],
           "d_dt_proj_w": gf["ddt_proj_w"], "d_A": gf["dA"], "d_D": gf["dD"], "d_dt_bias": gf["ddelta_bias"],
           "d_conv_w_b": gb["dconv_w"], "d_conv_b_b": gb["dconv_b"], "d_x_proj_w_b": gb["dx_proj_w"],
           "d_dt_proj_w_b": gb["ddt_proj_w"], "d_A_b": gb["dA"], "d_D_b": gb["dD"],
           "d_dt_bias_b": gb["ddelta_bias"]}
    return p, res


@pytest.mark.parametrize("case", cases.INNER_CASES, ids=lambda c: c[0])
def test_inner_vs_reference(case):
    g = load_golden("inner")
    name = case[0]
    p, res = _run_inner(case, "f64")
    assert np.isclose(cases.checksum(p), g[name + ".checksum"], rtol=1e-12)
    keys = [k[len(name) + 1:] for k in g if k.startswith(name + ".") and not k.endswith("checksum")]
    assert len(keys) >= 10
    for k in keys:
        assert k in res, k
        assert rel_err(res[k], g[name + "." + k]) < 2e-4, k   # reference autograd runs in fp32


def test_headline_fixture_is_self_consistent():
    """tests/golden/headline.npz (the reference's AudioMamba at BASELINE configs 3 and 2, make_golden.py --headline) stores outputs only;
    its inputs -- 92 M seeded parameter values and the clip -- are regenerated from cases.py.  The stored checksum pins that generator: if
    it drifts, the GPU tests would compare the product with goldens of different weights."""
    import cases
    g = load_golden("headline")
    case = cases.HEADLINE_CASES[1]                      # AuM-Small: 24 M parameters, a few seconds
    name = case[0]
    keys = [str(k) for k in g[name + ".keys"]]
    assert len(keys) > 200 and g[name + ".logits"].shape == (case[6], case[5])
    # shapes of the state dict follow from the architecture (MM:678-685): rebuild them from the package's own model
    import torch
    from aum.model import AudioMamba
    with torch.device("meta"):
        model = AudioMamba(spectrogram_size=case[4], depth=case[2], embed_dim=case[3], num_classes=case[5], bimamba_type=case[1])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sorted(shapes) == keys
    vals = cases.model_state(shapes, name)
    d = cases.model_inputs(*case[:7])
    ref = float(g[name + ".checksum"])
    assert abs(float(cases.checksum(dict(vals, **d))) - ref) <= 1e-9 * abs(ref)
    base = cases.HEADLINE_CASES[0][0]
    assert sum(k.startswith(base + ".gnorm.") for k in g) == 271 and g[base + ".logits"].shape == (1, 527)
