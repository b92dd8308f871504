"""Shared parity checks: run an aum_hip.Lib (the product libaum_hip.so on a GPU, or the tests-only lane-array
build on the host) against the oracle on seeded inputs.  Used by test_emu_kernels.py (CPU) and
test_gpu_kernels.py (-m gpu)."""
import numpy as np
import pytest
import torch

import aum_hip
import cases
from conftest import rel_err, rel_err_by, rms_err
from oracle import oracle as O

TOL_F32 = 1e-4      # north_star bar for fp32 is 1e-3; the kernels are held to 1e-4 of the fp64 oracle
TOL_BF16 = 1e-2     # north_star bar for bf16 I/O


def T(a, dev, dtype=torch.float32):
    return None if a is None else torch.tensor(np.asarray(a)).to(dtype).to(dev)


def N(t):
    return None if t is None else t.detach().float().cpu().numpy()


def rq(a, dtype):
    """round-trip through the activation dtype (what the kernel will see)."""
    return None if a is None else torch.tensor(np.asarray(a)).to(dtype).float().numpy()


def _scan_errors(pairs):
    """{name: (got, reference)} -> three errors per tensor: `name` max error over the largest reference element (rel_err), `rms:name`
    rms error over the rms of the reference (the typical element), and for the per-state tensors `state:name` rel_err taken state by
    state (dA column n, dB / dC row n against that state's own largest element: the late states of dA are orders of magnitude below
    the first ones and invisible to the other two)."""
    errs = {}
    for k, (got, ref) in pairs.items():
        errs[k] = rel_err(got, ref)
        errs["rms:" + k] = rms_err(got, ref)
        if k in ("dA", "dA_b", "dB", "dC"):
            errs["state:" + k] = rel_err_by(got, ref, 1)
    return errs


def check_scan(lib, dev, case, dtype=torch.float32, reverse=False, bidir=False, tol=None, strided=False, generic=False,
               rowpair=False, ckpt=False, lane_ckpt=False):
    """lane_ckpt: L = 513 rows -- let the forward fill the lane-entry checkpoint and hand it to the backward (the row kernels of
    scan_row_kernels.h; without it the backward runs the previous-generation kernels).
    ckpt: hand the forward's chunk-entry-state checkpoint to the backward (long rows; the backward then skips its pre-pass)"""
    name, batch, dim, length, dstate, has_z, has_D, has_bias, softplus = case
    d = cases.scan_inputs(*case)
    tol = tol or (TOL_F32 if dtype == torch.float32 else TOL_BF16)
    act = lambda a: T(a, dev, dtype)
    q = {k: rq(d[k], dtype) for k in ("u", "delta", "z", "B", "C", "dout")}
    rng = np.random.default_rng(7)
    A_b = (d["A"] * np.exp(rng.normal(0, 0.1, d["A"].shape))).astype(np.float32) if bidir else None
    u, delta, z = act(d["u"]), act(d["delta"]), act(d["z"])
    if strided:   # d-major storage like the reference's xz view (MS:185-189): batch stride = len
        mk = lambda t: None if t is None else t.permute(1, 0, 2).contiguous().permute(1, 0, 2)
        u, delta, z = mk(u), mk(delta), mk(z)
    Bm, Cm = act(d["B"]).unsqueeze(1), act(d["C"]).unsqueeze(1)
    A, D, bias = T(d["A"], dev), T(d["D"], dev), T(d["delta_bias"], dev)
    x_ck = aum_hip.scan_ckpt(u, dstate, lib=lib) if ckpt else None
    assert not ckpt or x_ck is not None
    x_lane = aum_hip.scan_lane_ckpt(u, dstate, bidir, lib=lib) if lane_ckpt else None
    assert not lane_ckpt or x_lane is not None
    if x_lane is not None:
        x_lane.fill_(float("nan"))      # every entry the backward reads must have been written by the forward
    out, out_pre, last = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, softplus, reverse, T(A_b, dev),
                                          want_out_pre=True, want_last_state=not bidir, generic=generic, rowpair=rowpair,
                                          x_ck=x_ck, x_lane=x_lane, lib=lib)
    ref = O.scan_fwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus,
                     reverse, "f64")
    ref_out, ref_pre = ref["out"], ref["y_pre"]
    if bidir:
        rb = O.scan_fwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus, True, "f64")
        ref_out, ref_pre = ref_out + rb["out"], ref_pre + rb["y_pre"]
    pairs = {"out": (N(out), ref_out), "out_pre": (N(out_pre), ref_pre)}
    if not bidir:
        pairs["last_state"] = (N(last), ref["last_state"])
    # backward
    dout = act(d["dout"])
    g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, out_pre if has_z else None, softplus, reverse,
                         T(A_b, dev), generic=generic, rowpair=rowpair, x_ck=x_ck, x_lane=x_lane, lib=lib)
    gr = O.scan_bwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus,
                    reverse, "f64")
    if bidir:
        gb = O.scan_bwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus,
                        True, "f64")
        for k in ("du", "ddelta", "dB", "dC", "dD", "dz", "ddelta_bias"):
            if gr[k] is not None:
                gr[k] = gr[k] + gb[k]
        gr["dA_b"] = gb["dA"]
    for k in ("du", "ddelta", "dA", "dA_b", "dB", "dC", "dD", "dz", "ddelta_bias"):
        if gr.get(k) is None:
            assert g.get(k) is None, k
            continue
        pairs[k] = (N(g[k]), gr[k])
    errs = _scan_errors(pairs)
    bad = {k: v for k, v in errs.items() if not (v < tol * (4 if k.split(":")[-1].startswith("d") else 1))}
    assert not bad, (name, str(dtype), "rev" if reverse else "fwd", "bidir" if bidir else "uni", bad, errs)
    return errs


def check_conv(lib, dev, case, dtype=torch.float32, reverse=False, silu=True, generic=False):
    name = case[0]
    d = cases.conv_inputs(*case)
    tol = 1e-5 if dtype == torch.float32 else TOL_BF16
    q = {k: rq(d[k], dtype) for k in ("x", "dout")}
    x, dy = T(d["x"], dev, dtype), T(d["dout"], dev, dtype)
    w, b = T(d["weight"], dev), T(d["bias"], dev)
    y = aum_hip.conv1d_fwd(x, w, b, silu, reverse, generic=generic, lib=lib)
    ry = O.conv1d_fwd(q["x"], d["weight"], d["bias"], silu, reverse, "f64")
    dx, dw, db = aum_hip.conv1d_bwd(x, w, b, dy, silu, reverse, generic=generic, lib=lib)
    rg = O.conv1d_bwd(q["x"], d["weight"], d["bias"], q["dout"], silu, reverse, "f64")
    errs = {"y": rel_err(N(y), ry), "dx": rel_err(N(dx), rg["dx"]), "dw": rel_err(N(dw), rg["dweight"])}
    if b is not None:
        errs["db"] = rel_err(N(db), rg["dbias"])
    bad = {k: v for k, v in errs.items() if not v < tol * (4 if k != "y" else 1)}
    assert not bad, (name, bad)
    return errs


def check_conv_tm(lib, dev, case, dtype=torch.float32, reverse=False, silu=True, xz_layout=False):
    """aum_conv1d_tm_fwd / _bwd against the same oracle as the channel-major conv, on (batch, len, dim) operands; xz_layout: x and dx
    are the first halves of (batch, len, 2 dim) tensors, as in the block."""
    name = case[0]
    d = cases.conv_inputs(*case)
    tol = 1e-5 if dtype == torch.float32 else TOL_BF16
    q = {k: rq(d[k], dtype) for k in ("x", "dout")}
    tm = lambda a: T(np.ascontiguousarray(a.transpose(0, 2, 1)), dev, dtype)
    x, dy = tm(d["x"]), tm(d["dout"])
    dim = x.shape[2]
    dx_out = None
    if xz_layout:
        xz = torch.zeros(x.shape[0], x.shape[1], 2 * dim, dtype=x.dtype, device=x.device)
        xz[:, :, :dim] = x
        x = xz[:, :, :dim]
        dxz = torch.zeros_like(xz)
        dx_out = dxz[:, :, :dim]
    assert aum_hip.conv1d_tm_supported(x, d["weight"].shape[1])
    w, b = T(d["weight"], dev), T(d["bias"], dev)
    y = aum_hip.conv1d_tm_fwd(x, w, b, silu, reverse, lib=lib)
    ry = O.conv1d_fwd(q["x"], d["weight"], d["bias"], silu, reverse, "f64")
    dx, dw, db = aum_hip.conv1d_tm_bwd(x, w, b, dy, silu, reverse, dx_out=dx_out, lib=lib)
    rg = O.conv1d_bwd(q["x"], d["weight"], d["bias"], q["dout"], silu, reverse, "f64")
    untm = lambda t: N(t).transpose(0, 2, 1)
    errs = {"y": rel_err(untm(y), ry), "dx": rel_err(untm(dx), rg["dx"]), "dw": rel_err(N(dw), rg["dweight"])}
    if b is not None:
        errs["db"] = rel_err(N(db), rg["dbias"])
    if xz_layout:
        assert float(dxz[:, :, dim:].abs().max()) == 0.0, (name, "dx wrote outside its half")
    bad = {k: v for k, v in errs.items() if not v < tol * (4 if k != "y" else 1)}
    assert not bad, (name, bad)
    return errs


def check_norm(lib, dev, case, dtype=torch.float32, res_dtype=torch.float32, generic=False):
    name, lead, cols, has_res, prenorm = case
    d = cases.norm_inputs(*case)
    tol = 1e-5 if dtype == torch.float32 else TOL_BF16
    x = T(d["x"], dev, dtype).reshape(-1, cols)
    res = T(d["residual"], dev, res_dtype)
    res = None if res is None else res.reshape(-1, cols)
    w = T(d["weight"], dev)
    y, rstd, res_out = aum_hip.rmsnorm_fwd(x, w, res, 1e-5, residual_dtype=res_dtype, generic=generic, lib=lib)
    qx = rq(d["x"], dtype).reshape(-1, cols)
    qr = None if d["residual"] is None else rq(d["residual"], res_dtype).reshape(-1, cols)
    r = O.rmsnorm_fwd(qx, d["weight"], None, qr, 1e-5, "f64")
    errs = {"y": rel_err(N(y), r["y"]), "res_out": rel_err(N(res_out), r["residual_out"]),
            "rstd": rel_err(N(rstd), r["rstd"])}
    dy = T(d["dy"], dev, dtype).reshape(-1, cols)
    dres = None if d["dres"] is None else T(d["dres"], dev, res_out.dtype).reshape(-1, cols)
    dx, dw, dres_in = aum_hip.rmsnorm_bwd(dy, res_out, w, rstd, dres, has_res, x_dtype=dtype, generic=generic, lib=lib)
    qdres = None if d["dres"] is None else rq(d["dres"], res_out.dtype).reshape(-1, cols)
    rb = O.rmsnorm_bwd(rq(d["dy"], dtype).reshape(-1, cols), N(res_out), d["weight"], N(rstd), qdres, False, "f64")
    errs["dx"] = rel_err(N(dx), rb["dx"])
    errs["dw"] = rel_err(N(dw), rb["dweight"])
    if has_res:
        errs["dres_in"] = rel_err(N(dres_in), rb["dx"])
    bad = {k: v for k, v in errs.items() if not v < tol * 4}
    assert not bad, (name, bad)
    return errs


def check_wave_scan(lib, dev):
    rng = np.random.default_rng(3)
    for rev in (False, True):
        P = rng.uniform(0.2, 1.0, 64).astype(np.float32)
        S = rng.normal(0, 1, 64).astype(np.float32)
        _, So = aum_hip.selftest_wave_scan(T(P, dev), T(S, dev), rev, lib=lib)
        x, ref = 0.0, np.zeros(64)
        for l in (range(63, -1, -1) if rev else range(64)):
            x = float(P[l]) * x + float(S[l])
            ref[l] = x
        assert rel_err(N(So), ref) < 1e-5, ("wave scan", rev)


def check_wave_sum32(lib, dev):
    rng = np.random.default_rng(5)
    v = rng.normal(0, 1, (32, 64)).astype(np.float32)
    got = N(aum_hip.selftest_wave_sum32(T(v, dev), lib=lib))
    ref = np.array([v[2 * (l & 15) + ((l >> 4) & 1)].astype(np.float64).sum() for l in range(64)])
    assert rel_err(got[0], ref) < 1e-5, ("wave_sum32", got[0], ref)
    k16 = lambda l: 8 * ((l >> 3) & 1) + 4 * ((l >> 2) & 1) + 2 * ((l >> 4) & 1) + ((l >> 5) & 1)      # wave_sum16_value_of_lane
    ref16 = np.array([v[k16(l)].astype(np.float64).sum() for l in range(64)])
    assert rel_err(got[1], ref16) < 1e-5, ("wave_sum16", got[1], ref16)


def check_proj(lib, dev, case, dtype=torch.bfloat16):
    """aum_proj_fwd / _bwd_data / _bwd_weight against fp64 matmuls of the same (rounded) operands: SSI:467-468 and
    SSI:570-590 restated on channel-major operands.  Second-stage references take the kernel's own 16-bit x_dbl /
    dx_dbl as input so a 1-ulp rounding difference in stage one is not amplified into a false stage-two error."""
    name, dim, R, Nst, batch, length = case
    ntok, rt = batch * length, R + 2 * Nst
    rng = np.random.default_rng(11)
    tol = TOL_BF16 if dtype == torch.bfloat16 else 2e-3
    conv = rq(rng.normal(size=(dim, ntok)), dtype)
    w_x = rq(rng.normal(size=(rt, dim)) / np.sqrt(dim), dtype)
    w_dt = rq(rng.normal(size=(dim, R)) / np.sqrt(R), dtype)
    act = lambda a: T(a, dev, dtype).contiguous()
    x_dbl, delta = aum_hip.proj_fwd(act(conv), act(w_x), act(w_dt), Nst, lib=lib)
    errs = {"x_dbl": rel_err(N(x_dbl), w_x.astype(np.float64) @ conv),
            "delta": rel_err(N(delta), w_dt.astype(np.float64) @ N(x_dbl)[:R].astype(np.float64))}
    ddelta = rq(rng.normal(size=(dim, ntok)), dtype)
    du = rq(rng.normal(size=(dim, ntok)), dtype)
    dB = rng.normal(size=(batch, Nst, length)).astype(np.float32)
    dC = rng.normal(size=(batch, Nst, length)).astype(np.float32)
    dconv = act(du)
    dx_dbl = aum_hip.proj_bwd_data(act(ddelta), act(w_dt.T.copy()), act(w_x.T.copy()), T(dB, dev), T(dC, dev), dconv,
                                   length, lib=lib)
    flat = lambda g: g.transpose(1, 0, 2).reshape(Nst, ntok)
    ref_dx = np.concatenate([w_dt.T.astype(np.float64) @ ddelta, flat(dB), flat(dC)], axis=0)
    errs["dx_dbl"] = rel_err(N(dx_dbl), ref_dx)
    errs["dconv"] = rel_err(N(dconv), du + w_x.T.astype(np.float64) @ N(dx_dbl).astype(np.float64))
    dw_x = aum_hip.proj_bwd_weight(act(conv), dx_dbl, True, lib=lib)
    dw_dt = aum_hip.proj_bwd_weight(act(ddelta), x_dbl[:R], False, lib=lib)
    assert dw_x.shape == (rt, dim) and dw_dt.shape == (dim, R)
    errs["dw_x"] = rel_err(N(dw_x), N(dx_dbl).astype(np.float64) @ conv.T)
    errs["dw_dt"] = rel_err(N(dw_dt), ddelta.astype(np.float64) @ N(x_dbl)[:R].astype(np.float64).T)
    for k, e in errs.items():
        assert e < (1e-4 if k.startswith("dw") else tol), (name, k, e, errs)
    return errs


def check_scan_accumulate(lib, dev, case, dtype=torch.float32, tol=None):
    """long rows: a reverse-time call with accumulate_into= lands on the forward-time call's out / du / ddelta / dz / dB / dC /
    dD / ddelta_bias; the result must be the oracle's sum of the two directions (SSI:507, 554-559)"""
    name, batch, dim, length, dstate, has_z, has_D, has_bias, softplus = case
    d = cases.scan_inputs(*case)
    tol = tol or (TOL_F32 if dtype == torch.float32 else TOL_BF16)
    act = lambda a: T(a, dev, dtype)
    q = {k: rq(d[k], dtype) for k in ("u", "delta", "z", "B", "C", "dout")}
    A_b = (d["A"] * np.exp(np.random.default_rng(7).normal(0, 0.1, d["A"].shape))).astype(np.float32)
    u, delta, z = act(d["u"]), act(d["delta"]), act(d["z"])
    Bm, Cm = act(d["B"]).unsqueeze(1), act(d["C"]).unsqueeze(1)
    A, Ab, D, bias = T(d["A"], dev), T(A_b, dev), T(d["D"], dev), T(d["delta_bias"], dev)
    assert aum_hip.scan_accumulates(u, dstate, lib=lib)
    of, pre_f, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, softplus, False, want_out_pre=True, lib=lib)
    out, pre_b, _ = aum_hip.scan_fwd(u, delta, Ab, Bm, Cm, D, z, bias, softplus, True, want_out_pre=True, accumulate_into=of, lib=lib)
    assert out is of
    rf = O.scan_fwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus, False, "f64")
    rb = O.scan_fwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus, True, "f64")
    errs = {"out": rel_err(N(out), rf["out"] + rb["out"]), "out_pre_b": rel_err(N(pre_b), rb["y_pre"])}
    dout = act(d["dout"])
    g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre_f if has_z else None, softplus, False, lib=lib)
    g2 = aum_hip.scan_bwd(u, delta, Ab, Bm, Cm, D, z, bias, dout, pre_b if has_z else None, softplus, True, accumulate_into=g, lib=lib)
    gf = O.scan_bwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus, False, "f64")
    gb = O.scan_bwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus, True, "f64")
    for k in ("du", "ddelta", "dB", "dC", "dD", "dz", "ddelta_bias"):
        if gf.get(k) is None:
            continue
        assert g2[k] is g[k], k
        errs[k] = rel_err(N(g2[k]), gf[k] + gb[k])
    errs["dA"] = rel_err(N(g["dA"]), gf["dA"])
    errs["dA_b"] = rel_err(N(g2["dA"]), gb["dA"])
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (name, str(dtype), bad)
    return errs



def check_scan_tm(lib, dev, case, dtype=torch.float32, reverse=False, bidir=False, tol=None, xz_layout=False, backward=True, segments=1):
    """aum_scan_tm_fwd / _bwd (time-serial scan on token-major activations) against the fp64 oracle on the same seeded inputs as the
    channel-major kernels.  xz_layout: u and z are the two
    halves of one (batch, len, 2 dim) tensor (row stride 2 dim), as in_proj leaves them.  segments > 1: the time-segmented launches
    (aum_scan_tm_seg_fwd / _bwd), held to the same oracle and the same tolerances."""
    name, batch, dim, length, dstate, has_z, has_D, has_bias, softplus = case
    d = cases.scan_inputs(*case)
    tol = tol or (TOL_F32 if dtype == torch.float32 else TOL_BF16)
    q = {k: rq(d[k], dtype) for k in ("u", "delta", "z", "B", "C", "dout")}
    rng = np.random.default_rng(7)
    A_b = (d["A"] * np.exp(rng.normal(0, 0.1, d["A"].shape))).astype(np.float32) if bidir else None
    tm = lambda a: None if a is None else T(a, dev, dtype).transpose(1, 2).contiguous()      # (batch, len, X)
    u, delta, z, dout = tm(d["u"]), tm(d["delta"]), tm(d["z"]), tm(d["dout"])
    if xz_layout and z is not None:
        xz = torch.cat([u, z], dim=2).contiguous()
        u, z = xz[:, :, :dim], xz[:, :, dim:]
    Bm, Cm = tm(d["B"]), tm(d["C"])
    bcm = torch.cat([Bm, Cm], dim=2).contiguous()       # one (batch, len, 2N) row like x_dbl's B | C columns
    Bm, Cm = bcm[:, :, :dstate], bcm[:, :, dstate:]
    A, D, bias = T(d["A"], dev), T(d["D"], dev), T(d["delta_bias"], dev)
    ck = aum_hip.scan_tm_ckpt(batch, length, dim, dstate, bidir, dev, lib=lib) if backward else None
    if ck is not None:
        ck.fill_(float("nan"))
    out, out_pre = aum_hip.scan_tm_fwd(u, delta, A, Bm, Cm, D, z, bias, softplus, reverse, T(A_b, dev), want_out_pre=True, ckpt=ck,
                                       lib=lib, segments=segments)
    ref = O.scan_fwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus, reverse, "f64")
    ref_out, ref_pre = ref["out"], ref["y_pre"]
    if bidir:
        rb = O.scan_fwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], softplus, True, "f64")
        ref_out, ref_pre = ref_out + rb["out"], ref_pre + rb["y_pre"]
    cm = lambda t: N(t).transpose(0, 2, 1)
    pairs = {"out": (cm(out), ref_out), "out_pre": (cm(out_pre), ref_pre)}
    out2, none = aum_hip.scan_tm_fwd(u, delta, A, Bm, Cm, D, z, bias, softplus, reverse, T(A_b, dev), lib=lib, segments=segments)    # inference form
    assert none is None
    pairs["out_nopre"] = (cm(out2), ref_out)
    if backward:
        g = aum_hip.scan_tm_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, out_pre if has_z else None, ck, softplus, reverse, T(A_b, dev), lib=lib,
                                segments=segments, want_dA_xA=True)
        # the optional products d A .* A (the gradient of A_log) come out of the same partial-sum launch: exactly dA * A in fp32
        assert torch.equal(g["dA_xA"], g["dA"] * A) and (not bidir or torch.equal(g["dA_b_xA"], g["dA_b"] * T(A_b, dev)))
        gr = O.scan_bwd(q["u"], q["delta"], d["A"], q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus, reverse, "f64")
        if bidir:
            gb = O.scan_bwd(q["u"], q["delta"], A_b, q["B"], q["C"], d["D"], q["z"], d["delta_bias"], q["dout"], softplus, True, "f64")
            for k in ("du", "ddelta", "dB", "dC", "dD", "dz", "ddelta_bias"):
                if gr[k] is not None:
                    gr[k] = gr[k] + gb[k]
            gr["dA_b"] = gb["dA"]
        got = dict(du=cm(g["du"]), ddelta=cm(g["ddelta"]), dz=None if g["dz"] is None else cm(g["dz"]), dA=N(g["dA"]),
                   dA_b=N(g["dA_b"]), dB=N(g["dBC"])[:, :, :dstate].transpose(0, 2, 1), dC=N(g["dBC"])[:, :, dstate:].transpose(0, 2, 1),
                   dD=N(g["dD"]), ddelta_bias=N(g["ddelta_bias"]))
        for k in ("du", "ddelta", "dA", "dA_b", "dB", "dC", "dD", "dz", "ddelta_bias"):
            if gr.get(k) is None:
                assert got.get(k) is None, k
                continue
            pairs[k] = (got[k], gr[k])
    errs = _scan_errors(pairs)
    bad = {k: v for k, v in errs.items() if not (v < tol * (4 if k.split(":")[-1].startswith("d") else 1))}
    assert not bad, (name, str(dtype), "rev" if reverse else "fwd", "bidir" if bidir else "uni", bad, errs)
    return errs


def _tm_rows_vs_oracle(O, b, es, u, dl, z, Bm, Cm, A, A_b, D, bias, dout, softplus=True):
    """fp64 oracle on channels `es` of batch entry `b` of token-major tensors: both directions summed (A_b None: forward only)"""
    f = lambda t: t.float().cpu().numpy()
    rows = lambda t: np.ascontiguousarray(f(t[b][:, es]).T[None])            # (1, rows, L)
    bc = lambda t: np.ascontiguousarray(f(t[b]).T[None])                     # (1, N, L)
    args = (rows(u), rows(dl))
    Aq, Dq, bq = f(A[es]), f(D[es]), f(bias[es])
    zq = rows(z)
    fw = O.scan_fwd(*args, Aq, bc(Bm), bc(Cm), Dq, zq, bq, softplus, False, "f64")
    out, pre = fw["out"], fw["y_pre"]
    gr = None
    if dout is not None:
        gr = O.scan_bwd(*args, Aq, bc(Bm), bc(Cm), Dq, zq, bq, rows(dout), softplus, False, "f64")
    if A_b is not None:
        rb = O.scan_fwd(*args, f(A_b[es]), bc(Bm), bc(Cm), Dq, zq, bq, softplus, True, "f64")
        out, pre = out + rb["out"], pre + rb["y_pre"]
        if dout is not None:
            gb = O.scan_bwd(*args, f(A_b[es]), bc(Bm), bc(Cm), Dq, zq, bq, rows(dout), softplus, True, "f64")
            for k in ("du", "ddelta", "dz", "dB", "dC", "dD", "ddelta_bias"):
                gr[k] = gr[k] + gb[k]
            gr["dA_b"] = gb["dA"]
    return out, pre, gr


def check_scan_tm_grid(lib, dev, Bsz, L, E, rows, entries, chans, split, segments=(1, 1), dtype=torch.bfloat16, dout_mag=1.0):
    """(dtype float16 / dout_mag: the reference's own training precision -- every exps/**/aum-*.sh runs `--mixed_precision=fp16`, and its
    GradScaler hands k_scant_bwd a dout multiplied by 2^16: dout_mag = 16 is a realistic 2^-12 gradient under that scale.  The gradients are
    linear in dout, so the fp64 oracle on the scaled dout IS 2^16 x the unscaled gradient; everything must stay finite at the 16-bit bar.)
    The token-major Fo-Bi scan pair (k_scant_fwd / k_scant_bwd) at a whole launch of (Bsz, L, E), N = 16, bf16, the block's row layouts (z = second half of [x | z] rows, B / C = column blocks of 80-column x_dbl rows), batch-
    distinct random data -- against the ORACLE, not against themselves (the pattern of test_scan_headline_grid_b64):
    (i) sampled (batch entry, channel) rows of out, out_pre, du, ddelta, dz vs the fp64 oracle on exactly those rows, both
    directions summed (lanes 0 / 63, wave and workgroup boundaries, the carries that change waves in the backward's three-stage
    workgroups, the last unit); (ii) dB | dC of sampled batch entries vs the oracle summed over ALL 1 536 channels of the entry;
    (iii) dA, dA_b, dD, ddelta_bias of sampled channels vs the oracle over all 64 batch entries of those channels, and the whole
    tensors vs the sum of eight B = 8 launches (a batch-index error in a checkpoint row, a partial row or a b * stride shows up in
    every one of the three)."""
    torch.manual_seed(5)
    N, R = 16, 48
    bf = lambda t: t.to(dtype)
    xz = bf(torch.randn(Bsz, L, 2 * E, device=dev))
    u, z = bf(torch.randn(Bsz, L, E, device=dev)), xz[:, :, E:]
    dl = bf(0.5 * torch.randn(Bsz, L, E, device=dev))
    x_dbl = bf(torch.randn(Bsz, L, R + 2 * N, device=dev))
    Bm, Cm = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b = A * (1 + 0.1 * torch.rand(E, N, device=dev))
    D, bias = torch.rand(E, device=dev) + 0.5, torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
    dout = bf(dout_mag * torch.randn(Bsz, L, E, device=dev))
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev, dtype=dtype)
    ck.fill_(float("nan"))
    sf, sb = segments           # time segments of the forward / the backward launches (1: the uncut kernels)
    out, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=lib, segments=sf)
    g = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=lib, segments=sb)
    for k, v in g.items():
        assert v is None or bool(torch.isfinite(v).all()), ("not finite", k, str(dtype), dout_mag)
    f = lambda t: t.float().cpu().numpy()
    worst = {}

    def note(k, e, bar):
        worst[k] = max(worst.get(k, 0.0), e)
        assert e < bar, (k, e, bar)
    # (i) units are (batch entry, 64-channel group) numbered batch-major, 24 groups per entry; forward workgroups take 2 units, backward
    # workgroups 3 pairs (X, Y, Z: Y's carries cross waves): sample every residue of the unit number mod 2 and mod 3, lane 0 / 63 of a
    # group, the first and the last unit
    for b, es in rows.items():
        ro, rp, gr = _tm_rows_vs_oracle(O, b, es, u, dl, z, Bm, Cm, A, A_b, D, bias, dout)
        sl = lambda t: f(t[b][:, es]).T[None]
        note("out", rel_err(sl(out), ro), TOL_BF16)
        note("out_pre", rel_err(sl(pre), rp), TOL_BF16)
        note("rms:out", rms_err(sl(out), ro), TOL_BF16)
        for k in ("du", "ddelta", "dz"):
            note(k, rel_err(sl(g[k]), gr[k]), 4 * TOL_BF16)
            note("rms:" + k, rms_err(sl(g[k]), gr[k]), 4 * TOL_BF16)
    # (ii) dB | dC: the sum over all channels and both directions of one batch entry
    allch = list(range(E))
    for b in entries:
        _, _, gr = _tm_rows_vs_oracle(O, b, allch, u, dl, z, Bm, Cm, A, A_b, D, bias, dout)
        got = f(g["dBC"][b])                                   # (L, 2N)
        for k, blk in (("dB", got[:, :N]), ("dC", got[:, N:])):
            ref = gr[k][0].T                                   # (L, N)
            note(k, rel_err(blk, ref), 4 * TOL_BF16)
            note("rms:" + k, rms_err(blk, ref), 4 * TOL_BF16)
            note("state:" + k, rel_err_by(blk, ref, 1), 4 * TOL_BF16)
    # (iii) batch-summed parameter gradients of sampled channels: the oracle over all 64 entries
    es = list(chans)
    acc = None
    for b in range(Bsz):
        _, _, gr = _tm_rows_vs_oracle(O, b, es, u, dl, z, Bm, Cm, A, A_b, D, bias, dout)
        cur = {k: gr[k] for k in ("dA", "dA_b", "dD", "ddelta_bias")}
        acc = cur if acc is None else {k: acc[k] + cur[k] for k in acc}
    for k in acc:
        note(k, rel_err(f(g[k][es]), acc[k]), 4 * TOL_BF16)
        if k in ("dA", "dA_b"):
            note("state:" + k, rel_err_by(f(g[k][es]), acc[k], 1), 4 * TOL_BF16)
    # ... and the whole tensors against eight B = 8 launches (fp32 reassociation only)
    acc8 = {k: torch.zeros_like(g[k]) for k in ("dA", "dA_b", "dD", "ddelta_bias")}
    for b0 in range(0, Bsz, split):
        s8 = lambda t: t[b0:b0 + split]
        ck8 = aum_hip.scan_tm_ckpt(split, L, E, N, True, dev, dtype=dtype)
        o8, p8 = aum_hip.scan_tm_fwd(s8(u), s8(dl), A, s8(Bm), s8(Cm), D, s8(z), bias, True, A_b=A_b, want_out_pre=True, ckpt=ck8, lib=lib,
                                     segments=sf)
        assert torch.equal(o8, out[b0:b0 + split]) and torch.equal(p8, pre[b0:b0 + split]), b0
        g8 = aum_hip.scan_tm_bwd(s8(u), s8(dl), A, s8(Bm), s8(Cm), D, s8(z), bias, s8(dout), p8, ck8, True, A_b=A_b, lib=lib, segments=sb)
        for k in ("du", "ddelta", "dz", "dBC"):
            assert torch.equal(g8[k], g[k][b0:b0 + split]), (b0, k)
        for k in acc8:
            acc8[k] += g8[k]
    for k in acc8:
        assert (acc8[k] - g[k]).abs().max() <= 1e-4 * g[k].abs().max(), k
    return worst




def check_gemm(lib, dev, case, dtype, flags=0):
    """aum_gemm_tn (ABI 9) against an fp64 product of the same 16-bit operands: fp32 accumulation + one rounding -> within one 16-bit ulp
    of the largest result.  MS:185-189 / SSI:517, 540 are F.linear calls in the reference (cuBLAS, same contract)."""
    name, m, n, k, pad_a, pad_c = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    a_full = torch.randn(m, k + pad_a, generator=g).to(dtype).to(dev)
    b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(dev)
    a = a_full[:, :k]
    # the result is a block of a larger buffer: pad_c columns in front of it and 260 rows behind it must come back untouched
    c_buf = torch.full((m + 260, n + pad_c), 7.0, dtype=dtype, device=dev)
    c_full = c_buf[:m]
    out = aum_hip.gemm_tn(a, b, out=c_full[:, pad_c:], lib=lib, flags=flags)
    assert torch.all(c_buf[m:] == 7.0), "rows behind the result were written"
    ref = a.double().cpu() @ b.double().cpu().t()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 1.01 * ulp * ref.abs().max().item() + 1e-30, (name, err, ref.abs().max().item())
    if pad_c:
        assert torch.all(c_full[:, :pad_c] == 7.0), "columns outside the result were written"


def check_gemm_wgrad(lib, dev, t, n, k, splits, dtype, pad_y=0, pad_x=0):
    """aum_gemm_wgrad (ABI 10; autograd of MS:185-189, SSI:563) against an fp64 product of the same 16-bit token-major operands: every
    split's partial tile (its token range, an empty split = zeros) and the summed result; operands may be column slices of wider rows"""
    g = torch.Generator().manual_seed(t * 7 + n + k + splits)
    y_full = (torch.randn(t, n + pad_y, generator=g)).to(dtype).to(dev)
    x_full = (torch.randn(t, k + pad_x, generator=g)).to(dtype).to(dev)
    y, x = y_full[:, pad_y:], x_full[:, :k]
    part = aum_hip.gemm_wgrad(y, x, splits=splits, lib=lib, partials=True)
    assert part.shape == (splits, n, k) and part.dtype == torch.float32
    chunk = (((t + splits - 1) // splits) + 63) // 64 * 64
    yd, xd = y.double().cpu(), x.double().cpu()
    scale = float((yd.t() @ xd).abs().max()) + 1e-30
    for s_ in range(splits):
        t0, t1 = min(s_ * chunk, t), min((s_ + 1) * chunk, t)
        ref = yd[t0:t1].t() @ xd[t0:t1]
        err = (part[s_].double().cpu() - ref).abs().max().item()
        assert err <= 2e-6 * scale * max(1.0, (t1 - t0) ** 0.5 / 8), (t, n, k, splits, s_, err, scale)      # fp32 accumulation of exact 16-bit products
    total = aum_hip.gemm_wgrad(y, x, splits=splits, lib=lib)
    assert rel_err(N(total), (yd.t() @ xd).numpy()) < 1e-5
    assert torch.equal(total, aum_hip.gemm_wgrad(y, x, splits=splits, lib=lib))


def check_decode_kernels(lib, dev, dtype, batch=3, dim=70, dstate=16, width=4):
    """aum_causal_conv1d_update / aum_selective_state_update (ABI 10; MS:313-358) over several consecutive tokens against the reference's
    own fallback expressions (MS:322-327, 343-350) restated in fp64: the caches advanced in place, outputs in the activations' dtype"""
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update, selective_state_update_ref
    g = torch.Generator().manual_seed(17)
    r = lambda *s_: torch.randn(*s_, generator=g)
    w, b = r(dim, width), r(dim)
    A, D, dtb = -torch.rand(dim, dstate, generator=g) - 0.1, r(dim), r(dim) - 3.0
    conv = r(batch, dim, width).to(dev)
    conv_ref = conv.double().cpu()
    st = r(batch, dim, dstate).to(dev)
    st_ref = st.double().cpu()
    st_ref2 = st.clone().cpu()
    tol = 1e-5 if dtype == torch.float32 else TOL_BF16
    for step in range(5):
        x, dt, z = r(batch, dim).to(dtype), (0.5 * r(batch, dim)).to(dtype), r(batch, dim).to(dtype)
        Bm, Cm = r(batch, dstate).to(dtype), r(batch, dstate).to(dtype)
        for silu in (True, False):
            c_in = conv.clone()
            y = aum_hip.conv1d_update(x.to(dev), c_in, w.to(dev), b.to(dev), silu, lib=lib)
            cr = torch.cat([conv_ref[:, :, 1:], x.double()[:, :, None]], dim=2)
            yr = (cr * w.double()).sum(-1) + b.double()
            yr = torch.nn.functional.silu(yr) if silu else yr
            assert y.dtype == dtype and rel_err(N(y), yr.numpy()) < tol, ("conv_update", step, silu)
            assert torch.equal(c_in.cpu().double(), cr.float().double()), "conv window"
        conv, conv_ref = c_in, cr
        out = aum_hip.state_update(st, x.to(dev), dt.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev), D.to(dev), z.to(dev), dtb.to(dev), True, lib=lib)
        dtt = torch.nn.functional.softplus(dt.double() + dtb.double())
        st_ref = st_ref * torch.exp(dtt[:, :, None] * A.double()) + (dtt * x.double())[:, :, None] * Bm.double()[:, None, :]
        yr = ((st_ref * Cm.double()[:, None, :]).sum(-1) + D.double() * x.double()) * torch.nn.functional.silu(z.double())
        assert out.dtype == dtype and rel_err(N(out), yr.numpy()) < tol, ("state_update", step)
        assert rel_err(N(st), st_ref.numpy()) < 1e-5, ("state", step)
        # the package's own torch statement (the public *_ref name) agrees too
        o2 = selective_state_update_ref(st_ref2, x.float(), dt.float(), A, Bm.float(), Cm.float(), D, z.float(), dtb, True)
        assert rel_err(o2.numpy(), yr.numpy()) < 1e-4
    # no gate, no skip, no bias, no softplus
    x, dt = r(batch, dim).to(dtype), (0.1 * r(batch, dim)).abs().to(dtype)
    Bm, Cm = r(batch, dstate).to(dtype), r(batch, dstate).to(dtype)
    s0 = st.clone()
    out = aum_hip.state_update(st, x.to(dev), dt.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev), lib=lib)
    sr = s0.double().cpu() * torch.exp(dt.double()[:, :, None] * A.double()) + (dt.double() * x.double())[:, :, None] * Bm.double()[:, None, :]
    assert rel_err(N(out), (sr * Cm.double()[:, None, :]).sum(-1).numpy()) < tol


def check_gemm_args(lib, dev):
    """argument rules of aum_gemm_tn (csrc/gemm_args.h): refused shapes return an error code, nothing is launched"""
    a = torch.zeros(8, 64, dtype=torch.bfloat16, device=dev)
    for bad in (torch.zeros(128, 64, dtype=torch.bfloat16, device=dev),          # n % 256
                torch.zeros(256, 32, dtype=torch.bfloat16, device=dev),          # k mismatch / k % 64
                torch.zeros(256, 64, dtype=torch.float16, device=dev)):          # dtype mismatch
        assert not aum_hip.gemm_tn_supported(a, bad)
        with pytest.raises(RuntimeError):
            aum_hip.gemm_tn(a, bad, lib=lib)
    assert not aum_hip.gemm_tn_supported(a.float(), torch.zeros(256, 64, device=dev))


def check_dtproj(lib, dev, ntok, dim, rank, ncols, dtype):
    """aum_dtproj_tm_fwd (ABI 9; SSI:468) against an fp64 product of the same 16-bit operands: x_dbl rows of `ncols` columns whose first
    `rank` are the dt block (the rest -- B and C -- must not be read into the product)"""
    g = torch.Generator().manual_seed(ntok * 131 + dim + rank)
    x = torch.randn(ntok, ncols, generator=g).to(dtype).to(dev)
    w = (torch.randn(dim, rank, generator=g) / rank ** 0.5).to(dtype).to(dev)
    out = aum_hip.dtproj_tm_fwd(x, rank, w, lib=lib)
    ref = x[:, :rank].double().cpu() @ w.double().cpu().t()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.double().cpu() - ref).abs().max().item()
    assert out.shape == (ntok, dim) and err <= 1.01 * ulp * ref.abs().max().item() + 1e-30, (ntok, dim, rank, err)


def check_xdt_bwd(lib, dev, ntok, dim, dtype, pad=0):
    """aum_xdt_tm_bwd (ABI 10; SSI:570-574, 587, 590): dx_dbl[:, :48] against an fp64 product of the 16-bit operands (one rounding),
    dx_dbl[:, 48:] = the fp32 dB | dC rows rounded, du against fp64 (du_in + the kernel's OWN rounded dx_dbl . W_x) rounded once -- what
    the reference's GEMM, copy and addmm compute.  pad: extra columns behind ddelta / du rows (pitch > dim)."""
    g = torch.Generator().manual_seed(ntok * 5 + dim)
    R, C = 48, 80
    ddf = torch.randn(ntok, dim + pad, generator=g).to(dtype).to(dev)
    duf = torch.randn(ntok, dim + pad, generator=g).to(dtype).to(dev)
    ddelta, du = ddf[:, :dim], duf[:, :dim]
    dbc = torch.randn(ntok, C - R, generator=g).to(dev)
    wdt_t = (torch.randn(R, dim, generator=g) / dim ** 0.5).to(dtype).to(dev)
    wx_t = (torch.randn(dim, C, generator=g) / C ** 0.5).to(dtype).to(dev)
    du_in = du.double().cpu()
    tail_in = duf[:, dim:].clone()
    dx = aum_hip.xdt_tm_bwd(ddelta, dbc, wdt_t, wx_t, du, lib=lib)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ref_r = ddelta.double().cpu() @ wdt_t.double().cpu().t()
    er = (dx[:, :R].double().cpu() - ref_r).abs().max().item()
    assert dx.shape == (ntok, C) and er <= 1.01 * ulp * ref_r.abs().max().item(), ("dx_dbl dt block", ntok, dim, er)
    assert torch.equal(dx[:, R:].cpu(), dbc.to(dtype).cpu()), "dx_dbl B | C block"
    ref_u = du_in + dx.double().cpu() @ wx_t.double().cpu().t()
    eu = (du.double().cpu() - ref_u).abs().max().item()
    assert eu <= 1.01 * ulp * ref_u.abs().max().item(), ("du", ntok, dim, eu)
    assert torch.equal(duf[:, dim:], tail_in), "columns behind the du rows were written"


def check_xdt(lib, dev, ntok, dim, rank, dtype, ncols=80):
    """aum_xdt_tm_fwd (ABI 9; SSI:467-468): x_dbl against an fp64 product of the 16-bit operands (one rounding), delta against an fp64
    product of the kernel's OWN rounded x_dbl (what two separate GEMMs compute).  ncols: 80 (AuM-Base rows) or 56 (AuM-Small)."""
    g = torch.Generator().manual_seed(ntok * 7 + dim + rank + ncols)
    u = torch.randn(ntok, dim, generator=g).to(dtype).to(dev)
    wx = (torch.randn(ncols, dim, generator=g) / dim ** 0.5).to(dtype).to(dev)
    wdt = (torch.randn(dim, rank, generator=g) / rank ** 0.5).to(dtype).to(dev)
    x_dbl, delta = aum_hip.xdt_tm_fwd(u, wx, wdt, lib=lib)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ref_x = u.double().cpu() @ wx.double().cpu().t()
    ex = (x_dbl.double().cpu() - ref_x).abs().max().item()
    assert x_dbl.shape == (ntok, ncols) and ex <= 1.01 * ulp * ref_x.abs().max().item(), ("x_dbl", ntok, dim, rank, ex)
    ref_d = x_dbl[:, :rank].double().cpu() @ wdt.double().cpu().t()
    ed = (delta.double().cpu() - ref_d).abs().max().item()
    assert delta.shape == (ntok, dim) and ed <= 1.01 * ulp * ref_d.abs().max().item(), ("delta", ntok, dim, rank, ed)


def check_cast_bank(lib, device):
    """aum_cast_bank (ABI 13): the 16-bit copies and transposes of a group of fp32 matrices, bit-equal to Tensor.to() / .t().contiguous()
    (round-to-nearest-even), at the model's shapes (in / out / x / dt projections of AuM-Base: ragged 64-tiles in rows = 80 and cols = 48),
    subnormal / huge / signed-zero / tie values included; argument rules"""
    import ctypes
    torch.manual_seed(0)
    shapes = [(80, 1536), (1536, 48), (768, 1536), (3072, 768), (8, 4), (72, 68)] if device == "cuda" else [(80, 132), (136, 48), (8, 4)]
    for dtype in (torch.bfloat16, torch.float16):
        for shape in shapes:
            n = 3 if shape[0] * shape[1] > 100000 else 5
            ps = [torch.randn(shape, device=device) * (10.0 ** (i - 2)) for i in range(n)]
            ps[0].view(-1)[:8] = torch.tensor([0.0, -0.0, 1e-41, -3e38, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -9, 65520.0, 6e-8], device=device)   # ties, range ends
            assert aum_hip.cast_bank_supported(ps, dtype)
            for want_t in (False, True):
                bank, bank_t = aum_hip.cast_bank(ps, dtype, want_t, lib=lib)
                assert bank.shape == (n,) + shape and bank.dtype == dtype and (bank_t is None) == (not want_t)
                for i, p in enumerate(ps):
                    ref = p.to(dtype)
                    assert torch.equal(bank[i].view(torch.int16), ref.view(torch.int16)), (shape, dtype, i)
                    if want_t:
                        assert bank_t.shape == (n, shape[1], shape[0])
                        assert torch.equal(bank_t[i].view(torch.int16), ref.t().contiguous().view(torch.int16)), (shape, dtype, i)
    p = torch.randn(8, 4, device=device)
    tab = torch.tensor([p.data_ptr()], dtype=torch.int64, device=device)
    out = torch.empty(8, 4, dtype=torch.bfloat16, device=device)
    st = lib.stream(p)
    assert lib.c.aum_cast_bank(tab.data_ptr(), 1, 8, 4, out.data_ptr(), None, 1, st) == 0
    assert lib.c.aum_cast_bank(tab.data_ptr(), 1, 4, 8, out.data_ptr(), None, 1, st) != 0         # rows % 8
    assert lib.c.aum_cast_bank(tab.data_ptr(), 1, 8, 6, out.data_ptr(), None, 1, st) != 0         # cols % 4
    assert lib.c.aum_cast_bank(tab.data_ptr(), 1, 8, 4, out.data_ptr(), None, 0, st) != 0         # fp32 is not a 16-bit type
    assert lib.c.aum_cast_bank(None, 1, 8, 4, out.data_ptr(), None, 1, st) != 0
    assert not aum_hip.cast_bank_supported([torch.randn(6, 4, device=device)], torch.bfloat16)
