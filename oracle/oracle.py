"""ORACLE -- test infrastructure only.

numpy front-end over oracle/_build/libaum_oracle.so (plain C, built by `make -C oracle`), the CPU
restatement of the reference's hot-path arithmetic.  Allowed importers: tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never the product path (audio-mamba-aum_amd/*).

Pinned against the reference: tests/golden/*.npz hold outputs of the reference's own
selective_scan_ref / mamba_inner_ref / bimamba_inner_ref / rms_norm_ref (and the MS:272 conv
expression) produced by tests/golden/make_golden.py importing /root/reference in the build
container; tests/test_oracle_golden.py checks this module against them.

Citations: SSI = vim-mamba_ssm/mamba_ssm/ops/selective_scan_interface.py,
MS = vim-mamba_ssm/mamba_ssm/modules/mamba_simple.py, LN = vim-mamba_ssm/mamba_ssm/ops/triton/layernorm.py
(relative to /root/reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libaum_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("aum_oracle.c", "aum_oracle_impl.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def num_threads():
    return int(lib().aum_oracle_num_threads())


_DT = {"f32": (np.float32, ctypes.c_float), "f64": (np.float64, ctypes.c_double)}


def _arr(a, npdt):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=npdt))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _bc3(Bm):
    """Accept (batch, dstate, len) or (batch, 1, dstate, len) (SSI:31-36)."""
    Bm = np.asarray(Bm)
    if Bm.ndim == 4:
        assert Bm.shape[1] == 1, "oracle restates the G=1 case only"
        Bm = Bm[:, 0]
    return Bm


def scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, reverse=False,
             prec="f32"):
    """selective_scan_ref (SSI:86-152).  Returns dict(out, y_pre, last_state)."""
    npdt, cdt = _DT[prec]
    u, delta, A = _arr(u, npdt), _arr(delta, npdt), _arr(A, npdt)
    B, C = _arr(_bc3(B), npdt), _arr(_bc3(C), npdt)
    D, z, delta_bias = _arr(D, npdt), _arr(z, npdt), _arr(delta_bias, npdt)
    batch, dim, length = u.shape
    dstate = A.shape[1]
    assert A.shape == (dim, dstate) and B.shape == (batch, dstate, length) and C.shape == B.shape
    y_pre = np.empty_like(u)
    out = np.empty_like(u)
    last = np.empty((batch, dim, dstate), dtype=npdt)
    getattr(lib(), "aum_oracle_scan_fwd_" + prec)(
        _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), int(bool(delta_softplus)),
        int(bool(reverse)), batch, dim, length, dstate, _p(y_pre), _p(out), _p(last))
    return {"out": out, "y_pre": y_pre, "last_state": last}


def scan_bc_abs_sums(u, delta, A, B, C, z, delta_bias, dout, delta_softplus=False, reverse=False):
    """Diagnostic for tests (numpy fp64, small shapes): S_B[b,n,t] = sum_d |g_n,d,t delta_d,t u_d,t| and S_C[b,n,t] = sum_d |dy_d,t x_n,d,t|,
    the sums of the MAGNITUDES of the terms of dB / dC (same recurrences as scan_bwd: SSI:86-152 and its adjoint).  A reduction that
    rounds every term to bf16 before adding (the HIP kernels' matrix-pipe channel sums for 16-bit activations) is within 2^-8 S of
    the exact sum whatever the cancellation; tests bound dB / dC by that on top of the parity tolerance."""
    f = lambda a: None if a is None else np.asarray(a, np.float64)
    u, delta, A, B, C, z, delta_bias, dout = f(u), f(delta), f(A), f(_bc3(B)), f(_bc3(C)), f(z), f(delta_bias), f(dout)
    batch, dim, length = u.shape
    dstate = A.shape[1]
    dt = delta + (delta_bias[None, :, None] if delta_bias is not None else 0.0)
    if delta_softplus:
        dt = np.where(dt > 20, dt, np.log1p(np.exp(np.minimum(dt, 20))))
    dy = dout if z is None else dout * z / (1.0 + np.exp(-z))
    order = range(length - 1, -1, -1) if reverse else range(length)
    a = np.exp(dt[:, :, None, :] * A[None, :, :, None])                     # (b, d, n, t)
    x = np.zeros((batch, dim, dstate))
    xs = np.zeros((batch, dim, dstate, length))
    for t in order:
        x = a[:, :, :, t] * x + (dt[:, :, t] * u[:, :, t])[:, :, None] * B[:, None, :, t]
        xs[:, :, :, t] = x
    S_C = np.abs(dy[:, :, None, :] * xs).sum(1)
    g = np.zeros((batch, dim, dstate))
    S_B = np.zeros((batch, dstate, length))
    a_next = np.zeros((batch, dim, dstate))
    for t in reversed(list(order)):
        g = dy[:, :, t][:, :, None] * C[:, None, :, t] + a_next * g
        S_B[:, :, t] = np.abs(g * (dt[:, :, t] * u[:, :, t])[:, :, None]).sum(1)
        a_next = a[:, :, :, t]
    return S_B, S_C


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False, reverse=False, prec="f32"):
    """Analytic adjoint of scan_fwd (what selective_scan_cuda.bwd returns, SSI:62-65)."""
    npdt, cdt = _DT[prec]
    u, delta, A = _arr(u, npdt), _arr(delta, npdt), _arr(A, npdt)
    B, C = _arr(_bc3(B), npdt), _arr(_bc3(C), npdt)
    D, z, delta_bias, dout = _arr(D, npdt), _arr(z, npdt), _arr(delta_bias, npdt), _arr(dout, npdt)
    batch, dim, length = u.shape
    dstate = A.shape[1]
    g = {
        "du": np.zeros_like(u), "ddelta": np.zeros_like(u), "dA": np.zeros_like(A),
        "dB": np.zeros_like(B), "dC": np.zeros_like(C), "dD": np.zeros(dim, npdt),
        "dz": np.zeros_like(u), "ddelta_bias": np.zeros(dim, npdt),
    }
    getattr(lib(), "aum_oracle_scan_bwd_" + prec)(
        _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), _p(dout),
        int(bool(delta_softplus)), int(bool(reverse)), batch, dim, length, dstate,
        _p(g["du"]), _p(g["ddelta"]), _p(g["dA"]), _p(g["dB"]), _p(g["dC"]), _p(g["dD"]),
        _p(g["dz"]) if z is not None else None, _p(g["ddelta_bias"]))
    if D is None:
        g["dD"] = None
    if z is None:
        g["dz"] = None
    if delta_bias is None:
        g["ddelta_bias"] = None
    return g


def conv1d_fwd(x, weight, bias=None, silu=True, reverse=False, prec="f32"):
    """act(conv1d(x)[..., :L]) of MS:272.  weight: (dim, W) or (dim, 1, W)."""
    npdt, cdt = _DT[prec]
    x = _arr(x, npdt)
    weight = _arr(np.asarray(weight).reshape(x.shape[1], -1), npdt)
    bias = _arr(bias, npdt)
    batch, dim, length = x.shape
    y = np.empty_like(x)
    getattr(lib(), "aum_oracle_conv1d_fwd_" + prec)(
        _p(x), _p(weight), _p(bias), int(bool(silu)), int(bool(reverse)), batch, dim, length,
        weight.shape[1], _p(y))
    return y


def conv1d_bwd(x, weight, bias, dy, silu=True, reverse=False, prec="f32"):
    npdt, cdt = _DT[prec]
    x, dy = _arr(x, npdt), _arr(dy, npdt)
    wshape = np.asarray(weight).shape
    weight = _arr(np.asarray(weight).reshape(x.shape[1], -1), npdt)
    bias = _arr(bias, npdt)
    batch, dim, length = x.shape
    dx = np.zeros_like(x)
    dw = np.zeros_like(weight)
    db = np.zeros(dim, npdt)
    getattr(lib(), "aum_oracle_conv1d_bwd_" + prec)(
        _p(x), _p(weight), _p(bias), _p(dy), int(bool(silu)), int(bool(reverse)), batch, dim, length,
        weight.shape[1], _p(dx), _p(dw), _p(db))
    return {"dx": dx, "dweight": dw.reshape(wshape), "dbias": db if bias is not None else None}


def rmsnorm_fwd(x, weight, bias=None, residual=None, eps=1e-5, prec="f32"):
    """rms_norm_ref(upcast=True) (LN:35-48) + the fused kernel's saved tensors.  x: (..., cols)."""
    npdt, cdt = _DT[prec]
    shape = np.asarray(x).shape
    x2 = _arr(np.asarray(x).reshape(-1, shape[-1]), npdt)
    r2 = None if residual is None else _arr(np.asarray(residual).reshape(-1, shape[-1]), npdt)
    weight, bias = _arr(weight, npdt), _arr(bias, npdt)
    rows, cols = x2.shape
    y = np.empty_like(x2)
    res_out = np.empty_like(x2)
    rstd = np.empty(rows, npdt)
    getattr(lib(), "aum_oracle_rmsnorm_fwd_" + prec)(
        _p(x2), _p(r2), _p(weight), _p(bias), cdt(eps), rows, cols, _p(y), _p(res_out), _p(rstd))
    return {"y": y.reshape(shape), "residual_out": res_out.reshape(shape), "rstd": rstd}


def rmsnorm_bwd(dy, residual_out, weight, rstd, dresidual_out=None, has_bias=False, prec="f32"):
    """LN:254-277 (rms branch).  Returns dx (= gradient for x and for residual), dweight, dbias."""
    npdt, cdt = _DT[prec]
    shape = np.asarray(dy).shape
    dy2 = _arr(np.asarray(dy).reshape(-1, shape[-1]), npdt)
    x2 = _arr(np.asarray(residual_out).reshape(-1, shape[-1]), npdt)
    dr2 = None if dresidual_out is None else _arr(np.asarray(dresidual_out).reshape(-1, shape[-1]), npdt)
    weight, rstd = _arr(weight, npdt), _arr(rstd, npdt)
    rows, cols = dy2.shape
    dx = np.empty_like(dy2)
    dw = np.zeros(cols, npdt)
    db = np.zeros(cols, npdt)
    getattr(lib(), "aum_oracle_rmsnorm_bwd_" + prec)(
        _p(dy2), _p(dr2), _p(x2), _p(weight), _p(rstd), rows, cols, _p(dx), _p(dw),
        _p(db) if has_bias else None)
    return {"dx": dx.reshape(shape), "dweight": dw, "dbias": db if has_bias else None}


# ----------------------------------------------------------------------------------------------
# Inner block: conv -> x_proj -> dt_proj -> scan(s) -> (out_proj).  Restates mamba_inner_ref
# (SSI:636-670), bimamba_inner_ref (SSI:673-709) and the v2 composition of MS:214-246.
# ----------------------------------------------------------------------------------------------

def _inner_pre(xz, conv_w, conv_b, x_proj_w, dt_proj_w, d_state, reverse, prec):
    npdt = _DT[prec][0]
    xz = np.asarray(xz, npdt)
    batch, two_e, L = xz.shape
    E = two_e // 2
    R = np.asarray(dt_proj_w).shape[1]
    x, z = xz[:, :E], xz[:, E:]                                        # SSI:645
    xc = conv1d_fwd(x, conv_w, conv_b, silu=True, reverse=reverse, prec=prec)   # SSI:646
    xc_t = xc.transpose(0, 2, 1).reshape(batch * L, E)                  # 'b d l -> (b l) d'
    x_dbl = xc_t @ np.asarray(x_proj_w, npdt).T                         # SSI:650
    delta = (np.asarray(dt_proj_w, npdt) @ x_dbl[:, :R].T).reshape(E, batch, L).transpose(1, 0, 2)  # SSI:651-652
    Bm = x_dbl[:, R:R + d_state].reshape(batch, L, d_state).transpose(0, 2, 1)   # SSI:654-658
    Cm = x_dbl[:, R + d_state:R + 2 * d_state].reshape(batch, L, d_state).transpose(0, 2, 1)
    return dict(x=np.ascontiguousarray(x), z=np.ascontiguousarray(z), xc=xc, x_dbl=x_dbl,
                delta=np.ascontiguousarray(delta), B=np.ascontiguousarray(Bm), C=np.ascontiguousarray(Cm))


def inner_no_out_proj_fwd(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, reverse=False,
                          prec="f32"):
    """mamba_inner_ref minus out_proj (the MambaInnerFnNoOutProj semantic, SSI:155-224); `reverse`
    folds the xz.flip(-1) ... .flip(-1) sandwich of MS:229-246."""
    st = _inner_pre(xz, conv_w, conv_b, x_proj_w, dt_proj_w, np.asarray(A).shape[1], reverse, prec)
    r = scan_fwd(st["xc"], st["delta"], A, st["B"], st["C"], D, st["z"], delta_bias, True, reverse, prec)
    st["out_z"] = r["out"]
    return st


def inner_fwd(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, D, delta_bias,
              A_b=None, prec="f32"):
    """mamba_inner_ref (A_b None) or bimamba_inner_ref (SSI:673-709).  Returns dict incl. 'out' (B,L,Dm)."""
    npdt = _DT[prec][0]
    st = _inner_pre(xz, conv_w, conv_b, x_proj_w, dt_proj_w, np.asarray(A).shape[1], False, prec)
    y = scan_fwd(st["xc"], st["delta"], A, st["B"], st["C"], D, st["z"], delta_bias, True, False, prec)["out"]
    if A_b is not None:  # SSI:707-708
        y = y + scan_fwd(st["xc"], st["delta"], A_b, st["B"], st["C"], D, st["z"], delta_bias, True, True,
                         prec)["out"]
    st["out_z"] = y
    out = y.transpose(0, 2, 1) @ np.asarray(out_proj_w, npdt).T          # SSI:709
    if out_proj_b is not None:
        out = out + np.asarray(out_proj_b, npdt)
    st["out"] = out
    return st


def inner_bwd(st, dout_z, xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, A_b=None,
              reverse=False, prec="f32"):
    """Adjoint of the chain conv -> x_proj -> dt_proj -> scan(s) w.r.t. out_z (B,E,L).  This is the
    mathematically complete gradient = autograd through *_inner_ref (includes the
    reverse-direction dz term the reference's fused backward drops at SSI:560/599)."""
    npdt = _DT[prec][0]
    xz = np.asarray(xz, npdt)
    batch, two_e, L = xz.shape
    E = two_e // 2
    R = np.asarray(dt_proj_w).shape[1]
    N = np.asarray(A).shape[1]
    x_proj_w = np.asarray(x_proj_w, npdt)
    dt_proj_w = np.asarray(dt_proj_w, npdt)
    g = scan_bwd(st["xc"], st["delta"], A, st["B"], st["C"], D, st["z"], delta_bias, dout_z, True,
                 reverse if A_b is None else False, prec)
    grads = {"dA": g["dA"], "dD": g["dD"], "ddelta_bias": g["ddelta_bias"]}
    if A_b is not None:
        gb = scan_bwd(st["xc"], st["delta"], A_b, st["B"], st["C"], D, st["z"], delta_bias, dout_z, True,
                      True, prec)
        for k in ("du", "ddelta", "dB", "dC", "dz"):
            g[k] = g[k] + gb[k]
        grads["dD"] = g["dD"] + gb["dD"]
        grads["ddelta_bias"] = g["ddelta_bias"] + gb["ddelta_bias"]
        grads["dA_b"] = gb["dA"]
    dxc = g["du"]
    dx_dbl = np.zeros_like(st["x_dbl"])
    dx_dbl[:, R:R + N] = g["dB"].transpose(0, 2, 1).reshape(batch * L, N)
    dx_dbl[:, R + N:R + 2 * N] = g["dC"].transpose(0, 2, 1).reshape(batch * L, N)
    ddelta2 = g["ddelta"].transpose(1, 0, 2).reshape(E, batch * L)         # 'b d l -> d (b l)'
    grads["ddt_proj_w"] = ddelta2 @ st["x_dbl"][:, :R]
    dx_dbl[:, :R] = ddelta2.T @ dt_proj_w
    xc_t = st["xc"].transpose(0, 2, 1).reshape(batch * L, E)
    grads["dx_proj_w"] = dx_dbl.T @ xc_t
    dxc = dxc + (dx_dbl @ x_proj_w).reshape(batch, L, E).transpose(0, 2, 1)
    cg = conv1d_bwd(st["x"], conv_w, conv_b, dxc, silu=True, reverse=reverse, prec=prec)
    grads["dconv_w"], grads["dconv_b"] = cg["dweight"], cg["dbias"]
    grads["dxz"] = np.concatenate([cg["dx"], g["dz"]], axis=1)
    return grads


def inner_full_bwd(st, dout, xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, D,
                   delta_bias, A_b=None, prec="f32"):
    """Adjoint of inner_fwd w.r.t. its (B,L,Dm) output."""
    npdt = _DT[prec][0]
    dout = np.asarray(dout, npdt)
    batch, L, Dm = dout.shape
    out_proj_w = np.asarray(out_proj_w, npdt)
    dout2 = dout.reshape(batch * L, Dm)
    dout_z = (dout2 @ out_proj_w).reshape(batch, L, -1).transpose(0, 2, 1)
    grads = inner_bwd(st, np.ascontiguousarray(dout_z), xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D,
                      delta_bias, A_b, False, prec)
    oz_t = st["out_z"].transpose(0, 2, 1).reshape(batch * L, -1)
    grads["dout_proj_w"] = dout2.T @ oz_t
    grads["dout_proj_b"] = dout2.sum(0) if out_proj_b is not None else None
    return grads
