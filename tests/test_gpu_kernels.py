"""GPU parity tests proper: the product library libaum_hip.so (hand-written gfx950 kernels, called through the
C ABI) against the oracle on the same seeded inputs.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

import aum_hip
import cases
import kernel_checks as KC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need a GPU"
    return aum_hip.get()     # raises ImportError if the extension is missing: no fallback


def test_wave_scan_primitive(lib):
    KC.check_wave_scan(lib, "cuda")


@pytest.mark.parametrize("case", cases.SCAN_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("path", ["workgroup", "generic"])
def test_scan_f32(lib, case, mode, path):
    if mode == "bidir" and case[3] > lib.max_single_pass_len:
        pytest.skip("direction fusion is single-pass only")
    KC.check_scan(lib, "cuda", case, torch.float32, reverse=(mode == "rev"), bidir=(mode == "bidir"),
                  generic=(path == "generic"))


@pytest.mark.parametrize("case", ["l65", "l513", "l2049", "l130_n4"])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_scan_16bit(lib, case, mode, dtype):
    c = [x for x in cases.SCAN_CASES if x[0] == case][0]
    if mode == "bidir" and c[3] > lib.max_single_pass_len:
        pytest.skip("direction fusion is single-pass only")
    KC.check_scan(lib, "cuda", c, dtype, reverse=(mode == "rev"), bidir=(mode == "bidir"),
                  tol=1e-2 if dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("case", cases.SCAN_WIDE_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_one_row_backward_vs_row_pair(lib, case, mode, dtype):
    for rowpair in (False, True):
        KC.check_scan(lib, "cuda", case, dtype, reverse=(mode == "rev"), bidir=(mode == "bidir"), rowpair=rowpair)


@pytest.mark.parametrize("case", [c for c in cases.SCAN_CASES if c[0] == "l513"] + cases.SCAN_ROW_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_scan_row_kernels_lane_checkpoint(lib, case, mode, dtype):
    """L = 513, scan_row_kernels.h: forward fills the lane-entry checkpoint, backward reads it (no recomputed forward scan)"""
    kw = dict(reverse=(mode == "rev"), bidir=(mode == "bidir"), lane_ckpt=True, tol=2e-3 if dtype == torch.float16 else None)
    KC.check_scan(lib, "cuda", case, dtype, **kw)
    KC.check_scan(lib, "cuda", case, dtype, strided=True, **kw)


def test_scan_row_kernels_full_size(lib):
    """AuM-Base block shape (B=8 of the 64, E=1536, L=513, N=16, bf16, d-major rows): the checkpointed row-kernel backward
    against the previous-generation backward (which recomputes the forward scan) on the same inputs -- same gradients up to
    fp32 reassociation -- and bitwise repeatable from launch to launch (no atomics; race screen for the tile updates)"""
    torch.manual_seed(0)
    Bsz, E, L, N = 8, 1536, 513, 16
    dt = torch.bfloat16
    mk = lambda: torch.randn(E, Bsz, L, device="cuda").to(dt).permute(1, 0, 2)
    u, z, dout = mk(), mk(), mk()
    delta = (0.5 * torch.randn(E, Bsz, L, device="cuda")).to(dt).permute(1, 0, 2)
    Bm, Cm = torch.randn(Bsz, 1, N, L, device="cuda").to(dt), torch.randn(Bsz, 1, N, L, device="cuda").to(dt)
    A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device="cuda"))
    A_b = A * 1.05
    D, bias = torch.ones(E, device="cuda"), torch.full((E,), -4.0, device="cuda") + torch.rand(E, device="cuda")
    for kw in (dict(A_b=A_b), dict(reverse=False), dict(reverse=True)):
        bidir = "A_b" in kw
        ck = aum_hip.scan_lane_ckpt(u, N, bidir, lib=lib)
        ck.fill_(float("nan"))
        out, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, want_out_pre=True, x_lane=ck, lib=lib, **kw)
        out_rp, pre_rp, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, want_out_pre=True, rowpair=True, lib=lib, **kw)
        assert torch.isfinite(ck).all()
        assert (out.float() - out_rp.float()).abs().max() <= 2e-2 * out_rp.float().abs().max()
        g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, x_lane=ck, lib=lib, **kw)
        g2 = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, x_lane=ck, lib=lib, **kw)
        gold = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, lib=lib, **kw)
        for k, v in g.items():
            if v is None:
                continue
            assert torch.equal(v, g2[k]), (kw.keys(), k)
            a, b = v.float(), gold[k].float()
            assert (a - b).abs().max() <= 1e-2 * b.abs().max() + 1e-6, (list(kw), k, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("case", cases.SCAN_LONG_CASES + [c for c in cases.SCAN_CASES if c[0] == "l2049"], ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_chunked_one_row_backward(lib, case, mode, dtype):
    for rowpair in (False, True):
        KC.check_scan(lib, "cuda", case, dtype, reverse=(mode == "rev"), rowpair=rowpair)
    KC.check_scan(lib, "cuda", case, dtype, reverse=(mode == "rev"), strided=True)
    if case[3] % 512 == 1:
        KC.check_scan(lib, "cuda", case, dtype, reverse=(mode == "rev"), ckpt=True)


@pytest.mark.parametrize("case", [c for c in cases.SCAN_LONG_CASES if c[3] % 512 == 1] + [c for c in cases.SCAN_CASES if c[0] == "l2049"],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_second_direction_accumulates(lib, case, dtype):
    KC.check_scan_accumulate(lib, "cuda", case, dtype)


def test_scan_long_form_full_size(lib):
    """long-form shape (BASELINE config 5: B=2 of the 8, E=1536, L=4097, N=16, bf16, d-major rows): the chunked one-row
    backward against the row-pair kernels on the same inputs, and bitwise repeatable from launch to launch"""
    torch.manual_seed(0)
    Bsz, E, L, N = 2, 1536, 4097, 16
    dt = torch.bfloat16
    mk = lambda: torch.randn(E, Bsz, L, device="cuda").to(dt).permute(1, 0, 2)
    u, z, dout = mk(), mk(), mk()
    delta = (0.5 * torch.randn(E, Bsz, L, device="cuda")).to(dt).permute(1, 0, 2)
    Bm, Cm = torch.randn(Bsz, 1, N, L, device="cuda").to(dt), torch.randn(Bsz, 1, N, L, device="cuda").to(dt)
    A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device="cuda"))
    D, bias = torch.ones(E, device="cuda"), torch.full((E,), -4.0, device="cuda") + torch.rand(E, device="cuda")
    for reverse in (False, True):
        _, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, reverse, want_out_pre=True, lib=lib)
        g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, reverse, lib=lib)
        g2 = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, reverse, lib=lib)
        ck = aum_hip.scan_ckpt(u, N, lib=lib)
        aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, reverse, want_out_pre=True, x_ck=ck, lib=lib)
        gc = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, reverse, x_ck=ck, lib=lib)
        gp = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, reverse, rowpair=True, lib=lib)
        for k, v in g.items():
            if v is None:
                continue
            assert torch.equal(v, g2[k]), (reverse, k)
            c = gc[k].float()      # the forward's checkpoint and the backward's pre-pass hold the same states up to rounding
            assert (c - v.float()).abs().max() <= 5e-3 * v.float().abs().max() + 1e-6, ("checkpoint", reverse, k)
            a, b = v.float(), gp[k].float()
            assert (a - b).abs().max() <= 2e-2 * b.abs().max() + 1e-6, (reverse, k, float((a - b).abs().max()), float(b.abs().max()))


def test_scan_strided_layout(lib):
    c = [x for x in cases.SCAN_CASES if x[0] == "l65"][0]
    KC.check_scan(lib, "cuda", c, torch.float32, bidir=True, strided=True)
    KC.check_scan(lib, "cuda", c, torch.bfloat16, bidir=False, strided=True)


@pytest.mark.parametrize("case", cases.CONV_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("reverse", [False, True])
def test_conv(lib, case, reverse):
    KC.check_conv(lib, "cuda", case, torch.float32, reverse=reverse)
    KC.check_conv(lib, "cuda", case, torch.float32, reverse=reverse, silu=False)
    KC.check_conv(lib, "cuda", case, torch.bfloat16, reverse=reverse)


@pytest.mark.parametrize("case", cases.NORM_CASES, ids=lambda c: c[0])
def test_norm(lib, case):
    KC.check_norm(lib, "cuda", case, torch.float32)
    KC.check_norm(lib, "cuda", case, torch.bfloat16, torch.float32)
    KC.check_norm(lib, "cuda", case, torch.float16, torch.float32)


def test_scan_full_size_properties(lib):
    """AuM-Base scan shape (B=8 of the 64, E=1536, L=513, N=16, bf16): size-independent properties instead of the
    oracle -- (1) the fused bidirectional call equals the sum of a forward-time and a reverse-time call,
    (2) linearity in u for fixed delta (the recurrence is linear in u), (3) the reverse call equals flip/forward/flip."""
    torch.manual_seed(0)
    Bsz, E, L, N = 8, 1536, 513, 16
    dev, dt = "cuda", torch.bfloat16
    u = torch.randn(Bsz, E, L, device=dev).to(dt)
    delta = (0.5 * torch.randn(Bsz, E, L, device=dev)).to(dt)
    z = torch.randn(Bsz, E, L, device=dev).to(dt)
    Bm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
    Cm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1)
    A_b = A * 1.1
    D = torch.ones(E, device=dev)
    bias = torch.full((E,), -4.0, device=dev)
    f = lambda **kw: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, None, bias, True, lib=lib, **kw)[0].float()
    o_bi = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, None, bias, True, A_b=A_b, lib=lib)[0].float()
    o_f = f()
    o_r = aum_hip.scan_fwd(u, delta, A_b, Bm, Cm, D, None, bias, True, reverse=True, lib=lib)[0].float()
    scale = o_bi.abs().max()
    assert ((o_f + o_r) - o_bi).abs().max() / scale < 2e-2          # each term rounded to bf16 separately
    fl = lambda t: t.flip([-1]).contiguous()
    o_r2 = aum_hip.scan_fwd(fl(u), fl(delta), A_b, fl(Bm), fl(Cm), D, None, bias, True, lib=lib)[0].float().flip([-1])
    assert (o_r - o_r2).abs().max() / scale < 1e-2
    o2 = aum_hip.scan_fwd((2 * u.float()).to(dt), delta, A, Bm, Cm, D, None, bias, True, lib=lib)[0].float()
    assert (o2 - 2 * o_f).abs().max() / scale < 2e-2
    # gated output: out == out_pre * silu(z)
    out, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, lib=lib)
    ref = pre.float() * torch.nn.functional.silu(z.float())
    assert (out.float() - ref).abs().max() / ref.abs().max() < 1e-2


def test_generic_conv_and_norm_kernels(lib):
    """the any-width / any-cols kernels stay covered now that width 4 / cols <= 2048 take the vectorised kernels"""
    for case in cases.CONV_CASES:
        KC.check_conv(lib, "cuda", case, torch.float32, reverse=False, generic=True)
        KC.check_conv(lib, "cuda", case, torch.float32, reverse=True, generic=True)
    for case in cases.NORM_CASES:
        KC.check_norm(lib, "cuda", case, torch.float32, generic=True)


@pytest.mark.parametrize("case", cases.PROJ_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_proj(lib, case, dtype):
    KC.check_proj(lib, "cuda", case, dtype)


def test_proj_full_size(lib):
    """AuM-Base block: d_inner 1536, dt_rank 48, d_state 16, 64 clips x 513 tokens (513 token tiles, 11 token splits)"""
    KC.check_proj(lib, "cuda", ("base_full", 1536, 48, 16, 64, 513), torch.bfloat16)
    KC.check_proj(lib, "cuda", ("base_b3", 1536, 48, 16, 3, 513), torch.bfloat16)        # ntok = 1539: ragged everything


def test_scan_backward_bitwise_repeatable(lib):
    """no atomics anywhere in the scan path (partials + a fixed-order reduce): the same launch gives the same bits, and a
    read-add-write race between waves of the 12/16-wave workgroups would show up here as a flipped bit sooner or later"""
    torch.manual_seed(0)
    Bsz, E, L, N = 4, 384, 513, 16
    dt = torch.bfloat16
    mk = lambda: torch.randn(E, Bsz, L, device="cuda").to(dt).permute(1, 0, 2)
    u, z, dout = mk(), mk(), mk()
    delta = (0.5 * torch.randn(E, Bsz, L, device="cuda")).to(dt).permute(1, 0, 2)
    Bm, Cm = torch.randn(Bsz, 1, N, L, device="cuda").to(dt), torch.randn(Bsz, 1, N, L, device="cuda").to(dt)
    A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1)
    D, bias = torch.ones(E, device="cuda"), torch.full((E,), -4.0, device="cuda")
    for A_b in (A * 1.05, None):
        _, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, lib=lib)
        ref = None
        for it in range(5):
            g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b, lib=lib)
            cur = {k: v.clone() for k, v in g.items() if v is not None}
            if ref is None:
                ref = cur
            for k in ref:
                assert torch.equal(ref[k], cur[k]), (it, k)


@pytest.mark.gpu
def test_scan_headline_grid_b64(lib):
    """The bench's own launch shape -- B = 64, E = 1536, L = 513, N = 16, bf16, channel-major rows: 98 304 rows, 1 024 backward
    workgroups -- checked instead of only timed.  (i) sampled rows of the per-(batch, channel) outputs (out, du, ddelta, dz) against
    the fp64 oracle run on exactly those rows (a row's result depends only on its own u, delta, z and its batch entry's B, C);
    (ii) dB / dC of sampled batch entries against a B = 1 launch on that entry (batch independence); (iii) the batch-summed dA, dA_b,
    dD, ddelta_bias against the sum of eight B = 8 launches; (iv) bitwise repeatable."""
    from oracle import oracle as O
    torch.manual_seed(3)
    Bsz, E, L, N = 64, 1536, 513, 16
    dt = torch.bfloat16
    mk = lambda s=1.0: (s * torch.randn(E, Bsz, L, device="cuda")).to(dt).permute(1, 0, 2)
    u, z, dout, delta = mk(), mk(), mk(), mk(0.5)
    Bm, Cm = torch.randn(Bsz, 1, N, L, device="cuda").to(dt), torch.randn(Bsz, 1, N, L, device="cuda").to(dt)
    A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device="cuda"))
    A_b = A * (1 + 0.1 * torch.rand(E, N, device="cuda"))
    D, bias = torch.rand(E, device="cuda") + 0.5, torch.full((E,), -4.0, device="cuda") + torch.rand(E, device="cuda")
    out, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, lib=lib)
    g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b, lib=lib)
    g2 = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b, lib=lib)
    for k, v in g.items():
        assert v is None or torch.equal(v, g2[k]), k
    f = lambda t: t.float().cpu().numpy()
    # (i) sampled rows: first / last workgroup of a batch entry, a wave boundary, the last batch entry
    rows = [(0, 0), (0, 95), (0, 96), (17, 700), (31, 1535), (63, 0), (63, 1535), (40, 1000)]
    for b in sorted({r[0] for r in rows}):
        es = [e for (bb, e) in rows if bb == b]
        sl = lambda t: f(t[b:b + 1, es])
        args = (sl(u), sl(delta))
        Aq, Abq, Dq, bq = f(A[es]), f(A_b[es]), f(D[es]), f(bias[es])
        Bq, Cq = f(Bm[b:b + 1, 0]), f(Cm[b:b + 1, 0])
        rf = O.scan_fwd(*args, Aq, Bq, Cq, Dq, sl(z), bq, True, False, "f64")
        rb = O.scan_fwd(*args, Abq, Bq, Cq, Dq, sl(z), bq, True, True, "f64")
        assert KC.rel_err(sl(out), rf["out"] + rb["out"]) < KC.TOL_BF16
        gf = O.scan_bwd(*args, Aq, Bq, Cq, Dq, sl(z), bq, sl(dout), True, False, "f64")
        gb = O.scan_bwd(*args, Abq, Bq, Cq, Dq, sl(z), bq, sl(dout), True, True, "f64")
        for k in ("du", "ddelta", "dz"):
            assert KC.rel_err(sl(g[k]), gf[k] + gb[k]) < 4 * KC.TOL_BF16, (b, k)
    # (ii) dB / dC of a batch entry do not depend on the other entries (a B = 1 launch groups the rows differently: fp32 reassociation)
    for b in (0, 29, 63):
        s1 = lambda t: t[b:b + 1]
        o1, p1, _ = aum_hip.scan_fwd(s1(u), s1(delta), A, s1(Bm), s1(Cm), D, s1(z), bias, True, A_b=A_b, want_out_pre=True, lib=lib)
        g1 = aum_hip.scan_bwd(s1(u), s1(delta), A, s1(Bm), s1(Cm), D, s1(z), bias, s1(dout), p1, True, A_b=A_b, lib=lib)
        assert torch.equal(o1, out[b:b + 1])
        for k in ("dB", "dC"):
            assert (g1[k] - g[k][b:b + 1]).abs().max() <= 1e-5 * g1[k].abs().max(), (b, k)
    # (iii) batch-summed parameter gradients: the same sum taken over eight B = 8 launches (fp32 reassociation only)
    acc = {k: torch.zeros_like(g[k]) for k in ("dA", "dA_b", "dD", "ddelta_bias")}
    for b0 in range(0, Bsz, 8):
        s8 = lambda t: t[b0:b0 + 8]
        g8 = aum_hip.scan_bwd(s8(u), s8(delta), A, s8(Bm), s8(Cm), D, s8(z), bias, s8(dout), s8(pre), True, A_b=A_b, lib=lib)
        for k in acc:
            acc[k] += g8[k]
    for k in acc:
        assert (acc[k] - g[k]).abs().max() <= 1e-4 * g[k].abs().max(), k


@pytest.mark.gpu
def test_sum_rows(lib):
    """aum_sum_rows at the callers' shapes: every launch geometry of the dispatcher, fp32 and 16-bit partials, fixed summation order"""
    torch.manual_seed(0)
    for shape, dt in (((4096, 768), torch.float32), ((2048, 768), torch.float32), ((1024, 4, 48), torch.float32), ((42, 1536, 80), torch.float32), ((42, 48, 1536), torch.float32),
                      ((4, 3072, 768), torch.bfloat16), ((8, 768, 1536), torch.bfloat16), ((2, 192, 8), torch.float16), ((5, 40), torch.float32),
                      ((86, 16), torch.float32)):
        t = torch.randn(shape, device="cuda").to(dt)
        got = aum_hip.sum_rows(t, lib=lib)
        ref = t.double().sum(0)
        assert got.dtype == torch.float32 and got.shape == t.shape[1:]
        assert (got.double() - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()) * shape[0] ** 0.5, (shape, dt)
        assert torch.equal(got, aum_hip.sum_rows(t, lib=lib))


@pytest.mark.gpu
def test_sum_rows_multi(lib):
    """aum_sum_rows_multi (ABI 11) at the two callers' shapes -- the conv weight + bias partials of aum_conv1d_tm_bwd at the bench batch, the skinny
    dt_proj / x_proj partial sets (the second stored transposed) -- against fp64 sums; every row grouping; repeatable bit for bit"""
    torch.manual_seed(0)
    nparts = int(lib.c.aum_conv1d_tm_nparts(64, 513))
    cases = [([(nparts, 1536, 4), (nparts, 1536)], [0, 0]), ([(42, 1536, 48), (42, 1536, 80)], [0, 80]), ([(3, 8)], [0]), ([(2, 256 * 512 * 8)], [0]),
             ([(7, 64, 16), (9, 40), (33, 24, 8), (64, 8)], [16, 0, 8, 0]), ([(600, 768), (600, 16)], [0, 0]),
             ([(14, 768, 1536), (42, 1536, 48), (42, 1536, 80), (nparts, 1536, 4), (nparts, 1536)], [0, 0, 80, 0, 0]),      # a layer's backward
             ([(2 + q, 8 * (q + 1)) for q in range(aum_hip.SUM_MAX_JOBS)], [0] * aum_hip.SUM_MAX_JOBS)]
    for shapes, tr in cases:
        parts = [torch.randn(sh, device="cuda") for sh in shapes]
        got = aum_hip.sum_rows_multi(parts, tr, lib=lib)
        again = aum_hip.sum_rows_multi(parts, tr, lib=lib)
        for t, g, g2, tc in zip(parts, got, again, tr):
            ref = t.double().sum(0)
            ref = ref.t() if tc else ref
            assert g.shape == ref.shape and g.is_contiguous() and g.dtype == torch.float32
            assert (g.double() - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()) * t.shape[0] ** 0.5, (shapes, tr)
            assert torch.equal(g, g2)
            one = aum_hip.sum_rows(t, lib=lib)                       # a job's additions are those of a launch of its own
            assert torch.equal(g, one.t().contiguous() if tc else one)


@pytest.mark.gpu
def test_cast_bank(lib):
    KC.check_cast_bank(lib, "cuda")


# ---- time-serial token-major kernels --------------------------------------------------------------------------------------------
def test_wave_sum_butterflies(lib):
    """the masked-DPP / permlane-swap butterflies on the real lanes (the emulator states their result, not their data movement)"""
    KC.check_wave_sum32(lib, "cuda")


@pytest.mark.parametrize("case", cases.SCAN_TM_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_scan_tm(lib, case, mode, dtype):
    for xz in ((False, True) if dtype == torch.bfloat16 else (False,)):
        KC.check_scan_tm(lib, "cuda", case, dtype, reverse=(mode == "rev"), bidir=(mode == "bidir"), xz_layout=xz, backward=True,
                         tol=2e-3 if dtype == torch.float16 else None)


@pytest.mark.parametrize("case", cases.CONV_TM_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv_tm(lib, case, reverse, dtype):
    for silu, xz in ((True, False), (True, True), (False, False)):
        KC.check_conv_tm(lib, "cuda", case, dtype, reverse, silu, xz)


def test_scan_tm_headline_shape(lib):
    """The bench's own launch -- B = 64, E = 1536, L = 513, N = 16, bf16, [x | z] rows -- checked, not only timed: (i) forward and
    backward are bitwise repeatable (every hand-over through memory -- the Fo-Bi partials, the carries that change waves, the
    register-free global->LDS loads -- is ordered by a barrier or a counted wait; one that names too few operations shows up as a
    run-to-run difference); (ii) the direction pair equals the sum of two one-direction launches (which the small-shape parity
    tests tie to the oracle) within the rounding of the bf16 partial hand-over."""
    torch.manual_seed(0)
    Bsz, L, E, N = 64, 513, 1536, 16
    dev = "cuda"
    xz = torch.randn(Bsz, L, 2 * E, device=dev).bfloat16()
    u, z = torch.randn(Bsz, L, E, device=dev).bfloat16(), xz[:, :, E:]
    dl = (0.5 * torch.randn(Bsz, L, E, device=dev)).bfloat16()
    bc = torch.randn(Bsz, L, 48 + 2 * N, device=dev).bfloat16()
    Bm, Cm = bc[:, :, 48:48 + N], bc[:, :, 48 + N:]
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b, D, bias = A * 1.05, torch.ones(E, device=dev), torch.full((E,), -4.0, device=dev)
    dout = torch.randn(Bsz, L, E, device=dev).bfloat16()
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
    ref_f = ref_b = None
    for it in range(3):
        o, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=lib)
        g = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=lib)
        cur_f = (o.clone(), pre.clone())
        cur_b = {k: v.clone() for k, v in g.items() if v is not None and not k.startswith("_")}
        if ref_f is None:
            ref_f, ref_b = cur_f, cur_b
        assert torch.equal(ref_f[0], cur_f[0]) and torch.equal(ref_f[1], cur_f[1]), it
        for k in ref_b:
            assert torch.equal(ref_b[k], cur_b[k]), (it, k)
            assert torch.isfinite(cur_b[k].float()).all(), k
    ckf, ckb = aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, dev), aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, dev)
    _, pf = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, want_out_pre=True, ckpt=ckf, lib=lib)
    _, pb = aum_hip.scan_tm_fwd(u, dl, A_b, Bm, Cm, D, z, bias, True, reverse=True, want_out_pre=True, ckpt=ckb, lib=lib)
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
    assert rel(ref_f[1], pf.float() + pb.float()) < 1e-2
    gf = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, ref_f[1], ckf, True, lib=lib)
    gb = aum_hip.scan_tm_bwd(u, dl, A_b, Bm, Cm, D, z, bias, dout, ref_f[1], ckb, True, reverse=True, lib=lib)
    for k in ("du", "ddelta", "dBC", "dD", "ddelta_bias"):
        assert rel(ref_b[k], gf[k].float() + gb[k].float()) < 2e-2, k
    assert rel(ref_b["dA"], gf["dA"]) < 2e-2 and rel(ref_b["dA_b"], gb["dA"]) < 2e-2 and rel(ref_b["dz"], gf["dz"]) < 2e-2


def test_scan_tm_headline_grid_b64(lib):
    """the dominant kernels (k_scant_fwd / k_scant_bwd) at the bench's own launch -- B = 64, E = 1536, L = 513,
    N = 16, bf16, the block's row layouts, batch-distinct random data -- against the fp64 ORACLE (KC.check_scan_tm_grid: sampled rows
    of out / out_pre / du / ddelta / dz, dB | dC of whole batch entries, dA / dA_b / dD / ddelta_bias of sampled channels over all 64
    entries, the whole parameter gradients against eight B = 8 launches).  Units are (batch entry, 64-channel group) numbered
    batch-major, 24 groups per entry; forward workgroups take 2 units, backward workgroups 3 pairs (X, Y, Z: Y's carries cross
    waves): the sample holds every residue of the unit number mod 2 and mod 3, lanes 0 / 63 of a group, the first and the last unit."""
    rows = {0: [0, 63, 64, 127, 128, 700], 1: [5, 64 * 7 + 1, 1535], 17: [64 * 3 + 63, 64 * 4, 64 * 5 + 17], 31: [64 * 11, 1472],
            40: [1000, 1001], 62: [64 * 23 + 62, 3], 63: [0, 777, 1535]}
    worst = KC.check_scan_tm_grid(lib, "cuda", 64, 513, 1536, rows, (0, 29, 63), [0, 63, 64, 500, 1023, 1535], 8)
    print("scan_tm headline grid, worst errors:", {k: float("%.3g" % v) for k, v in sorted(worst.items())})


@pytest.mark.parametrize("dout_mag", [2.0 ** -12, 16.0], ids=["unscaled", "gradscaler_2p16"])
def test_scan_tm_headline_grid_b64_fp16(lib, dout_mag):
    """the reference trains in fp16 + GradScaler (`--mixed_precision=fp16`, exps/audioset/aum-base_scratch-audioset.sh:54).
    The bench launch (B = 64, E = 1536, L = 513) in float16 -- packed fp16 state checkpoints and 16-bit partial hand-overs with 5
    exponent bits -- against the fp64 oracle at the 16-bit bar, once with a realistic unscaled gradient magnitude (2^-12: the
    backward's outputs live among fp16's subnormals, so only the forward and finiteness are held to the bar there) and once with that
    gradient as GradScaler's initial scale 2^16 delivers it (16): every gradient finite and equal to 2^16 x the unscaled fp64 gradient
    at the bar, i.e. nothing overflows on the way and GradScaler has no inf to skip on at this magnitude."""
    rows = {0: [0, 63, 64, 700], 17: [64 * 3 + 63, 64 * 4], 40: [1000, 1001], 63: [0, 777, 1535]}
    if dout_mag < 1:
        # unscaled: du / ddelta / dz of a 2^-12 gradient are fp16 subnormals (absolute step 2^-24 ~ 2.4e-4 of 2^-12): a relative bar means
        # nothing there -- this is exactly why the reference scales.  Held here: the forward at the bar and finite gradients.
        import aum_hip
        torch.manual_seed(5)
        Bsz, L, E, N = 64, 513, 1536, 16
        h = lambda t: t.half()
        u, z, dl = h(torch.randn(Bsz, L, E, device="cuda")), h(torch.randn(Bsz, L, E, device="cuda")), h(0.5 * torch.randn(Bsz, L, E, device="cuda"))
        Bm, Cm = h(torch.randn(Bsz, L, N, device="cuda")), h(torch.randn(Bsz, L, N, device="cuda"))
        A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1)
        D, bias = torch.ones(E, device="cuda"), torch.full((E,), -4.0, device="cuda")
        dout = h(dout_mag * torch.randn(Bsz, L, E, device="cuda"))
        ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, "cuda", dtype=torch.float16)
        out, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A * 1.1, want_out_pre=True, ckpt=ck, lib=lib)
        g = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A * 1.1, lib=lib)
        assert bool(torch.isfinite(out).all()) and all(v is None or bool(torch.isfinite(v).all()) for v in g.values())
        # the fp32 parameter sums do not underflow: 2^16 x them equals the scaled run's sums up to the 16-bit inputs' rounding
        g16 = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, h(dout.float() * 65536.0), pre, ck, True, A_b=A * 1.1, lib=lib)
        for k in ("dA", "dA_b", "dD", "ddelta_bias", "dBC"):
            a, b = g[k].float() * 65536.0, g16[k].float()
            assert (a - b).abs().max() <= 2e-2 * b.abs().max(), (k, float((a - b).abs().max()), float(b.abs().max()))
        return
    worst = KC.check_scan_tm_grid(lib, "cuda", 64, 513, 1536, rows, (0, 63), [0, 63, 64, 1535], 8, dtype=torch.float16, dout_mag=dout_mag)
    print("scan_tm headline grid fp16, dout x 2^16, worst errors:", {k: float("%.3g" % v) for k, v in sorted(worst.items())})


@pytest.mark.parametrize("case", [c for c in cases.SCAN_TM_CASES if c[3] >= 9], ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("segments", [2, 3, 5])
def test_scan_tm_segments(lib, case, mode, segments):
    """aum_scan_tm_seg_fwd / _bwd on the GPU: rows cut into time segments (carry pass + main pass per direction, the adjoint the same
    way), same fp64 oracle and tolerances as the uncut launches; ragged and empty ranges, all three dtypes"""
    for dt, xz in ((torch.float32, False), (torch.bfloat16, True), (torch.float16, False)):
        KC.check_scan_tm(lib, "cuda", case, dt, reverse=(mode == "rev"), bidir=(mode == "bidir"), xz_layout=xz, backward=True, segments=segments)


def test_scan_tm_longform_grid_b8(lib):
    """SURVEY section 8 config 5 (long-form clips: B = 8, 128 x 8192 frames -> L = 4097 tokens, AuM-Base widths) on the time-segmented
    token-major launches the dispatch takes there (16 ranges of 264 steps: aum_hip.scan_tm_segments) against the fp64 ORACLE
    on whole 4097-step rows: sampled (entry, channel) rows of out / out_pre / du / ddelta / dz, dB | dC of whole batch entries,
    parameter gradients of sampled channels over all entries, and the batch-split identity."""
    import aum_hip
    sf, sb = aum_hip.scan_tm_segments(8, 1536, 4097, True, False), aum_hip.scan_tm_segments(8, 1536, 4097, True, True)
    assert sf > 1 and sb > 1, (sf, sb)
    rows = {0: [0, 63, 64, 700], 3: [5, 64 * 7 + 1, 1535], 7: [64 * 23 + 62, 3, 1000]}
    worst = KC.check_scan_tm_grid(lib, "cuda", 8, 4097, 1536, rows, (0, 7), [0, 63, 64, 1535], 4, segments=(sf, sb))
    print("scan_tm long-form grid, segments", (sf, sb), "worst errors:", {k: float("%.3g" % v) for k, v in sorted(worst.items())})


def test_scan_tm_small_inference_grid_b64(lib):
    """BASELINE config 2 at its bench batch on the dispatch the bench takes: AuM-Small, B = 64, E = 768, bf16,
    forward only -- token-major, 56-column x_dbl rows (dt rank 24 | B | C): x_dbl and delta from the fused aum_xdt_tm_fwd (round 4: its
    56-column form; before, a library GEMM + aum_dtproj_tm_fwd, still checked here), the scan without out_pre / checkpoints.  Both
    projection paths vs fp64 products at that shape; sampled rows of the Fo-Bi scan output vs the fp64 oracle."""
    O = KC.O
    torch.manual_seed(6)
    Bsz, L, E, N, R = 64, 513, 768, 16, 24
    dev = "cuda"
    bf = lambda t: t.bfloat16()
    xz = bf(torch.randn(Bsz, L, 2 * E, device=dev))
    u, z = bf(torch.randn(Bsz, L, E, device=dev)), xz[:, :, E:]
    x_dbl = bf(torch.randn(Bsz, L, R + 2 * N, device=dev))
    w_dt = bf(torch.randn(E, R, device=dev) / R ** 0.5 * 0.5)
    w_x = bf(torch.randn(R + 2 * N, E, device=dev) / E ** 0.5)
    u2 = u.reshape(-1, E)
    assert aum_hip.xdt_tm_supported(u2, w_x, w_dt), "the fused x/dt kernel takes AuM-Small's 56-column rows"
    xf, df = aum_hip.xdt_tm_fwd(u2, w_x, w_dt, lib=lib)
    rx = u2.double() @ w_x.double().t()
    assert xf.shape == (Bsz * L, R + 2 * N) and (xf.double() - rx).abs().max().item() <= 1.01 * 2.0 ** -8 * rx.abs().max().item()
    rd = xf[:, :R].double() @ w_dt.double().t()
    assert (df.double() - rd).abs().max().item() <= 1.01 * 2.0 ** -8 * rd.abs().max().item()
    x2 = x_dbl.reshape(-1, R + 2 * N)
    assert aum_hip.dtproj_tm_supported(x2, R, w_dt)
    dl = aum_hip.dtproj_tm_fwd(x2, R, w_dt, lib=lib).reshape(Bsz, L, E)
    ref = x2[:, :R].double() @ w_dt.double().t()
    err = (dl.reshape(-1, E).double() - ref).abs().max().item()
    assert err <= 1.01 * 2.0 ** -8 * ref.abs().max().item(), err
    Bm, Cm = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b = A * (1 + 0.1 * torch.rand(E, N, device=dev))
    D, bias = torch.rand(E, device=dev) + 0.5, torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
    out, none = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, lib=lib)
    assert none is None
    out2, _ = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, lib=lib)
    assert torch.equal(out, out2)
    f = lambda t: t.float().cpu().numpy()
    rows = {0: [0, 63, 64, 767], 1: [1, 700], 33: [64 * 5, 64 * 5 + 63, 400], 63: [0, 383, 767]}
    for b, es in rows.items():
        ro, _, _ = KC._tm_rows_vs_oracle(O, b, es, u, dl, z, Bm, Cm, A, A_b, D, bias, None)
        got = f(out[b][:, es]).T[None]
        assert KC.rel_err(got, ro) < KC.TOL_BF16 and KC.rms_err(got, ro) < KC.TOL_BF16, (b, KC.rel_err(got, ro), KC.rms_err(got, ro))


def test_conv_tm_headline_shape(lib):
    """the block's conv at B = 64, E = 1536, L = 513 on [x | z] rows: sampled channels against the oracle, dweight / dbias against an
    fp64 torch statement of the same sums, bitwise repeatable (fixed-order partial sums, no atomics)"""
    O = KC.O
    torch.manual_seed(1)
    Bsz, L, E = 64, 513, 1536
    xz = torch.randn(Bsz, L, 2 * E, device="cuda").bfloat16()
    x = xz[:, :, :E]
    w, b = 0.5 * torch.randn(E, 4, device="cuda"), 0.2 * torch.randn(E, device="cuda")
    dy = torch.randn(Bsz, L, E, device="cuda").bfloat16()
    y = aum_hip.conv1d_tm_fwd(x, w, b, True, lib=lib)
    dx, dw, db = aum_hip.conv1d_tm_bwd(x, w, b, dy, True, lib=lib)
    dx2, dw2, db2 = aum_hip.conv1d_tm_bwd(x, w, b, dy, True, lib=lib)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    ch = [0, 7, 511, 512, 1000, 1535]
    bs = [0, 31, 63]
    xs = x[bs][:, :, ch].float().cpu().numpy().transpose(0, 2, 1)
    dys = dy[bs][:, :, ch].float().cpu().numpy().transpose(0, 2, 1)
    ws, bb = w[ch].cpu().numpy(), b[ch].cpu().numpy()
    ry = O.conv1d_fwd(xs, ws, bb, True, False, "f64")
    rg = O.conv1d_bwd(xs, ws, bb, dys, True, False, "f64")
    assert KC.rel_err(y[bs][:, :, ch].float().cpu().numpy().transpose(0, 2, 1), ry) < KC.TOL_BF16
    assert KC.rel_err(dx[bs][:, :, ch].float().cpu().numpy().transpose(0, 2, 1), rg["dx"]) < 4 * KC.TOL_BF16
    # dw / db over ALL batch entries for the sampled channels: the oracle on the full batch of those channels
    xa = x[:, :, ch].float().cpu().numpy().transpose(0, 2, 1)
    dya = dy[:, :, ch].float().cpu().numpy().transpose(0, 2, 1)
    ra = O.conv1d_bwd(xa, ws, bb, dya, True, False, "f64")
    assert KC.rel_err(dw[ch].cpu().numpy(), ra["dweight"]) < 1e-3 and KC.rel_err(db[ch].cpu().numpy(), ra["dbias"]) < 1e-3


@pytest.mark.parametrize("case", cases.GEMM_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sched", ["auto", "lockstep", "pipelined", "paced"])
def test_gemm_tn(lib, case, dtype, sched):
    flags = {"auto": 0, "lockstep": aum_hip.GEMM_LOCKSTEP, "pipelined": aum_hip.GEMM_PIPELINED, "paced": aum_hip.GEMM_PACED}[sched]
    if sched == "paced" and case[3] < 448:          # the paced-store kernel needs seven K-steps per tile: refused, nothing launched
        a = torch.zeros(case[1], case[3], dtype=dtype, device="cuda")
        with pytest.raises(RuntimeError):
            aum_hip.gemm_tn(a, torch.zeros(case[2], case[3], dtype=dtype, device="cuda"), lib=lib, flags=flags)
        return
    KC.check_gemm(lib, "cuda", case, dtype, flags=flags)


@pytest.mark.parametrize("case", cases.GEMM_WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_wgrad(lib, case, dtype):
    KC.check_gemm_wgrad(lib, "cuda", *case[:4], dtype, *case[4:])


def test_gemm_wgrad_full_size(lib):
    """the bench's two weight-gradient GEMMs (64 x 513 tokens; d W_in = dxz^T . hidden: 36 tiles x 7 splits, d W_out = dout^T . out_z: 18 x 14)
    against fp64 on sampled rows of the result, with the operands laid out as the block has them, bitwise repeatable"""
    t = 64 * 513
    for n, k in ((3072, 768), (768, 1536), (1536, 48), (1536, 80)):          # the last two: the skinny kernel (dt_proj / x_proj weight gradients)
        torch.manual_seed(n + k)
        y = (torch.randn(t, n, device="cuda") * 0.1).to(torch.bfloat16)
        x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
        out = aum_hip.gemm_wgrad(y, x, lib=lib)
        rows = torch.cat([torch.arange(0, 40, device="cuda"), torch.randint(0, n, (60,), device="cuda"), torch.arange(n - 40, n, device="cuda")])
        ref = y[:, rows].double().t() @ x.double()
        assert (out[rows].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (n, k)
        for _ in range(2):
            assert torch.equal(out, aum_hip.gemm_wgrad(y, x, lib=lib))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_decode_kernels(lib, dtype):
    """streaming inference, one token per call (MS:313-358): conv window update and single-step state update on the device"""
    KC.check_decode_kernels(lib, "cuda", dtype)
    KC.check_decode_kernels(lib, "cuda", dtype, batch=64, dim=1536)


def test_gemm_tn_argument_rules(lib):
    KC.check_gemm_args(lib, "cuda")


@pytest.mark.parametrize("shape", cases.GEMM_FULL_CASES, ids=lambda s: "x".join(map(str, s)))
def test_gemm_tn_full_size(lib, shape):
    """the bench's own projection GEMMs (64 x 513 tokens: 128 full row blocks + one of 64 rows; 1548 / 387 / 774 workgroups) on sampled
    rows against an fp64 product, bitwise repeatable from launch to launch"""
    m, n, k = shape
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    out = aum_hip.gemm_tn(a, b, lib=lib)
    rows = torch.cat([torch.arange(0, 257, device="cuda"), torch.randint(0, m, (300,), device="cuda"), torch.arange(m - 257, m, device="cuda")])
    ref = a[rows].double() @ b.double().t()
    assert (out[rows].double() - ref).abs().max().item() <= 1.01 * 2.0 ** -8 * ref.abs().max().item()
    for _ in range(3):
        assert torch.equal(out, aum_hip.gemm_tn(a, b, lib=lib))
    # every schedule: the same sums in the same order (the default here is the paced-store kernel of round 6: AGPR accumulators, a tile's
    # stores under the next tile's K-steps, buffer stores that drop the ragged block's rows) -- and nothing written outside the result
    for fl in (aum_hip.GEMM_LOCKSTEP, aum_hip.GEMM_PIPELINED, aum_hip.GEMM_PACED):
        buf = torch.full((m + 5, n + 16), 3.0, device="cuda", dtype=torch.bfloat16)
        o = buf[:m, 8:8 + n]
        if fl == aum_hip.GEMM_PACED and k < 448:         # the pacing needs seven K-steps of 64: named on a shorter product it is refused
            with pytest.raises(RuntimeError, match="UNSUPPORTED"):
                aum_hip.gemm_tn(a, b, out=o, lib=lib, flags=fl)
            continue
        for _ in range(2):
            aum_hip.gemm_tn(a, b, out=o, lib=lib, flags=fl)
            assert torch.equal(o, out), fl
        assert bool((buf[m:] == 3.0).all()) and bool((buf[:, :8] == 3.0).all()) and bool((buf[:, 8 + n:] == 3.0).all()), fl


def test_norm_headline_shape(lib):
    """the block's fused add + RMSNorm at the bench's own launch: 64 x 513 = 32 832 rows of d_model 768, 16-bit branch output on an fp32
    residual stream (prenorm, LN:254-277): sampled rows against the oracle, dweight (the sum over ALL rows) against an fp64 torch
    statement of the same sum, bitwise repeatable (fixed-order partial sums)"""
    O = KC.O
    torch.manual_seed(2)
    rows, cols = 64 * 513, 768
    x = torch.randn(rows, cols, device="cuda").bfloat16()
    res = 2.0 * torch.randn(rows, cols, device="cuda")
    w = 1.0 + 0.2 * torch.randn(cols, device="cuda")
    dy = torch.randn(rows, cols, device="cuda").bfloat16()
    dres = torch.randn(rows, cols, device="cuda")
    y, rstd, res_out = aum_hip.rmsnorm_fwd(x, w, res, 1e-5, residual_dtype=torch.float32, lib=lib)
    dx, dw, dres_in = aum_hip.rmsnorm_bwd(dy, res_out, w, rstd, dres, True, x_dtype=torch.bfloat16, lib=lib)
    dx2, dw2, dres_in2 = aum_hip.rmsnorm_bwd(dy, res_out, w, rstd, dres, True, x_dtype=torch.bfloat16, lib=lib)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(dres_in, dres_in2)
    pick = torch.cat([torch.arange(0, 40), torch.randint(0, rows, (200,)), torch.arange(rows - 40, rows)]).cuda()
    n = lambda t: t[pick].float().cpu().numpy()
    r = O.rmsnorm_fwd(n(x), w.cpu().numpy(), None, n(res), 1e-5, "f64")
    assert KC.rel_err(n(y), r["y"]) < KC.TOL_BF16 and KC.rel_err(n(res_out), r["residual_out"]) < 1e-6
    assert KC.rel_err(rstd[pick].cpu().numpy(), r["rstd"]) < 1e-5
    rb = O.rmsnorm_bwd(n(dy), n(res_out), w.cpu().numpy(), rstd[pick].cpu().numpy(), n(dres), False, "f64")
    assert KC.rel_err(n(dres_in), rb["dx"]) < 1e-5                      # the fp32 gradient of the residual stream
    assert KC.rel_err(n(dx), rb["dx"]) < 4 * KC.TOL_BF16                # the same values rounded for the 16-bit branch
    xhat = res_out.double() * rstd.double()[:, None]
    assert KC.rel_err(dw.cpu().numpy(), (dy.double() * xhat).sum(0).cpu().numpy()) < 1e-4


@pytest.mark.parametrize("case", cases.DTPROJ_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dtproj_tm(lib, case, dtype):
    KC.check_dtproj(lib, "cuda", *case, dtype)


def test_dtproj_tm_headline_shape(lib):
    """the bench's own launch: 64 x 513 tokens, d_inner 1536, dt_rank 48, x_dbl rows of 80 -- every element against the library's fp32
    product of the same operands (one rounding each), bitwise repeatable"""
    torch.manual_seed(3)
    x = torch.randn(64 * 513, 80, device="cuda").bfloat16()
    w = (torch.randn(1536, 48, device="cuda") / 48 ** 0.5).bfloat16()
    out = aum_hip.dtproj_tm_fwd(x, 48, w, lib=lib)
    ref = x[:, :48].float() @ w.float().t()
    assert (out.float() - ref).abs().max().item() <= 1.01 * 2.0 ** -8 * ref.abs().max().item()
    assert torch.equal(out, aum_hip.dtproj_tm_fwd(x, 48, w, lib=lib))


def test_gemm_tn_random_shapes(lib):
    """40 seeded shapes (m 1 .. 1500 incl. every kind of last row block, n 256 .. 1024, k 64 .. 512), every kernel of aum_gemm_tn: the
    product against fp64, and nothing written outside the result (rows behind it, columns beside it)"""
    import random
    rnd = random.Random(20260928)
    for it in range(40):
        m = rnd.choice([rnd.randint(1, 1500), 256 * rnd.randint(1, 5) + rnd.choice([0, 1, 64, 127, 128, 129, 255])])
        n, k = 256 * rnd.randint(1, 4), 64 * rnd.randint(1, 12)
        pad_a, pad_c = 8 * rnd.randint(0, 3), 8 * rnd.randint(0, 3)
        flags = rnd.choice([0, aum_hip.GEMM_LOCKSTEP, aum_hip.GEMM_PIPELINED] + ([aum_hip.GEMM_PACED] if k >= 448 else []))
        KC.check_gemm(lib, "cuda", (f"rnd{it}_{m}_{n}_{k}", m, n, k, pad_a, pad_c), torch.bfloat16, flags=flags)


@pytest.mark.parametrize("case", cases.XDT_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_xdt_tm(lib, case, dtype):
    KC.check_xdt(lib, "cuda", *case[:3], dtype, *case[3:])


@pytest.mark.parametrize("case", cases.XDT_BWD_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_xdt_tm_bwd(lib, case, dtype):
    KC.check_xdt_bwd(lib, "cuda", case[0], case[1], dtype, case[2])


def test_xdt_tm_bwd_headline_shape(lib):
    """aum_xdt_tm_bwd at the bench's own launch (64 x 513 tokens, d_inner 1536): dx_dbl and du of every token against fp32 products of
    the same operands (the three library calls it replaces: SSI:570-574, 587, 590), bitwise repeatable"""
    torch.manual_seed(8)
    M, E = 64 * 513, 1536
    ddelta = torch.randn(M, E, device="cuda").bfloat16()
    du0 = torch.randn(M, E, device="cuda").bfloat16()
    dbc = torch.randn(M, 32, device="cuda")
    wdt_t = (torch.randn(48, E, device="cuda") / E ** 0.5).bfloat16()
    wx_t = (torch.randn(E, 80, device="cuda") / 80 ** 0.5).bfloat16()
    du = du0.clone()
    dx = aum_hip.xdt_tm_bwd(ddelta, dbc, wdt_t, wx_t, du, lib=lib)
    rr = ddelta.float() @ wdt_t.float().t()
    assert (dx[:, :48].float() - rr).abs().max().item() <= 1.01 * 2.0 ** -8 * rr.abs().max().item()
    assert torch.equal(dx[:, 48:], dbc.bfloat16())
    ru = du0.float() + dx.float() @ wx_t.float().t()
    assert (du.float() - ru).abs().max().item() <= 1.01 * 2.0 ** -8 * ru.abs().max().item()
    du2 = du0.clone()
    dx2 = aum_hip.xdt_tm_bwd(ddelta, dbc, wdt_t, wx_t, du2, lib=lib)
    assert torch.equal(dx, dx2) and torch.equal(du, du2)


def test_xdt_tm_headline_shape(lib):
    """the bench's own launch (64 x 513 tokens, d_inner 1536, dt_rank 48): x_dbl and delta of every token against the library's fp32
    products of the same operands, bitwise repeatable"""
    torch.manual_seed(4)
    u = torch.randn(64 * 513, 1536, device="cuda").bfloat16()
    wx = (torch.randn(80, 1536, device="cuda") / 1536 ** 0.5).bfloat16()
    wdt = (torch.randn(1536, 48, device="cuda") / 48 ** 0.5).bfloat16()
    x_dbl, delta = aum_hip.xdt_tm_fwd(u, wx, wdt, lib=lib)
    rx = u.float() @ wx.float().t()
    assert (x_dbl.float() - rx).abs().max().item() <= 1.01 * 2.0 ** -8 * rx.abs().max().item()
    rd = x_dbl[:, :48].float() @ wdt.float().t()
    assert (delta.float() - rd).abs().max().item() <= 1.01 * 2.0 ** -8 * rd.abs().max().item()
    x2, d2 = aum_hip.xdt_tm_fwd(u, wx, wdt, lib=lib)
    assert torch.equal(x_dbl, x2) and torch.equal(delta, d2)
