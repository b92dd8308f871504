"""Kernel SOURCES vs the oracle, on the host: the lane-array build (tests/emu) executes the same kernel
bodies that hipcc compiles for gfx950, one 64-lane wavefront at a time, so index arithmetic, tails, chunk
carries, direction handling and reductions are checked here without a GPU.  (The GPU parity tests proper
are in test_gpu_kernels.py and run the real libaum_hip.so.)"""
import os
import sys

import pytest
import torch

import aum_hip
import cases
import kernel_checks as KC

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu():
    import build_emu
    return aum_hip.Lib(build_emu.build(), host=True)


def test_wave_scan_primitive(emu):
    KC.check_wave_scan(emu, "cpu")


@pytest.mark.parametrize("case", cases.SCAN_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("path", ["workgroup", "generic"])
def test_scan(emu, case, mode, path):
    """path=workgroup: the 8-wave LDS-tiled production kernels (dstate<=16); generic: the single-wave kernels."""
    if mode == "bidir" and case[3] > emu.max_single_pass_len:
        pytest.skip("direction fusion is single-pass only; host composes two reverse-flag calls")
    KC.check_scan(emu, "cpu", case, torch.float32, reverse=(mode == "rev"), bidir=(mode == "bidir"),
                  generic=(path == "generic"))


@pytest.mark.parametrize("case", cases.SCAN_WIDE_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
def test_scan_one_row_backward_vs_row_pair(emu, case, mode):
    """L = 513: the backward is scan_half_kernels.h (one row per wave, 12/16-wave workgroups of 96/64 rows); both it and
    the row-pair kernel it replaces must match the oracle on shapes with several, partly idle row groups"""
    for rowpair in (False, True):
        KC.check_scan(emu, "cpu", case, torch.float32, reverse=(mode == "rev"), bidir=(mode == "bidir"), rowpair=rowpair)
    KC.check_scan(emu, "cpu", case, torch.bfloat16, reverse=(mode == "rev"), bidir=(mode == "bidir"))


@pytest.mark.parametrize("case", [c for c in cases.SCAN_CASES if c[0] == "l513"] + cases.SCAN_ROW_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
def test_scan_row_kernels_lane_checkpoint(emu, case, mode):
    """L = 513: scan_row_kernels.h -- the forward fills the lane-entry checkpoint, the backward reads it instead of recomputing the
    forward scan; the tail step of all (direction, state) pairs is one lane-parallel block per row"""
    kw = dict(reverse=(mode == "rev"), bidir=(mode == "bidir"), lane_ckpt=True)
    KC.check_scan(emu, "cpu", case, torch.float32, **kw)
    KC.check_scan(emu, "cpu", case, torch.bfloat16, **kw)
    KC.check_scan(emu, "cpu", case, torch.float32, strided=True, **kw)
    KC.check_scan(emu, "cpu", case, torch.float16, tol=2e-3, **kw)


def test_scan_lane_checkpoint_contract(emu):
    """x_lane exists only for rows the row kernels take; handing one to any other call is refused, not ignored"""
    assert aum_hip.scan_lane_ckpt(torch.zeros(2, 4, 65), 16, False, lib=emu) is None
    assert aum_hip.scan_lane_ckpt(torch.zeros(2, 4, 1025), 16, False, lib=emu) is None
    ck = aum_hip.scan_lane_ckpt(torch.zeros(2, 4, 513), 16, True, lib=emu)
    assert ck.shape == (2, 4, 2, 16, 64) and emu.c.aum_selective_scan_lane_ckpt_bytes(2, 4, 513, 16, 1) == ck.numel() * 4
    z = lambda *s: torch.zeros(*s)
    with pytest.raises(RuntimeError):       # the row-pair kernels have no use for it
        aum_hip.scan_fwd(z(2, 4, 513), z(2, 4, 513), -torch.ones(4, 16), z(2, 16, 513), z(2, 16, 513),
                         x_lane=torch.empty(2, 4, 1, 16, 64), rowpair=True, lib=emu)


@pytest.mark.parametrize("case", cases.SCAN_LONG_CASES + [c for c in cases.SCAN_CASES if c[0] == "l2049"], ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev"])
def test_scan_chunked_one_row_backward(emu, case, mode):
    """L = 512 m (+1) >= 1024: the backward is scanh_bwd_chunked (pre-pass for the chunk-entry states, adjoint carried from
    chunk to chunk); it and the row-pair kernel it replaces (AUM_SCAN_ROWPAIR) must both match the oracle"""
    for rowpair in (False, True):
        KC.check_scan(emu, "cpu", case, torch.float32, reverse=(mode == "rev"), rowpair=rowpair)
    KC.check_scan(emu, "cpu", case, torch.bfloat16, reverse=(mode == "rev"))
    KC.check_scan(emu, "cpu", case, torch.float32, reverse=(mode == "rev"), strided=True)
    KC.check_scan(emu, "cpu", case, torch.float16, reverse=(mode == "rev"), tol=2e-3)
    if case[3] % 512 == 1:      # rows with a checkpoint: the backward reads the forward's chunk-entry states instead of its pre-pass
        KC.check_scan(emu, "cpu", case, torch.float32, reverse=(mode == "rev"), ckpt=True)
        KC.check_scan(emu, "cpu", case, torch.bfloat16, reverse=(mode == "rev"), ckpt=True, strided=True)


@pytest.mark.parametrize("case", [c for c in cases.SCAN_LONG_CASES if c[3] % 512 == 1] + [c for c in cases.SCAN_CASES if c[0] == "l2049"],
                         ids=lambda c: c[0])
def test_scan_second_direction_accumulates(emu, case):
    KC.check_scan_accumulate(emu, "cpu", case, torch.float32)
    KC.check_scan_accumulate(emu, "cpu", case, torch.bfloat16)


def test_scan_accumulate_contract(emu):
    """AUM_SCAN_ACCUMULATE is refused (not ignored) where the chunked kernels do not run"""
    case = [c for c in cases.SCAN_CASES if c[0] == "l65"][0]
    d = cases.scan_inputs(*case)
    t = lambda a: torch.tensor(a)
    u, delta, z = t(d["u"]), t(d["delta"]), t(d["z"])
    Bm, Cm = t(d["B"]).unsqueeze(1), t(d["C"]).unsqueeze(1)
    assert not aum_hip.scan_accumulates(u, 16, lib=emu)
    of, _, _ = aum_hip.scan_fwd(u, delta, t(d["A"]), Bm, Cm, t(d["D"]), z, t(d["delta_bias"]), True, lib=emu)
    with pytest.raises(RuntimeError):
        aum_hip.scan_fwd(u, delta, t(d["A"]), Bm, Cm, t(d["D"]), z, t(d["delta_bias"]), True, True, accumulate_into=of, lib=emu)


def test_scan_checkpoint_contract(emu):
    """x_ck exists only for rows the chunked kernels take; handing one to any other call is refused, not ignored"""
    u = torch.zeros(2, 4, 513)
    assert aum_hip.scan_ckpt(u, 16, lib=emu) is None and aum_hip.scan_ckpt(torch.zeros(2, 4, 1024), 16, lib=emu) is None
    ck = aum_hip.scan_ckpt(torch.zeros(1, 4, 2049), 16, lib=emu)
    assert ck.shape == (1, 4, 4, 16) and emu.c.aum_selective_scan_ckpt_bytes(1, 4, 2049, 16) == ck.numel() * 4
    case = [c for c in cases.SCAN_CASES if c[0] == "l2049"][0]
    with pytest.raises(RuntimeError):
        KC.check_scan(emu, "cpu", case, torch.float32, ckpt=True, rowpair=True)


def test_scan_chunked_one_row_backward_row_groupings(emu, monkeypatch):
    """the chunked kernel sizes its workgroups (16..64 rows) to the grid; small test shapes always get 16 rows, so the
    64-row grouping (four rows per wave, a ragged second group) is forced through the debug bit"""
    case = [c for c in cases.SCAN_LONG_CASES if c[0] == "l1024_d70_n8"][0]
    monkeypatch.setattr(aum_hip.debug, "ablate", 1 << 5)
    for reverse in (False, True):
        KC.check_scan(emu, "cpu", case, torch.float32, reverse=reverse)


@pytest.mark.parametrize("mode", ["fwd", "bidir"])
def test_scan_bf16_and_strided(emu, mode):
    case = [c for c in cases.SCAN_CASES if c[0] == "l65"][0]
    KC.check_scan(emu, "cpu", case, torch.bfloat16, bidir=(mode == "bidir"))
    KC.check_scan(emu, "cpu", case, torch.float16, bidir=(mode == "bidir"), tol=2e-3)
    KC.check_scan(emu, "cpu", case, torch.float32, bidir=(mode == "bidir"), strided=True)


@pytest.mark.parametrize("case", cases.CONV_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("reverse", [False, True])
def test_conv(emu, case, reverse):
    KC.check_conv(emu, "cpu", case, torch.float32, reverse=reverse)
    KC.check_conv(emu, "cpu", case, torch.float32, reverse=reverse, silu=False)


def test_conv_bf16(emu):
    KC.check_conv(emu, "cpu", cases.CONV_CASES[2], torch.bfloat16)


@pytest.mark.parametrize("case", [c for c in cases.CONV_CASES if 512 <= c[3] <= 520], ids=lambda c: c[0])
@pytest.mark.parametrize("reverse", [False, True])
def test_conv_rows_kernels_16bit(emu, case, reverse):
    """len = 512 + tail, 16-bit: conv_rows_kernels.h (8 rows of a channel per wave, wave-uniform tail steps)"""
    KC.check_conv(emu, "cpu", case, torch.bfloat16, reverse=reverse)
    KC.check_conv(emu, "cpu", case, torch.float16, reverse=reverse, silu=False)


@pytest.mark.parametrize("case", cases.NORM_CASES, ids=lambda c: c[0])
def test_norm(emu, case):
    KC.check_norm(emu, "cpu", case, torch.float32)
    KC.check_norm(emu, "cpu", case, torch.bfloat16, torch.float32)


def test_generic_conv_and_norm_kernels(emu):
    """the any-width / any-cols kernels stay covered now that width 4 / cols <= 2048 take the vectorised kernels"""
    for case in cases.CONV_CASES:
        KC.check_conv(emu, "cpu", case, torch.float32, reverse=False, generic=True)
        KC.check_conv(emu, "cpu", case, torch.float32, reverse=True, generic=True)
    for case in cases.NORM_CASES:
        KC.check_norm(emu, "cpu", case, torch.float32, generic=True)


@pytest.mark.parametrize("case", cases.PROJ_CASES, ids=lambda c: c[0])
def test_proj(emu, case):
    KC.check_proj(emu, "cpu", case, torch.bfloat16)
    KC.check_proj(emu, "cpu", case, torch.float16)


def test_proj_limits(emu):
    x = torch.zeros(64, 8, dtype=torch.float32)
    assert not aum_hip.proj_supported(64, 48, 16, 8, torch.float32)
    assert not aum_hip.proj_supported(96, 48, 16, 8, torch.bfloat16)
    assert not aum_hip.proj_supported(64, 48, 32, 8, torch.bfloat16)
    assert aum_hip.proj_supported(1536, 48, 16, 64 * 513, torch.bfloat16)
    with pytest.raises(RuntimeError, match="UNSUPPORTED|unsupported"):
        aum_hip.proj_fwd(x, torch.zeros(80, 64), torch.zeros(64, 48), 16, lib=emu)


def test_sum_rows_emu(emu):
    """aum_sum_rows on the lane-array build: fp32 / bf16 partials, shapes of the three callers (norm partials, projection splits, split-K)"""
    torch.manual_seed(0)
    for shape, dt in (((1024, 16), torch.float32), ((37, 768), torch.float32), ((42, 48, 64), torch.float32), ((4, 24, 16), torch.bfloat16), ((3, 7), torch.float32)):
        t = torch.randn(shape).to(dt)
        got = aum_hip.sum_rows(t, lib=emu)
        assert got.dtype == torch.float32 and got.shape == t.shape[1:]
        assert torch.allclose(got, t.float().sum(0), rtol=1e-5, atol=1e-5)


def test_sum_rows_multi_emu(emu):
    """aum_sum_rows_multi (ABI 11) on the lane-array build: the two pairs a layer's backward hands over (conv weight + bias partials; the
    skinny dt_proj / x_proj partial sets, the second stored transposed), one to four jobs, and the shapes that go job by job instead"""
    torch.manual_seed(0)
    dw, db = torch.randn(9, 24, 4), torch.randn(9, 24)
    gw, gb = aum_hip.sum_rows_multi([dw, db], lib=emu)
    assert gw.shape == (24, 4) and gb.shape == (24,)
    assert torch.allclose(gw, dw.sum(0), rtol=1e-5, atol=1e-5) and torch.allclose(gb, db.sum(0), rtol=1e-5, atol=1e-5)
    p1, p2 = torch.randn(5, 32, 48), torch.randn(5, 32, 80)
    g1, g2 = aum_hip.sum_rows_multi([p1, p2], [0, 80], lib=emu)
    assert g1.shape == (32, 48) and g2.shape == (80, 32) and g2.is_contiguous()
    assert torch.allclose(g1, p1.sum(0), rtol=1e-5, atol=1e-5) and torch.allclose(g2, p2.sum(0).t(), rtol=1e-5, atol=1e-5)
    eight = [torch.randn(3 + q, 8 * (q + 1)) for q in range(aum_hip.SUM_MAX_JOBS)]
    for got, t in zip(aum_hip.sum_rows_multi(eight, lib=emu), eight):
        assert torch.allclose(got, t.sum(0), rtol=1e-5, atol=1e-5)
    one = aum_hip.sum_rows_multi([torch.ones(1, 16)], lib=emu)
    assert torch.equal(one[0], torch.ones(16))
    odd = [torch.randn(3, 7), torch.randn(3, 2, 5)]              # inner % 8 != 0: one by one (sum_rows' own fallback), transposition included
    g = aum_hip.sum_rows_multi(odd, [0, 5], lib=emu)
    assert torch.allclose(g[0], odd[0].sum(0), rtol=1e-5, atol=1e-5) and torch.allclose(g[1], odd[1].sum(0).t(), rtol=1e-5, atol=1e-5)
    import ctypes
    jobs = (aum_hip.SumJob * 1)()
    jobs[0].src, jobs[0].dst, jobs[0].outer, jobs[0].inner, jobs[0].tr_cols = dw.data_ptr(), gw.data_ptr(), 9, 96, 5          # 96 % 5 != 0
    assert emu.c.aum_sum_rows_multi(ctypes.cast(jobs, ctypes.c_void_p), 1, None) != 0
    assert emu.c.aum_sum_rows_multi(ctypes.cast(jobs, ctypes.c_void_p), aum_hip.SUM_MAX_JOBS + 1, None) != 0


def test_cast_bank_emu(emu):
    KC.check_cast_bank(emu, "cpu")


# ---- time-serial token-major kernels (scan_tm_kernels.h, conv_tm_kernels.h) ---------------------------------------------------
def test_wave_sum_butterflies(emu):
    KC.check_wave_sum32(emu, "cpu")


@pytest.mark.parametrize("case", cases.SCAN_TM_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
def test_scan_tm(emu, case, mode):
    """aum_scan_tm_fwd / _bwd (lanes = channels, time serial, token-major operands, state checkpoints every 8 steps, the Fo-Bi pair
    meeting in the middle, three pairs per workgroup in the backward) against the fp64 oracle: fp32 at 1e-3, 16-bit I/O at 1e-2;
    ragged blocks, L < 8, z / D absent, two batch entries, the [x | z] row layout of the block"""
    for dt, xz in ((torch.float32, False), (torch.bfloat16, True)):
        KC.check_scan_tm(emu, "cpu", case, dt, reverse=(mode == "rev"), bidir=(mode == "bidir"), xz_layout=xz, backward=True)


@pytest.mark.parametrize("case", [c for c in cases.SCAN_TM_CASES if c[3] >= 9], ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["fwd", "rev", "bidir"])
@pytest.mark.parametrize("segments", [2, 3, 5])
def test_scan_tm_segments(emu, case, mode, segments):
    """aum_scan_tm_seg_fwd / _bwd: the rows cut into time segments that run as waves of their own (carry pass + main pass; the adjoint
    the same way in the backward), against the same fp64 oracle at the same tolerances as the uncut launches: ranges of one block,
    ragged last ranges, empty ranges (more segments than blocks), z / D absent, both directions and the Fo-Bi pair"""
    for dt, xz in ((torch.float32, False), (torch.bfloat16, True)):
        KC.check_scan_tm(emu, "cpu", case, dt, reverse=(mode == "rev"), bidir=(mode == "bidir"), xz_layout=xz, backward=True, segments=segments)


@pytest.mark.parametrize("case", cases.CONV_TM_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("reverse", [False, True])
def test_conv_tm(emu, case, reverse):
    """aum_conv1d_tm_fwd / _bwd against the conv oracle: chunk seams (64 steps), width 3, no bias, more channels than one wave
    covers, x / dx as the first half of wider rows, all three dtypes"""
    for dt, silu, xz in ((torch.float32, True, False), (torch.bfloat16, True, True), (torch.float16, False, False)):
        KC.check_conv_tm(emu, "cpu", case, dt, reverse, silu, xz)


def test_tm_limits(emu):
    """shapes outside the token-major kernels' limits are refused (callers fall back to the channel-major kernels)"""
    assert aum_hip.scan_tm_supported(1536, 16) and not aum_hip.scan_tm_supported(1536, 8) and not aum_hip.scan_tm_supported(96, 16)
    x = torch.zeros(1, 4, 12)
    assert not aum_hip.conv1d_tm_supported(x.bfloat16(), 4) and aum_hip.conv1d_tm_supported(torch.zeros(1, 4, 16).bfloat16(), 4)
    assert not aum_hip.conv1d_tm_supported(torch.zeros(1, 4, 16), 5)
    with pytest.raises(RuntimeError, match="UNSUPPORTED|unsupported"):
        aum_hip.conv1d_tm_fwd(x.bfloat16(), torch.zeros(12, 4), None, lib=emu)


@pytest.mark.parametrize("case", [c for c in cases.GEMM_CASES if c[1] * c[2] * c[3] < 2e8], ids=lambda c: c[0])
def test_gemm_tn_contract(emu, case):
    """the host build's aum_gemm_tn (a plain loop behind the same argument rules, tests/emu/aum_emu.cpp) keeps the contract the GPU
    tests hold the MFMA kernel to; the kernel's own tile layout is checked in test_gemm_layout.py"""
    KC.check_gemm(emu, "cpu", case, torch.bfloat16)


def test_gemm_tn_argument_rules(emu):
    KC.check_gemm_args(emu, "cpu")


@pytest.mark.parametrize("case", [c for c in cases.DTPROJ_CASES if c[0] * c[1] < 1e5], ids=lambda c: "x".join(map(str, c)))
def test_dtproj_tm_contract(emu, case):
    """the host build's aum_dtproj_tm_fwd (a plain loop behind the shared argument rules) keeps the contract the GPU tests hold the MFMA
    kernel to: only the first dt_rank columns of the x_dbl rows enter the product, fp32 accumulation, one rounding"""
    KC.check_dtproj(emu, "cpu", *case, torch.bfloat16)


def test_xdt_tm_contract(emu):
    """the host build's aum_xdt_tm_fwd (plain loops behind the shared argument rules): x_dbl rounded once, delta from the rounded x_dbl"""
    KC.check_xdt(emu, "cpu", 33, 256, 24, torch.bfloat16)
    KC.check_xdt(emu, "cpu", 33, 256, 24, torch.bfloat16, 56)


def test_xdt_tm_bwd_contract(emu):
    """the host build's aum_xdt_tm_bwd (plain loops behind the shared argument rules): dx_dbl rounded once, du from the rounded dx_dbl, in place"""
    KC.check_xdt_bwd(emu, "cpu", 33, 256, torch.bfloat16, 8)


def test_scan_tm_grid_small(emu):
    """the whole-launch oracle check of test_gpu_kernels.py::test_scan_tm_headline_grid_b64 (sampled rows, whole-entry dB | dC, batch-
    summed parameter gradients, batch splits) on a launch small enough for the lane-array build: 5 entries x 3 channel groups = 15
    units -> forward workgroups of 2 and backward workgroups of 3 pairs, both with a ragged last workgroup"""
    rows = {0: [0, 63, 64, 191], 1: [5, 100], 2: [128, 127], 4: [0, 191, 77]}
    KC.check_scan_tm_grid(emu, "cpu", 5, 41, 192, rows, (0, 4), [0, 63, 64, 191], 1)


def test_scan_tm_grid_small_segments(emu):
    """the long-form whole-launch check (test_gpu_kernels.py::test_scan_tm_longform_grid_b8) at a size the lane-array build finishes:
    forward cut into 4 ranges, backward into 3, rows of 75 steps (ranges of 24 / 32 steps, the last ragged)"""
    rows = {0: [0, 63, 64, 191], 1: [5, 100], 2: [128, 127]}
    KC.check_scan_tm_grid(emu, "cpu", 3, 75, 192, rows, (0, 2), [0, 63, 64, 191], 1, segments=(4, 3))


@pytest.mark.parametrize("case", cases.GEMM_WGRAD_CASES[:4] + cases.GEMM_WGRAD_CASES[6:8], ids=lambda c: "x".join(map(str, c)))
def test_gemm_wgrad_contract(emu, case):
    """the host twin of aum_gemm_wgrad (tests/emu/aum_emu.cpp: shared argument rules and split boundaries): what the host-side dispatch
    tests run against; the device kernel's own parity is test_gpu_kernels.py::test_gemm_wgrad*"""
    KC.check_gemm_wgrad(emu, "cpu", *case[:4], torch.bfloat16, *case[4:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_kernels_contract(emu, dtype):
    """the host twins of the per-token kernels (tests/emu/aum_emu.cpp) against the reference's step arithmetic in fp64"""
    KC.check_decode_kernels(emu, "cpu", dtype)
