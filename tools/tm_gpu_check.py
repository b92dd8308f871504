#!/usr/bin/env python3
"""GPU parity of the time-serial token-major scan against the fp64 oracle (the shared checks of tests/kernel_checks.py), plus a
bitwise-repeatability screen of the direction pair's hand-over through `out`.  Prints one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-mamba-aum_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch  # noqa: E402

import aum_hip  # noqa: E402
import cases  # noqa: E402
import kernel_checks as KC  # noqa: E402

lib = aum_hip.get()
bwd = "--bwd" in sys.argv
bad = 0
KC.check_wave_sum32(lib, "cuda")
print("ok  wave_sum32 / wave_sum16")
for case in cases.CONV_TM_CASES:
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for rev in (False, True):
            for silu, xz in ((True, False), (True, True), (False, False)):
                try:
                    e = KC.check_conv_tm(lib, "cuda", case, dt, rev, silu, xz)
                    print("ok ", case[0], str(dt)[6:], "rev" if rev else "", "silu" if silu else "", "xz" if xz else "",
                          " ".join(f"{k}={v:.1e}" for k, v in e.items()), flush=True)
                except AssertionError as ex:
                    bad += 1
                    print("BAD", case[0], str(dt)[6:], rev, silu, xz, ex, flush=True)
for case in cases.SCAN_TM_CASES:
    for mode in ("fwd", "rev", "bidir"):
        for dt, xz in ((torch.float32, False), (torch.bfloat16, True), (torch.bfloat16, False), (torch.float16, False)):
            try:
                e = KC.check_scan_tm(lib, "cuda", case, dt, reverse=(mode == "rev"), bidir=(mode == "bidir"), xz_layout=xz, backward=bwd,
                                     tol=2e-3 if dt == torch.float16 else None)
                print("ok ", case[0], mode, str(dt)[6:], "xz" if xz else "", " ".join(f"{k}={v:.1e}" for k, v in e.items()), flush=True)
            except AssertionError as ex:
                bad += 1
                print("BAD", case[0], mode, str(dt)[6:], ex, flush=True)
# repeatability at the bench shape: the two waves of a pair exchange partial sums through `out` around one barrier
torch.manual_seed(0)
Bsz, L, E, N = 64, 513, 1536, 16
u = torch.randn(Bsz, L, E, device="cuda").bfloat16()
z = torch.randn(Bsz, L, E, device="cuda").bfloat16()
dl = (0.5 * torch.randn(Bsz, L, E, device="cuda")).bfloat16()
bc = torch.randn(Bsz, L, 2 * N, device="cuda").bfloat16()
A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(E, 1)
D = torch.ones(E, device="cuda")
bias = torch.full((E,), -4.0, device="cuda")
outs = []
for i in range(4):
    o, pre = aum_hip.scan_tm_fwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, A_b=A * 1.05, want_out_pre=True)
    outs.append((o.clone(), pre.clone()))
same = all(torch.equal(outs[0][0], o) and torch.equal(outs[0][1], p) for o, p in outs[1:])
print("bench-shape bidirectional forward bitwise repeatable over 4 launches:", same, "finite:", bool(torch.isfinite(outs[0][0].float()).all()))
# the same launch against two one-direction launches (fp32 sums of the two, rounded once vs the partial's extra rounding)
of, pf = aum_hip.scan_tm_fwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, want_out_pre=True)
ob, pb = aum_hip.scan_tm_fwd(u, dl, A * 1.05, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, reverse=True, want_out_pre=True)
ref = pf.float() + pb.float()
err = (outs[0][1].float() - ref).abs().max().item() / ref.abs().max().item()
print(f"bench-shape pair vs two one-direction launches: out_pre max rel diff {err:.2e}")
bad += (not same) + (err > 1e-2)
if bwd:
    # backward at the bench shape: bitwise repeatable (no atomics; every hand-over through memory is ordered by a counted wait or a
    # barrier -- a wait that names too few operations shows up here as a run-to-run difference), and the pair against two
    # one-direction launches
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, "cuda")
    _, pre = aum_hip.scan_tm_fwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, A_b=A * 1.05, want_out_pre=True, ckpt=ck)
    dout = torch.randn(Bsz, L, E, device="cuda").bfloat16()
    runs = []
    for i in range(4):
        r = aum_hip.scan_tm_bwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, dout, pre, ck, True, A_b=A * 1.05)
        runs.append({k: v.clone() for k, v in r.items() if v is not None and not k.startswith("_")})
    same_b = all(torch.equal(runs[0][k], r[k]) for r in runs[1:] for k in runs[0])
    fin = all(bool(torch.isfinite(v.float()).all()) for v in runs[0].values())
    print("bench-shape bidirectional backward bitwise repeatable over 4 launches:", same_b, "finite:", fin)
    ckf = aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, "cuda")
    ckb = aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, "cuda")
    aum_hip.scan_tm_fwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, want_out_pre=True, ckpt=ckf)
    aum_hip.scan_tm_fwd(u, dl, A * 1.05, bc[:, :, :N], bc[:, :, N:], D, z, bias, True, reverse=True, want_out_pre=True, ckpt=ckb)
    # one-direction backwards with the PAIR's pre-gate output (dz depends on the total)
    rf = aum_hip.scan_tm_bwd(u, dl, A, bc[:, :, :N], bc[:, :, N:], D, z, bias, dout, pre, ckf, True)
    rb = aum_hip.scan_tm_bwd(u, dl, A * 1.05, bc[:, :, :N], bc[:, :, N:], D, z, bias, dout, pre, ckb, True, reverse=True)
    worst = 0.0
    for k in ("du", "ddelta", "dBC", "dD", "ddelta_bias"):
        ref = rf[k].float() + rb[k].float()
        worst = max(worst, (runs[0][k].float() - ref).abs().max().item() / ref.abs().max().item())
    for k, ref in (("dA", rf["dA"]), ("dA_b", rb["dA"]), ("dz", rf["dz"].float())):
        worst = max(worst, (runs[0][k].float() - ref.float()).abs().max().item() / ref.float().abs().max().item())
    print(f"bench-shape backward pair vs two one-direction launches: max rel diff {worst:.2e}")
    bad += (not same_b) + (not fin) + (worst > 2e-2)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
