#!/usr/bin/env python3
"""What the skinny x_proj / dt_proj gradient pieces of the token-major block's backward cost one by one at the bench shape
(M = 64 x 513 tokens, E = 1536, dt rank 48, N = 16; bf16): HIP-event medians, the HBM-bound time of each piece beside it.
  python tools/skinny_probe.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import aum_hip  # noqa: E402
import mamba_ssm.ops.selective_scan_interface as ssi  # noqa: E402

dev, bf = "cuda", torch.bfloat16
M, E, R, N = 64 * 513, 1536, 48, 16
X = R + 2 * N
torch.manual_seed(0)
ddelta = torch.randn(M, E, device=dev).to(bf)
du = torch.randn(M, E, device=dev).to(bf)
conv_out = torch.randn(M, E, device=dev).to(bf)
x_dbl = torch.randn(M, X, device=dev).to(bf)
dBC = torch.randn(M, 2 * N, device=dev)
w_dt = torch.randn(E, R, device=dev).to(bf)
w_x = torch.randn(X, E, device=dev).to(bf)
dx_dbl = torch.empty_like(x_dbl)
# 16 independent buffers > L2 + MALL so that every repetition reads HBM
flush = torch.empty(600 * 2 ** 20, dtype=torch.uint8, device=dev)


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


splits = ssi._pick_splits(M, ssi._WGRAD_SPLITS[1])
rows = [
    ("a  dx_dbl[:, R:] = dBC (cast)", lambda: dx_dbl[:, R:].copy_(dBC), M * 2 * N * 6),
    ("b  dx_dbl[:, :R] = ddelta @ W_dt", lambda: dx_dbl[:, :R].copy_(torch.matmul(ddelta, w_dt)), M * E * 2),
    ("c  dW_dt = ddelta^T @ x_dbl[:, :R]", lambda: ssi.split_k_wgrad(ddelta.t(), x_dbl[:, :R], splits, torch.float32), M * E * 2),
    ("d  dW_x = dx_dbl^T @ conv_out", lambda: ssi.split_k_wgrad(dx_dbl.t(), conv_out, splits, torch.float32), M * E * 2),
    ("e  du += dx_dbl @ W_x", lambda: du.addmm_(dx_dbl, w_x), M * E * 4),
    ("f  forward: x_dbl, delta = xdt(conv_out)", lambda: aum_hip.xdt_tm_fwd(conv_out, w_x, w_dt), M * E * 4),
]
w_dt_t, w_x_t = w_dt.t().contiguous(), w_x.t().contiguous()
rows += [
    ("P  aum_xdt_tm_bwd (a + b + e in one pass)", lambda: aum_hip.xdt_tm_bwd(ddelta, dBC, w_dt_t, w_x_t, du), M * E * 6),
    ("Qc aum_gemm_wgrad k=48 (c) incl. partial sums", lambda: aum_hip.gemm_wgrad(ddelta, x_dbl[:, :R]), M * E * 2),
    ("Qd aum_gemm_wgrad k=80 (d) incl. partial sums", lambda: aum_hip.gemm_wgrad(conv_out, dx_dbl), M * E * 2),
    ("Qc kernel alone (partials kept)", lambda: aum_hip.gemm_wgrad(ddelta, x_dbl[:, :R], partials=True), M * E * 2),
    ("Qd kernel alone (partials kept)", lambda: aum_hip.gemm_wgrad(conv_out, dx_dbl, partials=True), M * E * 2),
]
for sp in (6, 12, 21, 42):
    rows.append((f"Qd kernel alone, {sp} splits", (lambda sp=sp: aum_hip.gemm_wgrad(conv_out, dx_dbl, splits=sp, partials=True)), M * E * 2))
tot = 0.0
for name, fn, nbytes in rows:
    t = timed(fn)
    tot += t if name[0] in "abcde" and name[1] == " " else 0
    print(f"{name:45s} {t:8.1f} us   HBM bound {nbytes / 6.29e6:6.1f} us (6.29 TB/s copy rate)", flush=True)
print(f"backward pieces a-e together: {tot:.1f} us per layer; HBM bound (ddelta, conv_out once; du read + write): {M * E * 8 / 6.29e6:.1f} us")
