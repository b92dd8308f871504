#!/usr/bin/env python3
"""GPU: where does a variant of aum_gemm_tn (flags argument) differ from the default kernel?  usage: gemm_dbg.py <flags> [N K [M]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402

fl = int(sys.argv[1])
N, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (768, 1536)
M = int(sys.argv[4]) if len(sys.argv) > 4 else 64 * 513
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
ref = aum_hip.gemm_tn(x, w, flags=aum_hip.GEMM_LOCKSTEP).float()
for rep in range(3):
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    aum_hip.gemm_tn(x, w, out=out, flags=fl)
    torch.cuda.synchronize()
    o = out.float()
    bad = ~(o == ref)
    print(f"rep {rep}: mismatching elements {int(bad.sum())} of {M * N}; NaN left {int(torch.isnan(o).sum())}; max abs diff {float((o - ref).nan_to_num(1e9).abs().max()):.4g}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  rows:", rows[:12].tolist(), "...", rows[-4:].tolist(), "count", len(rows))
        print("  cols:", cols[:24].tolist(), "...", cols[-4:].tolist(), "count", len(cols))
        # pattern inside the first bad 256 x 192 tile
        r0, c0 = int(rows[0]) // 256 * 256, int(cols[0]) // 192 * 192
        t = bad[r0:r0 + 256, c0:c0 + 192]
        print("  first bad tile at", (r0, c0), "bad per 16-row fragment:", t.view(16, 16, 192).any(1).sum(1).tolist() if t.shape == (256, 192) else t.shape)
        print("  bad per 16-col group:", t.view(-1, 12, 16).any(2).any(0).tolist() if t.shape[1] == 192 else None)
        i, j = int(rows[0]), int(cols[0])
        print("  sample", (i, j), float(o[i, j]), "vs", float(ref[i, j]))
