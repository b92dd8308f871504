#!/usr/bin/env python3
"""GPU: aum_gemm_tn and the library GEMM on the four projection shapes with the activation operand (a) re-read from the same buffer every launch
(it stays in the 256 MB Infinity Cache), (b) rotated over buffers larger than that cache together (first touch from HBM), (c) behind a streaming
kernel that has just written it (the in-step situation).  Why: inside the step both run 5-12 % slower than in a GEMM-only loop."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402
from aum import tunable  # noqa: E402

tunable.enable()
VARIANTS = [v for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else []) if v]      # variant libraries (tools/build_gemm_variant.sh)
VLIBS = {v: aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", f"libaum_hip_{v}.so"), host=False) for v in VARIANTS}
M = 64 * 513
dev = "cuda"
torch.manual_seed(0)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, K, N in [("in_proj_fwd", 768, 3072), ("out_proj_fwd", 1536, 768), ("out_proj_dgrad", 768, 1536), ("in_proj_dgrad", 3072, 768)]:
    nb = max(2, int(700e6 // (M * K * 2)) + 1)
    xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    src = torch.randn(M, K, device=dev)
    wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    arms = [("hip", lambda x: aum_hip.gemm_tn(x, wt, out=out))]
    for v in VARIANTS:
        arms.append((v, lambda x, L=VLIBS[v]: aum_hip.gemm_tn(x, wt, out=out, lib=L)))
    arms.append(("lib", lambda x: torch.matmul(x, wt.t(), out=out)))
    for label, mm in arms:
        for _ in range(3):
            mm(xs[0])
        same = statistics.median(timed(lambda i: mm(xs[0]), 10) for _ in range(5))
        rot = statistics.median(timed(lambda i: mm(xs[i % nb]), 10) for _ in range(5))
        # behind a producer: a cast kernel writes the operand, the GEMM reads it; the producer alone is timed and subtracted
        prod = statistics.median(timed(lambda i: xs[i % nb].copy_(src), 10) for _ in range(5))
        both = statistics.median(timed(lambda i: (xs[i % nb].copy_(src), mm(xs[i % nb])), 10) for _ in range(5))
        res[label] = (round(same, 1), round(rot, 1), round(both - prod, 1))
    print(f"{name:15s} n={N:5d} k={K:5d}  " + "  ".join(f"{k}: same {v[0]} rotated {v[1]} behind-producer {v[2]}" for k, v in res.items()), flush=True)
