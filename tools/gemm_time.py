#!/usr/bin/env python3
"""GPU: time aum_gemm_tn with given flags on the four projection shapes (no parity check: ablation builds give wrong results).
usage: gemm_time.py <flags> [label]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402

fl = int(sys.argv[1])
M = 64 * 513
torch.manual_seed(0)
res = []
for N, K in ((3072, 768), (768, 1536), (1536, 768), (768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: aum_hip.gemm_tn(x, w, out=o, flags=fl)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        r.append(a.elapsed_time(b) * 100)
    res.append(f"{N}x{K}: {statistics.median(r):.1f} us")
print(sys.argv[2] if len(sys.argv) > 2 else fl, " | ".join(res))
