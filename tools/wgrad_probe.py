#!/usr/bin/env python3
"""aum_gemm_wgrad (+ aum_sum_rows of its partial tiles) against the library path it replaces (split-K strided batched GEMMs + aum_sum_rows, TunableOp
solutions) on the bench's two weight-gradient GEMMs; interleaved rounds, HIP events, operands rotated through 4 buffers (HBM-cold)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import torch  # noqa: E402
import aum_hip  # noqa: E402
import mamba_ssm.ops.selective_scan_interface as ssi  # noqa: E402

t = 64 * 513
for name, n, k, hint in (("in_proj  dW[3072][768]", 3072, 768, ssi._WGRAD_SPLITS[0]), ("out_proj dW[768][1536]", 768, 1536, ssi._WGRAD_SPLITS[1])):
    ys = [(torch.randn(t, n, device="cuda") * 0.1).bfloat16() for _ in range(4)]
    xs = [torch.randn(t, k, device="cuda").bfloat16() for _ in range(4)]
    fns = {"hip": lambda y, x: aum_hip.gemm_wgrad(y, x), "hip kernel only": lambda y, x: aum_hip.gemm_wgrad(y, x, partials=True),
           "lib": lambda y, x: ssi.split_k_wgrad(y.t(), x, ssi._pick_splits(t, hint), torch.float32)}
    for f in fns.values():
        for i in range(3):
            f(ys[i], xs[i])
    res = {kk: [] for kk in fns}
    for rnd in range(6):
        for kk, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8):
                f(ys[i % 4], xs[i % 4])
            e1.record()
            torch.cuda.synchronize()
            res[kk].append(e0.elapsed_time(e1) / 8 * 1e3)
    fl = 2.0 * t * n * k
    print(name, "  ".join(f"{kk}: {sorted(v)[len(v) // 2]:.1f} us ({fl / sorted(v)[len(v) // 2] / 1e6:.0f} TFLOP/s)" for kk, v in res.items()), flush=True)
    d = (fns["hip"](ys[0], xs[0]) - fns["lib"](ys[0], xs[0])).abs().max().item() / fns["lib"](ys[0], xs[0]).abs().max().item()
    print("   max |hip - lib| / max|lib| =", d)
