#!/usr/bin/env python3
"""Sweep the split-K count of the in/out-projection weight-gradient GEMMs (split_k_wgrad) at AuM-Base shapes with TunableOp tuning
each new batched shape online; writes the tuned solutions next to the timings so they can be merged into aum/tunableop_gfx950.csv."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import torch  # noqa: E402
from mamba_ssm.ops.selective_scan_interface import split_k_wgrad  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


ntok = 64 * 513
dxz2d = torch.randn(3072, ntok, device="cuda").to(torch.bfloat16)
h = torch.randn(ntok, 768, device="cuda").to(torch.bfloat16)
dout2 = torch.randn(ntok, 768, device="cuda").to(torch.bfloat16)
outz = torch.randn(1536, ntok, device="cuda").to(torch.bfloat16)
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
base = timeit(lambda: big.zero_())
for s in (1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 19, 24, 27):
    if ntok % s:
        continue
    ti = timeit(lambda: (big.zero_(), split_k_wgrad(dxz2d, h, s, torch.float32))) - base
    to = timeit(lambda: (big.zero_(), split_k_wgrad(dout2.t(), outz.t(), s, torch.float32))) - base
    print(f"splits {s:2d}: in_proj dW {ti:6.1f} us   out_proj dW {to:6.1f} us   (cold caches, incl. the partial sum)", flush=True)
out = os.path.join(ROOT, "gpurun_out", "tunableop_wgrad_sweep.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
import torch.cuda.tunable  # noqa: E402
torch.cuda.tunable.write_file(out)
print("wrote", out)
