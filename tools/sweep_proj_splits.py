#!/usr/bin/env python3
"""Sweep the token-split count of the projection weight-gradient kernel (kernel + partial sum) at AuM-Base shapes."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


E, ntok = 1536, 64 * 513
x = torch.randn(E, ntok, device="cuda").to(torch.bfloat16)
y80 = torch.randn(80, ntok, device="cuda").to(torch.bfloat16)
y48 = y80[:48]
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")      # 256 MiB: flush the caches between timed calls
for s in [int(v) for v in (sys.argv[1:] or ["0", "21", "42", "64", "85", "86", "128", "170"])]:
    aum_hip.debug.proj_splits = s
    tx = timeit(lambda: (big.zero_(), aum_hip.proj_bwd_weight(x, y80, True))) - timeit(lambda: big.zero_())
    td = timeit(lambda: (big.zero_(), aum_hip.proj_bwd_weight(x, y48, False))) - timeit(lambda: big.zero_())
    print(f"splits {s or 'default'}: dW_x {tx:.1f} us, dW_dt {td:.1f} us (cold caches, incl. partial sum)", flush=True)
