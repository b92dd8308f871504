#!/usr/bin/env python3
"""Do the four big projection GEMMs pay for the 513th token?  Times each at ntok = 64 * 513 and at 64 * 512 (TunableOp tuning both
online): if the round number is much faster, issuing 32768 tokens + a 64-token remainder as two GEMMs would win."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import torch  # noqa: E402


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


bf = torch.bfloat16
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
base = timeit(lambda: big.zero_())
w_in = torch.randn(3072, 768, device="cuda").to(bf)
w_out = torch.randn(768, 1536, device="cuda").to(bf)
for ntok in [int(v) for v in (sys.argv[1:] or ['32832', '32768', '64'])]:
    h = torch.randn(ntok, 768, device="cuda").to(bf)
    y2d = torch.randn(1536, ntok, device="cuda").to(bf)
    dout2 = torch.randn(ntok, 768, device="cuda").to(bf)
    dxz2d = torch.randn(3072, ntok, device="cuda").to(bf)
    t = {
        "in_proj fwd": timeit(lambda: (big.zero_(), torch.matmul(w_in, h.t()))) - base,
        "out_proj fwd": timeit(lambda: (big.zero_(), torch.matmul(y2d.t(), w_out.t()))) - base,
        "out_proj dgrad": timeit(lambda: (big.zero_(), torch.matmul(w_out.t(), dout2.t()))) - base,
        "in_proj dgrad": timeit(lambda: (big.zero_(), torch.matmul(dxz2d.t(), w_in))) - base,
    }
    print(f"ntok {ntok:6d}: " + "   ".join(f"{k} {v:6.1f} us" for k, v in t.items()), flush=True)
