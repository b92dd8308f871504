#!/usr/bin/env python3
"""Launches of the time-serial scan kernels (and the row kernels they replace) at the AuM-Base shape for a kernel trace:
  rocprofv3 --kernel-trace --stats -d gpurun_out/tm_time -o t -- python tools/tm_time.py [lib variant]
Per-kernel durations come from the trace (python's launch overhead exceeds the kernels' run time)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402

lib = aum_hip.get()
if len(sys.argv) > 1 and sys.argv[1] != "default":
    lib = aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", f"libaum_hip_{sys.argv[1]}.so"))
torch.manual_seed(0)
Bsz, E, L, N, dt, dev = 64, 1536, 513, 16, torch.bfloat16, "cuda"
xz = torch.randn(Bsz, L, 2 * E, device=dev).to(dt)
u = torch.randn(Bsz, L, E, device=dev).to(dt)
z = xz[:, :, E:]
dl = (0.5 * torch.randn(Bsz, L, E, device=dev)).to(dt)
xdbl = torch.randn(Bsz, L, 48 + 2 * N, device=dev).to(dt)
Bm, Cm = xdbl[:, :, 48:48 + N], xdbl[:, :, 48 + N:]
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
A_b, D, bias = A * 1.05, torch.ones(E, device=dev), torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
dsp = torch.nn.functional.softplus(dl.float() + bias).to(dt)
dout = torch.randn(Bsz, L, E, device=dev).to(dt)
cw, cb = torch.randn(E, 4, device=dev), torch.randn(E, device=dev)
ck2 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
for rep in range(6):
    # delta ready (the producer applied the softplus), training form
    _, pre = aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, want_out_pre=True, ckpt=ck2, lib=lib)
    aum_hip.scan_tm_bwd(u, dsp, A, Bm, Cm, D, z, None, dout, pre, ck2, False, A_b=A_b, lib=lib)
    # raw delta + bias + softplus inside
    _, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck2, lib=lib)
    aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck2, True, A_b=A_b, lib=lib)
    # the conv on the first half of the in_proj output rows
    aum_hip.conv1d_tm_fwd(xz[:, :, :E], cw, cb, True, lib=lib)
    aum_hip.conv1d_tm_bwd(xz[:, :, :E], cw, cb, dout, True, dx_out=torch.empty_like(xz)[:, :, :E], lib=lib)
    # inference form
    aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, lib=lib)
torch.cuda.synchronize()
