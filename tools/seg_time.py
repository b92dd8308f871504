#!/usr/bin/env python3
"""Time-segmented token-major scan (aum_scan_tm_seg_fwd / _bwd) at the long-form shape (B = 8, L = 4097, E = 1536, bf16, Fo-Bi):
launch-to-launch times (HIP events on the current stream, median of 7) for every segment count, next to the uncut token-major launch.
  python tools/seg_time.py [B L]        -> one JSON line per row; gpurun_out/seg_time.json"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402

lib = aum_hip.get()
torch.manual_seed(0)
Bsz, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 4097)
E, N, dt, dev = 1536, 16, torch.bfloat16, "cuda"
xz = torch.randn(Bsz, L, 2 * E, device=dev).to(dt)
u = torch.randn(Bsz, L, E, device=dev).to(dt)
z = xz[:, :, E:]
dl = (0.5 * torch.randn(Bsz, L, E, device=dev)).to(dt)
xdbl = torch.randn(Bsz, L, 48 + 2 * N, device=dev).to(dt)
Bm, Cm = xdbl[:, :, 48:48 + N], xdbl[:, :, 48 + N:]
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
A_b, D, bias = A * 1.05, torch.ones(E, device=dev), torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
dout = torch.randn(Bsz, L, E, device=dev).to(dt)
ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev, dtype=dt)


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


rows = []
ref = None
SEGS = [int(v) for v in os.environ.get("AUM_SEGS", "1,2,4,6,8,11,12,16,24,32").split(",")]
for seg in SEGS:
    if L // seg < 64:
        continue
    fw = lambda: aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=lib, segments=seg)
    out, pre = fw()
    bw = lambda: aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=lib, segments=seg)
    g = bw()
    inf = lambda: aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, lib=lib, segments=seg)
    if ref is None and seg != 1:
        o1, p1 = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=lib)
        g1 = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, p1, ck, True, A_b=A_b, lib=lib)
        ref = (o1.float(), g1["du"].float(), g1["dBC"].clone(), g1["dA"].clone())
        fw()
    if ref is None:
        ref = (out.float(), g["du"].float(), g["dBC"].clone(), g["dA"].clone())
    err = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    row = dict(batch=Bsz, len=L, segments=seg, fwd_train_ms=round(timed(fw), 4), fwd_infer_ms=round(timed(inf), 4), bwd_ms=round(timed(bw), 4),
               vs_uncut=dict(out=err(out, ref[0]), du=err(g["du"], ref[1]), dBC=err(g["dBC"], ref[2]), dA=err(g["dA"], ref[3])))
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "seg_time.json"), "w"), indent=1)
