#!/usr/bin/env python3
"""Where and when the waves of the time-serial scan ran (a -DAUM_SCANT_TRACE=1 build: every wave leaves HW_ID, XCC_ID and its
begin / end s_memrealtime stamps in the buffer passed as `ckpt`).  usage: tm_trace.py <variant> [uni|bidir]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def bwd_trace(lib, u, dl, A, Bm, Cm, z, Bsz, E, L, N, dev):
    """backward of the bidirectional scan: per-wave placement and the shader-clock cycles of each part of scant_bwd_run"""
    import numpy as np
    ref = aum_hip.get()
    D, A_b = torch.ones(E, device=dev), A * 1.05
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
    _, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, None, False, A_b=A_b, want_out_pre=True, ckpt=ck, lib=ref)
    dout = torch.randn(Bsz, L, E, device=dev).to(u.dtype)
    for rep in range(3):
        r = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, None, dout, pre, ck, False, A_b=A_b, lib=lib)
        torch.cuda.synchronize()
    nw = (Bsz * (E // 64) + 2) // 3 * 4          # three Fo-Bi pairs per workgroup of four waves
    tr = r["_ws"].view(torch.int64)[-nw * 16:].view(nw // 4, 4, 16).cpu().numpy()
    hw, xcc, t0, t1 = tr[..., 0], tr[..., 1] & 0xF, tr[..., 2], tr[..., 3]
    dur = (t1 - t0) / 100.0
    print(f"bwd: {nw} waves; kernel span {(t1.max() - t0.min()) / 100:.1f} us; wave duration us 5/50/95/100 %: "
          + " ".join(f"{np.percentile(dur, q):.0f}" for q in (5, 50, 95, 100)))
    cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xF)
    per_simd = collections.Counter((cuid * 4 + ((hw >> 4) & 3)).flatten().tolist())
    print("waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
    names = ["prologue", "block start: requests", "pass: staging", "pass: forward sweep", "pass: dC butterfly", "pass: reverse sweep",
             "pass: dB butterfly + carries", "block end: du/ddelta", "block end: stores, B/C", "block start: wait for the tensors",
             "block start: B/C, per-step values", ""]
    acc = tr[..., 4:16].reshape(-1, 12).astype(np.float64)
    tot = acc.sum(1).mean()
    print(f"mean shader-clock cycles per wave {tot:.0f} ({tot / dur.mean() / 1e3:.2f} GHz if the wave were stamping all the time)")
    for k, n in enumerate(names):
        if n:
            print(f"  {n:32s} {acc[:, k].mean():12.0f} cycles  {100 * acc[:, k].mean() / tot:5.1f} %")


def main():
    variant, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bidir")
    lib = aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", f"libaum_hip_{variant}.so"))
    torch.manual_seed(0)
    Bsz, E, L, N, dt, dev = 64, 1536, 513, 16, torch.bfloat16, "cuda"
    xz = torch.randn(Bsz, L, 2 * E, device=dev).to(dt)
    u = torch.randn(Bsz, L, E, device=dev).to(dt)
    z = xz[:, :, E:]
    dl = torch.nn.functional.softplus(0.5 * torch.randn(Bsz, L, E, device=dev) - 4.0).to(dt)
    xdbl = torch.randn(Bsz, L, 48 + 2 * N, device=dev).to(dt)
    Bm, Cm = xdbl[:, :, 48:48 + N], xdbl[:, :, 48 + N:]
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1)
    if mode == "bwd":
        return bwd_trace(lib, u, dl, A, Bm, Cm, z, Bsz, E, L, N, dev)
    bidir = mode == "bidir"
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, bidir, dev)          # used as the trace buffer by this build
    for rep in range(3):
        ck.zero_()
        aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, torch.ones(E, device=dev), z, None, False, A_b=A * 1.05 if bidir else None, want_out_pre=True, ckpt=ck, lib=lib)
        torch.cuda.synchronize()
    nwg = Bsz * (E // 64) // (2 if bidir else 4)
    tr = ck.view(torch.int64).flatten()[: nwg * 4 * 12].view(nwg, 4, 12).cpu().numpy()
    hw, xcc, t0, t1 = tr[..., 0], tr[..., 1] & 0xF, tr[..., 2], tr[..., 3]
    tmin = t0.min()
    print(f"{mode}: {nwg} workgroups x 4 waves; kernel span {(t1.max() - tmin) / 100:.1f} us (s_memrealtime, 100 MHz)")
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    cuid = xcc * 1000 + se * 100 + sh * 10 + cu
    per_cu = collections.Counter(cuid[:, 0].tolist())
    print("distinct CUs used:", len(per_cu), " workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
    per_simd = collections.Counter((cuid * 4 + simd).flatten().tolist())
    print("waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
    dur = (t1 - t0) / 100.0
    start = (t0 - tmin) / 100.0
    print(f"wave duration us: min {dur.min():.1f} median {float(sorted(dur.flatten())[dur.size // 2]):.1f} max {dur.max():.1f};  start offset us: median "
          f"{float(sorted(start.flatten())[start.size // 2]):.1f} max {start.max():.1f}")
    late = (start[:, 0] > 5.0).sum()
    print(f"workgroups starting more than 5 us after the first: {late}")
    by = collections.defaultdict(list)
    for w in range(nwg):
        for k in range(4):
            by[per_simd[int(cuid[w, k] * 4 + simd[w, k])]].append(dur[w, k])
    for n, v in sorted(by.items()):
        print(f"  SIMDs holding {n} waves: {len(v)} waves, mean duration {sum(v) / len(v):.1f} us")
    import numpy as np
    print("  duration percentiles us (5/25/50/75/95/100):", " ".join(f"{np.percentile(dur, q):.0f}" for q in (5, 25, 50, 75, 95, 100)))
    print("  mean duration by XCD:", " ".join(f"{dur[xcc == k].mean():.0f}" for k in range(8)))
    print("  mean duration by wave slot in the workgroup:", " ".join(f"{dur[:, k].mean():.0f}" for k in range(4)))
    print("  mean end time by XCD:", " ".join(f"{((t1 - tmin) / 100.0)[xcc == k].mean():.0f}" for k in range(8)))
    wg_end = ((t1 - tmin) / 100.0).max(1)
    order = np.argsort(wg_end)[-5:]
    print("  last workgroups to finish:", [(int(w), int(xcc[w, 0]), int(se[w, 0]), int(cu[w, 0]), round(float(wg_end[w]), 1)) for w in order])
    cu_end = collections.defaultdict(float)
    for w in range(nwg):
        cu_end[int(cuid[w, 0])] = max(cu_end[int(cuid[w, 0])], float(wg_end[w]))
    v = np.array(list(cu_end.values()))
    print("  per-CU finish time us percentiles (5/50/95/100):", " ".join(f"{np.percentile(v, q):.0f}" for q in (5, 50, 95, 100)))
    acc = tr[..., 4:9].astype(float).reshape(-1, 5).mean(0)
    names = ["prologue", "request", "steps", "flush", "park"]
    print("  shader-clock cycles per wave (mean): " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, acc)) + f"  total {acc.sum():.0f}")


if __name__ == "__main__":
    main()
