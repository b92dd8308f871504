#!/usr/bin/env python3
"""GPU: where do two identical launches of the token-major scan backward differ?  (bench shape; prints tensor, count, time steps)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip
lib = aum_hip.get()
torch.manual_seed(0)
Bsz, L, E, N = 64, int(os.environ.get("PROBE_L", "513")), 1536, 16
dev = "cuda"
xz = torch.randn(Bsz, L, 2 * E, device=dev).bfloat16()
u, z = torch.randn(Bsz, L, E, device=dev).bfloat16(), xz[:, :, E:]
dl = (0.5 * torch.randn(Bsz, L, E, device=dev)).bfloat16()
bc = torch.randn(Bsz, L, 48 + 2 * N, device=dev).bfloat16()
Bm, Cm = bc[:, :, 48:48 + N], bc[:, :, 48 + N:]
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
A_b, D, bias = A * 1.05, torch.ones(E, device=dev), torch.full((E,), -4.0, device=dev)
dout = torch.randn(Bsz, L, E, device=dev).bfloat16()
ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
ref = None
for it in range(int(os.environ.get("PROBE_N", "4"))):
    o, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=lib)
    ckc = ck.clone()
    g = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=lib)
    cur = {k: v.clone() for k, v in g.items() if v is not None and not k.startswith("_")}
    for k, v in g.items():
        if k.startswith("_") and torch.is_tensor(v):
            n = Bsz * L * 48 * 32
            cur["ws_dbc"] = v.flatten()[:n].clone().view(Bsz, L, 48, 32)
    cur["ckpt"] = ckc
    if ref is None:
        ref = cur
        continue
    for k in ref:
        d = (ref[k] != cur[k])
        if d.any():
            idx = d.nonzero()
            print(it, k, tuple(ref[k].shape), int(d.sum()), "first", idx[0].tolist(), flush=True)
            if idx.shape[1] == 4:
                ts = sorted(set(idx[:, 1].tolist())); sl = sorted(set(idx[:, 2].tolist())); cs = sorted(set(idx[:, 3].tolist()))
                print("   ws: t", len(ts), ts[:8], ts[-8:], "slots", sl, "cols", cs)
                for half, name in ((idx[:, 2] >= 24, "phase-1 partial slots"), (idx[:, 2] < 24, "final slots")):
                    sub = idx[half]
                    if len(sub):
                        t2 = sorted(set(sub[:, 1].tolist()))
                        print("     ", name, len(sub), "t sample", t2[:8], t2[-8:])
            if idx.shape[1] == 3:
                ts = sorted(set(idx[:, 1].tolist())); cs = sorted(set(idx[:, 2].tolist())); bs = sorted(set(idx[:, 0].tolist()))
                print("   t:", len(ts), "min", ts[0], "max", ts[-1], "even", sum(1 for t in ts if t % 2 == 0), "sample", ts[:6], ts[-6:], " cols:", cs, " batches:", len(bs))
                rel = ((ref[k] - cur[k]).abs() / (ref[k].abs() + 1e-6))[d]
                print("   rel diff max", float(rel.max()), "median", float(rel.median()))
print("done")
