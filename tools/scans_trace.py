#!/usr/bin/env python3
"""Phase time stamps of the state-per-wave scan backward (scan_state_kernels.h, AUM_DBG_TRACE): cycles per phase, per wave role."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip

bidir = "--bidir" in sys.argv
dt = torch.bfloat16
Bsz, E, L, N = 64, 1536, 513, 16
dev = "cuda"
torch.manual_seed(0)
mk = lambda: torch.randn(E, Bsz, L, device=dev).to(dt).permute(1, 0, 2)
u, z, dout = mk(), mk(), mk()
delta = 0.5 * mk()
Bm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
Cm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
A_b = A * 1.05 if bidir else None
D = torch.ones(E, device=dev)
bias = torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
ck = aum_hip.scan_lane_ckpt(u, N, bidir)
_, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, x_lane=ck)
# capture the workspace: monkeypatch torch.empty is overkill; re-run scan_bwd and grab ws through a hook on _launch
captured = {}
orig = aum_hip._launch
def hook(fn, args, t, lib, name, meta=None):
    if name.startswith("scan_bwd"):
        captured["ws_ptr"], captured["ws_bytes"] = args.workspace, args.workspace_bytes
        captured["keep"] = args
    return orig(fn, args, t, lib, name, meta)
aum_hip._launch = hook
for abl in (64 + 128,):
    aum_hip.debug.ablate = abl
    real_empty = torch.empty
    ws_holder = {}
    def spy_empty(*a, **k):
        t = real_empty(*a, **k)
        if len(a) == 1 and isinstance(a[0], tuple) and k.get("dtype") == torch.float32 and t.numel() > (1 << 22):
            ws_holder["ws"] = t
        return t
    torch.empty = spy_empty
    g = aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b, x_lane=ck)
    torch.empty = real_empty
    torch.cuda.synchronize()
    ws = ws_holder["ws"]
    NW, IT, SL = 16, 24, 8
    tr = ws[-NW * IT * SL:].view(torch.int32).cpu().numpy().astype(np.int64).reshape(NW, IT, SL) & 0xffffffff
    names = ["P3", "P2", "P1", "fetch", "barrier1", "publish", "barrier2"]
    for w in (0, 1, 4, 8, 9, 12, 15):
        d = np.diff(tr[w, 8:20, :], axis=1) & 0xffffffff
        nxt = (tr[w, 9:21, 0] - tr[w, 8:20, 7]) & 0xffffffff
        it_total = (tr[w, 9:21, 0] - tr[w, 8:20, 0]) & 0xffffffff
        print(f"wave {w:2d}: " + "  ".join(f"{n}={int(np.median(d[:, i]))}" for i, n in enumerate(names)) + f"  | iteration={int(np.median(it_total))} cycles")
