#!/usr/bin/env python3
"""Same-box A/B of the time-serial scan kernels: the default library against build variants (tools/build_variant.sh <name> <flags>
-> audio-mamba-aum_amd/aum_hip/variants/libaum_hip_<name>.so).  usage: tm_ab.py [fwd|bwd] <variant> [<variant> ...]
Times the AuM-Base shape (B = 64, E = 1536, L = 513, bf16), three alternating rounds, and prints the per-library median."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    what = sys.argv[1]
    d = os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip")
    libs = {"default": aum_hip.Lib(os.path.join(d, "libaum_hip.so"))}
    for v in sys.argv[2:]:
        libs[v] = aum_hip.Lib(os.path.join(d, "variants", f"libaum_hip_{v}.so"))
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
    ck2 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
    ck1 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, dev)
    cfgs = {}
    if what == "fwd":
        cfgs["uni_nosp"] = lambda lib: aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, lib=lib)
        cfgs["bidir_nosp"] = lambda lib: aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, lib=lib)
        cfgs["bidir_sp"] = lambda lib: aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, lib=lib)
        cfgs["bidir_nosp_pre"] = lambda lib: aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, want_out_pre=True, lib=lib)
        cfgs["bidir_nosp_train"] = lambda lib: aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, want_out_pre=True, ckpt=ck2, lib=lib)
        cfgs["bidir_sp_train"] = lambda lib: aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck2, lib=lib)      # the bench's launch
    else:
        _, pre = aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, A_b=A_b, want_out_pre=True, ckpt=ck2)
        _, pre1 = aum_hip.scan_tm_fwd(u, dsp, A, Bm, Cm, D, z, None, False, want_out_pre=True, ckpt=ck1)
        cfgs["bwd_bidir_nosp"] = lambda lib: aum_hip.scan_tm_bwd(u, dsp, A, Bm, Cm, D, z, None, dout, pre, ck2, False, A_b=A_b, lib=lib)
        cfgs["bwd_uni_nosp"] = lambda lib: aum_hip.scan_tm_bwd(u, dsp, A, Bm, Cm, D, z, None, dout, pre1, ck1, False, lib=lib)
        _, pre_sp = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck2)
        cfgs["bwd_bidir_sp"] = lambda lib: aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre_sp, ck2, True, A_b=A_b, lib=lib)
    res = {c: {k: [] for k in libs} for c in cfgs}
    # the library timed first in a group runs ~3 % behind an identical copy timed later (gpurun_out/r6_tm_ab_fwd_pairs.txt: "default" against "same"):
    # the order rotates with the round, and there are as many rounds as libraries (at least three)
    names = list(libs)
    for rep in range(max(3, len(names))):
        order = names[rep % len(names):] + names[:rep % len(names)]
        for c, fn in cfgs.items():
            for k in order:
                res[c][k].append(timeit(lambda: fn(libs[k])))
    def flat(o):
        if isinstance(o, dict):
            return [(k, v) for k, v in sorted(o.items()) if v is not None]
        return [(str(i), v) for i, v in enumerate(o) if v is not None]

    for c, fn in cfgs.items():
        line = {"config": c, **{k: round(statistics.median(v), 4) for k, v in res[c].items()}}
        # are the variants' results the default's, bit for bit?  (a re-cut that keeps every IEEE operation and its order must be)
        ref = [(k, v.clone()) for k, v in flat(fn(libs["default"]))]
        for name, lib in libs.items():
            if name == "default":
                continue
            got = dict(flat(fn(lib)))
            bad = [k for k, v in ref if not torch.equal(v, got[k])]
            line["equal_" + name] = "bit-equal" if not bad else "differs: " + ",".join(bad)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
