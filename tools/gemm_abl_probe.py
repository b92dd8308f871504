#!/usr/bin/env python3
"""GPU: where a projection GEMM's time goes -- aum_gemm_tn (the paced-store kernel) against ablation builds of it (tools/build_gemm_variant.sh
ps<bits> -DAUM_PS_ABL=<bits>; wrong results, timing only), any other variant library, other schedules (--vflags) and the library GEMM, all in
one process, interleaved rounds; --check name,... asserts bit-equality with the default build.
  python tools/gemm_abl_probe.py [--variants ps1,ps2,...] [--flags 0] [--vflags default_lockstep=1,...]
prints one line per shape: us (median) per build.  Writes gpurun_out/gemm_abl_probe.json."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=64 * 513)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", default="", help="comma list of variant libraries (tools/build_gemm_variant.sh <name> -DAUM_PS_ABL=<bits> ...)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--vflags", default="", help="name=flags,... : per-variant flags (default --flags)")
    ap.add_argument("--lib", type=int, default=1, help="also time the library GEMM (TunableOp picks)")
    ap.add_argument("--check", default="", help="comma list of builds whose result must equal the default build's bit for bit")
    a = ap.parse_args()
    if a.lib:
        from aum import tunable
        tunable.enable()
    vdir = os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants")
    vflags = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.vflags.split(",") if kv)
    libs = {"default": (aum_hip.get(), a.flags)}
    for v in [v for v in a.variants.split(",") if v]:
        name, path = v, os.path.join(vdir, f"libaum_hip_{v.split(':')[0]}.so")
        libs[name] = (aum_hip.Lib(path, host=False), vflags.get(name, a.flags))
    for name, fl in vflags.items():
        if name.startswith("default"):
            libs[name] = (aum_hip.get(), fl)
    dev = "cuda"
    torch.manual_seed(0)
    M = a.tokens
    shapes = [("in_proj_fwd", 768, 3072), ("out_proj_fwd", 1536, 768), ("out_proj_dgrad", 768, 1536), ("in_proj_dgrad", 3072, 768)]
    res = {}
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fns = {k: (lambda L=L, fl=fl: aum_hip.gemm_tn(x, wt, out=out, lib=L, flags=fl)) for k, (L, fl) in libs.items()}
        if a.lib:
            fns["library"] = lambda: torch.matmul(x, wt.t())
        if a.check:
            fns["default"]()
            ref = out.clone()
            for k in a.check.split(","):
                out.zero_()
                fns[k]()
                torch.cuda.synchronize()
                bad = (out != ref)
                print(f"  check {k}: {'bit-equal' if not bad.any() else f'{int(bad.sum())} of {out.numel()} differ, rows ' + str(bad.any(1).nonzero().flatten()[:8].tolist())}", flush=True)
        t = {k: [] for k in fns}
        for fn in fns.values():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / a.iters * 1e3)
        res[name] = {k: round(statistics.median(v), 1) for k, v in t.items()}
        print(f"{name:15s} n={N:5d} k={K:5d} " + "  ".join(f"{k}={v}" for k, v in res[name].items()), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_abl_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
