#!/usr/bin/env python3
"""GPU: aum_gemm_tn (csrc/gemm_kernels.h) against the library GEMM the step used before it, on the four K-contiguous projection GEMMs of
an AuM-Base layer at the bench shape (32 832 tokens): parity against an fp64 product on sampled rows, then interleaved timing rounds
(HIP events, median and min) in one process.  Writes gpurun_out/gemm_probe.json.   python tools/gemm_probe.py [--tokens N] [--rounds R]"""
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
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tunable", type=int, default=1)
    ap.add_argument("--shapes", default="", help="comma-separated subset of in_proj_fwd,out_proj_fwd,out_proj_dgrad,in_proj_dgrad")
    a = ap.parse_args()
    if a.tunable:
        from aum import tunable
        tunable.enable()
    dev = "cuda"
    torch.manual_seed(0)
    M = a.tokens
    shapes = [("in_proj_fwd", 768, 3072, False), ("out_proj_fwd", 1536, 768, False), ("out_proj_dgrad", 768, 1536, True),
              ("in_proj_dgrad", 3072, 768, True)]
    if a.shapes:
        shapes = [sh for sh in shapes if sh[0] in a.shapes.split(",")]
    res = {"tokens": M, "shapes": {}}
    for name, K, N, dgrad in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)          # (n, k): the form aum_gemm_tn takes
        w_lib = wt.t().contiguous() if dgrad else wt                                  # data gradient: the library multiplies by W (k, n) as stored
        lib_fn = (lambda: torch.matmul(x, w_lib)) if dgrad else (lambda: torch.matmul(x, w_lib.t()))
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        hip_fn = lambda: aum_hip.gemm_tn(x, wt, out=out)
        out0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        hip0_fn = lambda: aum_hip.gemm_tn(x, wt, out=out0, flags=aum_hip.GEMM_LOCKSTEP)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        hip1_fn = lambda: aum_hip.gemm_tn(x, wt, out=out1, flags=aum_hip.GEMM_STAGGERED)
        outw = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        hipw_fn = lambda: aum_hip.gemm_tn(x, wt, out=outw, split_tail=False)          # whole tiles only (the default splits a half-empty tail round)
        y = hipw_fn().clone()
        ysk = hip_fn().clone()
        y_lib = lib_fn()
        rows = torch.cat([torch.arange(0, 300, device=dev), torch.randint(0, M, (400,), device=dev), torch.arange(M - 300, M, device=dev)])
        ref = x[rows].double() @ wt.double().t()
        scale = ref.abs().max().item()
        err = ((ysk[rows].double() - ref).abs().max().item()) / scale
        err_lib = ((y_lib[rows].double() - ref).abs().max().item()) / scale
        same = float((y == y_lib).float().mean().item())
        assert torch.equal(hip0_fn(), y) and torch.equal(hip1_fn(), y)
        extra = {}
        for fl in [int(v) for v in os.environ.get("GEMM_PROBE_FLAGS", "4").split(",") if v]:
            o_ = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            extra[f"hip_f{fl}"] = (lambda fl=fl, o_=o_: aum_hip.gemm_tn(x, wt, out=o_, flags=fl))
            assert torch.equal(extra[f"hip_f{fl}"](), y), fl
        t = {"hip": [], "hip_whole": [], "hip_lockstep": [], "hip_staggered": [], "lib": [], **{k: [] for k in extra}}
        for fn in (hip_fn, hipw_fn, hip0_fn, hip1_fn, lib_fn):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for r in range(a.rounds):
            for key, fn in (("hip", hip_fn), ("hip_whole", hipw_fn), ("hip_lockstep", hip0_fn), ("hip_staggered", hip1_fn), ("lib", lib_fn)) + tuple(extra.items()):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t[key].append(e0.elapsed_time(e1) / a.iters * 1e3)
        flops = 2.0 * M * N * K
        ent = {"m": M, "n": N, "k": K, "rel_err_vs_fp64": err, "lib_rel_err_vs_fp64": err_lib, "bitwise_equal_frac_vs_lib": same}
        for key in t:
            med, mn = statistics.median(t[key]), min(t[key])
            ent[key] = {"us_median": round(med, 1), "us_min": round(mn, 1), "tflops_median": round(flops / med / 1e6, 1)}
        res["shapes"][name] = ent
        print(name, " ".join(f"{k}={ent[k]['us_median']}us/{ent[k]['tflops_median']}TF" for k in t), "err", f"{err:.2e}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
