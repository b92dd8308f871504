#!/usr/bin/env python3
"""Same-box A/B of the one-row backward scan: the default library against a build variant, e.g. the paired LDS tile layout
(-DAUM_SCANH_PAIRED=1, `paired`) or the batched tile update (-DAUM_SCANH_RMW_BATCH=1, `rmwbatch`).  usage: ab_paired.py <variant>
Expects audio-mamba-aum_amd/aum_hip/libaum_hip_<variant>.so next to the default library:
  AUM_EXTRA_CXXFLAGS=-DAUM_SCANH_PAIRED=1 python audio-mamba-aum_amd/csrc/build.py && \
  cp audio-mamba-aum_amd/aum_hip/libaum_hip.so audio-mamba-aum_amd/aum_hip/libaum_hip_paired.so && python audio-mamba-aum_amd/csrc/build.py
Prints the bidirectional backward's time at the AuM-Base shape for both, alternating, and the largest difference of their gradients."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def main():
    d = os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip")
    variant = sys.argv[1] if len(sys.argv) > 1 else "paired"        # libaum_hip_<variant>.so, built with the variant's -D flag
    libs = {"default": aum_hip.Lib(os.path.join(d, "libaum_hip.so")), variant: aum_hip.Lib(os.path.join(d, f"libaum_hip_{variant}.so"))}
    torch.manual_seed(0)
    Bsz, E, L, N, dt, dev = 64, 1536, 513, 16, torch.bfloat16, "cuda"
    mk = lambda: torch.randn(E, Bsz, L, device=dev).to(dt).permute(1, 0, 2)
    u, z, dout = mk(), mk(), mk()
    delta = (0.5 * torch.randn(E, Bsz, L, device=dev)).to(dt).permute(1, 0, 2)
    Bm, Cm = torch.randn(Bsz, 1, N, L, device=dev).to(dt), torch.randn(Bsz, 1, N, L, device=dev).to(dt)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b, D, bias = A * 1.05, torch.ones(E, device=dev), torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
    _, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, lib=libs["default"])
    run = lambda lib: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b, lib=lib)
    g = {k: run(lib) for k, lib in libs.items()}
    diff = {k: float((g["default"][k].float() - g[variant][k].float()).abs().max() / (g["default"][k].float().abs().max() + 1e-30))
            for k in g["default"] if g["default"][k] is not None}
    times = {k: [] for k in libs}
    for rep in range(3):
        for k, lib in libs.items():
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                run(lib)
            e.record()
            torch.cuda.synchronize()
            times[k].append(round(s.elapsed_time(e) / 8, 4))
    out = {"scan_bwd_bidir_ms": times, "max_rel_diff": diff}
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"ab_{variant}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
