#!/usr/bin/env python3
"""Per-kernel timing at AuM-Base shapes (B=64, E=1536, L=513, N=16) through the C ABI.
HIP events on the current stream; prints one JSON line per kernel with achieved algorithmic GB/s
(byte formulas of SURVEY.md 8d)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dmodel", type=int, default=768)
    ap.add_argument("--len", type=int, default=513)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--layout", default="dmajor", help="dmajor: [E][B][L] storage (the package's layout); bmajor: [B][E][L]")
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[a.dtype]
    s = 2 if dt != torch.float32 else 4
    Bsz, E, L, N = a.batch, 2 * a.dmodel, a.len, 16
    dev = "cuda"
    torch.manual_seed(0)
    # d-major layout [E][B][L] viewed as (B,E,L), as the host package stores it
    mk = lambda: torch.randn(E, Bsz, L, device=dev).to(dt).permute(1, 0, 2)
    if a.layout == "bmajor":
        mk = lambda: torch.randn(Bsz, E, L, device=dev).to(dt)
    u, z, dout = mk(), mk(), mk()
    delta = 0.5 * mk()
    Bm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
    Cm = torch.randn(Bsz, 1, N, L, device=dev).to(dt)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b = A * 1.05
    D = torch.ones(E, device=dev)
    bias = torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
    T = Bsz * E * L
    res = []

    def rec(name, sec, alg_bytes):
        r = {"kernel": name, "ms": round(sec * 1e3, 4), "alg_GB": round(alg_bytes / 1e9, 4),
             "alg_GBps": round(alg_bytes / sec / 1e9, 1), "frac_of_8TBps": round(alg_bytes / sec / 8e12, 4)}
        res.append(r)
        print(json.dumps(r), flush=True)

    want = lambda n: (not a.only) or any(o in n for o in a.only.split(","))
    if want("hbm_copy"):
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
        dst = torch.empty_like(src)
        rec("hbm_copy_1GiB", timeit(lambda: aum_hip.hbm_copy(src, dst)), 2 * src.numel())
        rec("torch_copy_1GiB", timeit(lambda: dst.copy_(src)), 2 * src.numel())
        del src, dst
    if want("frontend"):      # waveform -> tokens: log-mel kernel alone, the two-stage path, and the one-launch path
        from aum.frontend import FbankTables
        tabs = FbankTables(dev).tables
        wave = (torch.randn(Bsz, 160000, device=dev) * 0.1).clamp_(-1, 1)
        Dm = a.dmodel
        w = torch.randn(Dm, 256, device=dev) * 0.06
        w16 = w.to(torch.bfloat16)
        fbias = torch.zeros(Dm, device=dev)
        pos = torch.randn(512, Dm, device=dev) * 0.02
        cls_row = torch.zeros(Dm, device=dev)
        wave_bytes, spec_bytes, tok_bytes = wave.numel() * 4, Bsz * 1024 * 128 * 4, Bsz * 513 * Dm * 4
        rec("fbank_fwd", timeit(lambda: aum_hip.fbank_fwd(wave, tabs, 1024, -4.27, 4.57)), wave_bytes + spec_bytes)

        def two_stage():
            spec = aum_hip.fbank_fwd(wave, tabs, 1024, -4.27, 4.57)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                cols = spec.transpose(1, 2).reshape(Bsz, 8, 16, 64, 16).permute(0, 1, 3, 2, 4).reshape(Bsz * 512, 256)
                x = torch.nn.functional.linear(cols, w, fbias).reshape(Bsz, 512, Dm) + pos
            return torch.cat((x[:, :256], cls_row.expand(Bsz, 1, Dm), x[:, 256:]), dim=1)
        rec("frontend_two_stage", timeit(two_stage), wave_bytes + tok_bytes)
        for save in (False, True):
            rec("frontend_tokens" + ("_save_patches" if save else ""),
                timeit(lambda: aum_hip.frontend_tokens(wave, tabs, 1024, -4.27, 4.57, w16, fbias, pos, cls_row, 256, save_patches=save)),
                wave_bytes + tok_bytes + (Bsz * 512 * 256 * 2 if save else 0))
    bc = 2 * Bsz * N * L * s
    fused = L <= aum_hip.get().max_single_pass_len      # longer rows: chunked kernels, one launch per direction
    if want("scan_fwd"):
        rec("scan_fwd_uni", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True)), 4 * T * s + bc)
    if want("scan_fwd") and not fused:
        rec("scan_fwd_uni_train", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, want_out_pre=True)), 5 * T * s + bc)
        rec("scan_fwd_rev_train", timeit(lambda: aum_hip.scan_fwd(u, delta, A_b, Bm, Cm, D, z, bias, True, True, want_out_pre=True)), 5 * T * s + bc)
    if want("scan_fwd") and fused:
        rec("scan_fwd_bidir", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b)), 4 * T * s + bc)
        rec("scan_fwd_bidir_train", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True)), 5 * T * s + bc)
    if want("scan_bwd"):
        _, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b if fused else None, want_out_pre=True)
        bw_bytes = 8 * T * s + bc + 2 * Bsz * N * L * 4
        rec("scan_bwd_uni", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True), iters=10), bw_bytes)
    if want("scan_bwd") and not fused:
        rec("scan_bwd_rev", timeit(lambda: aum_hip.scan_bwd(u, delta, A_b, Bm, Cm, D, z, bias, dout, pre, True, True), iters=10), bw_bytes)
    if want("scan_ck") and fused and L == 513:      # row kernels with the lane-entry checkpoint (+ its bytes in the algorithmic count)
        ck2, ck1 = aum_hip.scan_lane_ckpt(u, N, True), aum_hip.scan_lane_ckpt(u, N, False)
        rec("scan_fwd_bidir_train_ck", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, x_lane=ck2)),
            5 * T * s + bc + ck2.numel() * 4)
        rec("scan_fwd_uni_train_ck", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, want_out_pre=True, x_lane=ck1)),
            5 * T * s + bc + ck1.numel() * 4)
        _, pre2, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, x_lane=ck2)
        bw_bytes = 8 * T * s + bc + 2 * Bsz * N * L * 4
        rec("scan_bwd_bidir_ck", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre2, True, A_b=A_b, x_lane=ck2), iters=10),
            bw_bytes + ck2.numel() * 4)
        _, pre1, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, want_out_pre=True, x_lane=ck1)
        rec("scan_bwd_uni_ck", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre1, True, x_lane=ck1), iters=10),
            bw_bytes + ck1.numel() * 4)
        for bits, label in ((1, "no_states"), (2, "no_rmw"), (16, "no_step_barrier")):
            aum_hip.debug.ablate = bits
            rec(f"ablate_bwd_bidir_ck_{label}", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre2, True, A_b=A_b, x_lane=ck2), iters=5), 1)
        aum_hip.debug.ablate = 1
        rec("ablate_fwd_bidir_ck_no_states", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, x_lane=ck2), iters=5), 1)
        aum_hip.debug.ablate = 0
    if want("scan_bwd") and fused:
        rec("scan_bwd_bidir", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b), iters=10), bw_bytes)
    if want("scan_tm"):       # time-serial scan on token-major activations (scan_tm_kernels.h), the layout in_proj / x_proj leave
        xz = torch.randn(Bsz, L, 2 * E, device=dev).to(dt)
        ut = torch.randn(Bsz, L, E, device=dev).to(dt)
        zt = xz[:, :, E:]
        dlt = (0.5 * torch.randn(Bsz, L, E, device=dev)).to(dt)
        R = a.dmodel // 16
        xdbl = torch.randn(Bsz, L, R + 2 * N, device=dev).to(dt)
        Bt, Ct = xdbl[:, :, R:R + N], xdbl[:, :, R + N:]
        ck1 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, False, dev)
        ck2 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
        Bx, Cx = Bt, Ct
        dsp = torch.nn.functional.softplus(dlt.float() + bias).to(dt)       # delta as a producer with a fused softplus would leave it
        for tag, dl_, bias_, sp_ in (("", dlt, bias, True), ("_nosp", dsp, None, False)):
            rec("scan_tm_fwd_uni" + tag, timeit(lambda: aum_hip.scan_tm_fwd(ut, dl_, A, Bx, Cx, D, zt, bias_, sp_)), 4 * T * s + bc)
            rec("scan_tm_fwd_bidir" + tag, timeit(lambda: aum_hip.scan_tm_fwd(ut, dl_, A, Bx, Cx, D, zt, bias_, sp_, A_b=A_b)), 4 * T * s + bc)
            rec("scan_tm_fwd_bidir_train" + tag, timeit(lambda: aum_hip.scan_tm_fwd(ut, dl_, A, Bx, Cx, D, zt, bias_, sp_, A_b=A_b, want_out_pre=True, ckpt=ck2)),
                5 * T * s + bc + ck2.numel() * 4)
            rec("scan_tm_fwd_bidir_pre_nockpt" + tag, timeit(lambda: aum_hip.scan_tm_fwd(ut, dl_, A, Bx, Cx, D, zt, bias_, sp_, A_b=A_b, want_out_pre=True)),
                5 * T * s + bc)
            rec("scan_tm_fwd_uni_train" + tag, timeit(lambda: aum_hip.scan_tm_fwd(ut, dl_, A, Bx, Cx, D, zt, bias_, sp_, want_out_pre=True, ckpt=ck1)),
                5 * T * s + bc + ck1.numel() * 4)
    if want("scan_tmb"):      # backward of the time-serial scan (delta ready: the producer applied the softplus)
        ut = torch.randn(Bsz, L, E, device=dev).to(dt)
        xz = torch.randn(Bsz, L, 2 * E, device=dev).to(dt)
        zt = xz[:, :, E:]
        R = a.dmodel // 16
        xdbl = torch.randn(Bsz, L, R + 2 * N, device=dev).to(dt)
        Bt, Ct = xdbl[:, :, R:R + N], xdbl[:, :, R + N:]
        dsp = torch.nn.functional.softplus(0.5 * torch.randn(Bsz, L, E, device=dev) + bias).to(dt)
        dot = torch.randn(Bsz, L, E, device=dev).to(dt)
        ck2 = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev)
        _, pre = aum_hip.scan_tm_fwd(ut, dsp, A, Bt, Ct, D, zt, None, False, A_b=A_b, want_out_pre=True, ckpt=ck2)
        bw_bytes = 8 * T * s + bc + 2 * Bsz * N * L * 4 + ck2.numel() * 4
        rec("scan_tm_bwd_bidir", timeit(lambda: aum_hip.scan_tm_bwd(ut, dsp, A, Bt, Ct, D, zt, None, dot, pre, ck2, False, A_b=A_b), iters=5, warm=1), bw_bytes)
    if want("conv"):
        w = torch.randn(E, 4, device=dev)
        b = torch.randn(E, device=dev)
        rec("conv_fwd", timeit(lambda: aum_hip.conv1d_fwd(u, w, b)), 2 * T * s)
        rec("conv_bwd", timeit(lambda: aum_hip.conv1d_bwd(u, w, b, dout)), 3 * T * s)
    if want("norm"):
        M, C = Bsz * L, a.dmodel
        x = torch.randn(M, C, device=dev).to(dt)
        r = torch.randn(M, C, device=dev)
        wn = torch.ones(C, device=dev)
        rec("rmsnorm_fwd", timeit(lambda: aum_hip.rmsnorm_fwd(x, wn, r, 1e-5)), M * C * (2 * s + 8))
        y, rstd, ro = aum_hip.rmsnorm_fwd(x, wn, r, 1e-5)
        dy = torch.randn(M, C, device=dev).to(dt)
        rec("rmsnorm_bwd", timeit(lambda: aum_hip.rmsnorm_bwd(dy, ro, wn, rstd, r, True, x_dtype=dt)), M * C * (2 * s + 12))
    if want("proj") and dt != torch.float32:
        R, rt, ntok = a.dmodel // 16, a.dmodel // 16 + 2 * N, Bsz * L
        conv2 = torch.randn(E, ntok, device=dev).to(dt)
        dd2 = torch.randn(E, ntok, device=dev).to(dt)
        w_x = (torch.randn(rt, E, device=dev) / E ** 0.5).to(dt)
        w_dt = (torch.randn(E, R, device=dev) / R ** 0.5).to(dt)
        w_xT, w_dtT = w_x.t().contiguous(), w_dt.t().contiguous()
        dBf, dCf = torch.randn(Bsz, N, L, device=dev), torch.randn(Bsz, N, L, device=dev)
        dconv = torch.zeros(E, ntok, device=dev).to(dt)
        act = E * ntok * s
        rec("proj_fwd", timeit(lambda: aum_hip.proj_fwd(conv2, w_x, w_dt, N)), 2 * act + rt * ntok * s)
        xd, _ = aum_hip.proj_fwd(conv2, w_x, w_dt, N)
        rec("proj_bwd_data", timeit(lambda: aum_hip.proj_bwd_data(dd2, w_dtT, w_xT, dBf, dCf, dconv, L)), 3 * act + rt * ntok * s)
        dxd = aum_hip.proj_bwd_data(dd2, w_dtT, w_xT, dBf, dCf, dconv, L)
        rec("proj_bwd_weight_x", timeit(lambda: aum_hip.proj_bwd_weight(conv2, dxd, True)), act + rt * ntok * s)
        rec("proj_bwd_weight_dt", timeit(lambda: aum_hip.proj_bwd_weight(dd2, xd[:R], False)), act + R * ntok * s)
        # the library GEMMs they replace (same operands, hipBLASLt through torch)
        rec("lib_x_proj_fwd", timeit(lambda: torch.matmul(conv2.t(), w_x.t())), act)
        xtm = torch.matmul(conv2.t(), w_x.t())
        rec("lib_dt_proj_fwd", timeit(lambda: torch.matmul(w_dt, xtm[:, :R].t())), act)
        rec("lib_dt_proj_dgrad", timeit(lambda: torch.matmul(dd2.t(), w_dt)), act)
        rec("lib_x_proj_dgrad", timeit(lambda: dconv.addmm_(w_x.t(), xtm.t())), 2 * act)
        rec("lib_x_proj_wgrad", timeit(lambda: torch.matmul(xtm.t(), conv2.t())), act)
        rec("lib_dt_proj_wgrad", timeit(lambda: torch.matmul(dd2, xtm[:, :R])), act)
    if want("ablate"):
        _, pre, _ = aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True)
        for bits, label in ((0, "full"), (1, "no_states"), (2, "no_lds_atomics"), (4, "no_partials"), (8, "no_epilogue"),
                            (3, "no_states_no_atomics"), (15, "loads_only"), (16, "no_step_barrier")):
            aum_hip.debug.ablate = bits
            rec(f"ablate_bwd_bidir_{label}", timeit(lambda: aum_hip.scan_bwd(u, delta, A, Bm, Cm, D, z, bias, dout, pre, True, A_b=A_b), iters=5), 1)
        for bits, label in ((0, "full"), (1, "no_states")):
            aum_hip.debug.ablate = bits
            rec(f"ablate_fwd_bidir_{label}", timeit(lambda: aum_hip.scan_fwd(u, delta, A, Bm, Cm, D, z, bias, True, A_b=A_b), iters=5), 1)
        aum_hip.debug.ablate = 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"kbench_{a.dtype}_B{a.batch}.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
