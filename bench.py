#!/usr/bin/env python3
"""bench.py -- clips/s for AuM-Base (d_model=768, 24 Fo-Bi blocks, d_state=16) forward + backward + Adam step on
synthetic 128-mel x 1024-frame spectrograms (BASELINE.json metric; configs[2] at N=1, configs[3] per-GPU batch 64 with
DDP over RCCL at N>1), plus the live roofline of the dominant kernel and the CPU baseline (oracle on the host cores).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus N ...          (launches itself under torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step on one per-GPU batch: fwd (bf16 autocast) -> BCE loss -> bwd (+ DDP bucketed all-reduce
overlapped with backward) -> Adam.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402   (no torch import inside)

tunable.enable(int(os.environ.get("LOCAL_RANK", "0")))     # GEMM solution selection, before torch issues any GEMM
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBPS = 6290.0          # the same guide's measured float4-copy rate: printed beside every fraction of the 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
# What this chip SUSTAINS, measured with the repo's own probes (the attainable bars of VERDICT r5 weak #7, printed beside the spec fractions):
MFMA_BF16_SUSTAINED_TFLOPS = 1890.0     # tools/mfma_probe.hip, random operands, two waves per SIMD: 1 892 TFLOP/s mean (profiles/r05_mfma_probe.txt)
# SIMD time per instruction (ns), tools/valu_probe.hip (profiles/r04_valu_probe.txt) at the waves per SIMD the scan kernels run with
VALU_NS = {2: {"pk": 2.087, "exp": 3.476}, 3: {"pk": 1.979, "exp": 3.459}}


def scan_valu_floor_ms(meta, backward, ndir, n_cu):
    """Arithmetic floor of one scan launch: the packed fp32 operations and exponentials SSI:101-152 (forward) / its adjoint (backward) cannot do
    without, per PAIR of states and step -- forward: exponent argument, delta u B, state fma, y fma (4 packed) + 2 v_exp_f32; backward: exponent
    argument, forward sweep (a x, + delta u B, dy x), reverse sweep (g, g delta u, S1, g w, S2, dA, a g) (11 packed) + 2 v_exp_f32 -- at the
    probe's SIMD time per instruction, over the (state pair, step, wave) items one SIMD executes.  Channel sums, softplus, gates, conversions,
    address arithmetic are NOT in the floor: `valu_floor_frac` = floor / launch time says how much of the launch is the unavoidable arithmetic."""
    batch, dim, length, dstate = meta[:4]
    items_per_simd = batch * (dim // 64) * length * (dstate // 2) * ndir / (n_cu * 4.0)
    c = VALU_NS[2 if backward else 3]
    ns = (11 if backward else 4) * c["pk"] + 2 * c["exp"]
    return items_per_simd * ns * 1e-6


def pmc_record(kind, kernel):
    """HBM traffic / vector-ALU occupancy of `kernel` from the counter passes committed under profiles/ (token-major kernels:
    tools/pmc_tm_bench.sh, channel-major: tools/pmc_job.sh, tools/valu_job.sh), with the commit and date they were taken at"""
    for f in (f"{kind}_tm.json", f"{kind}.json"):
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            d = json.load(open(path))
            if kernel in d:
                return d[kernel], {"file": "profiles/" + f, "commit": d.get("_commit", "round 2"), "date": d.get("_date", "round 2")}
    return None, None


def gemm_probe(dev, ntok, d_model, d_inner, iters=20):
    """the largest GEMM of the step, in_proj forward [ntok, d_model] x [d_model, 2 d_inner] in bf16, timed on the spot (HIP events on
    the current stream) against the dense MFMA peak: the kernel the step runs for this shape (aum_gemm_tn, the hand-written MFMA kernel,
    under the default dispatch) and, next to it, the library GEMM (hipBLASLt with the recorded TunableOp pick) on the same operands"""
    import aum_hip
    import mamba_ssm.ops.selective_scan_interface as ssi
    a = torch.randn(ntok, d_model, device=dev).bfloat16()
    w = torch.randn(2 * d_inner, d_model, device=dev).bfloat16()

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    flop = 2.0 * ntok * d_model * 2 * d_inner
    # the library GEMM exactly as the step issues it (ssi._mm_rows: one launch since round 5 -- a shape with a recorded TunableOp solution;
    # a shape without one would be tuned online here, a few seconds of candidate kernels at the end of every bench run)
    ms_lib = timed(lambda: ssi._mm_rows(a, w.t(), 1))
    use_hip = ssi._hip_gemm_ok(a, w.shape[0], w.shape[1]) and aum_hip.gemm_tn_supported(a, w)
    ms = timed(lambda: aum_hip.gemm_tn(a, w)) if use_hip else ms_lib
    tf = flop / (ms * 1e-3) / 1e12
    # the hand-written kernel on the GEMM the default dispatch gives it: the out_proj data gradient [ntok, d_model] x [d_model, d_inner]
    hip_entry = None
    g = torch.randn(ntok, d_model, device=dev).bfloat16()
    wo = torch.randn(d_model, d_inner, device=dev).bfloat16()           # out_proj.weight as nn.Linear stores it: the library's operand
    wo_t = wo.t().contiguous()                                         # its transposed copy (per-forward weight cache): aum_gemm_tn's
    if aum_hip.gemm_tn_supported(g, wo_t):
        ms_h = timed(lambda: aum_hip.gemm_tn(g, wo_t))
        ms_l = timed(lambda: ssi._mm_rows(g, wo, 4))                   # the recorded shapes of the step (no online tuning here)
        fl = 2.0 * ntok * d_model * d_inner
        hip_entry = {"kernel": f"out_proj data gradient [{ntok}x{d_model}] x [{d_model}x{d_inner}] bf16 (aum_gemm_tn, hand-written MFMA kernel"
                               + ("; the step's default for this shape)" if ssi._hip_gemm_ok(g, d_inner, d_model) else ")"),
                     "avg_launch_ms": round(ms_h, 4), "achieved": round(fl / (ms_h * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(fl / (ms_h * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "frac_of_sustained": round(fl / (ms_h * 1e-3) / 1e12 / MFMA_BF16_SUSTAINED_TFLOPS, 4),
                     "library_gemm_avg_launch_ms": round(ms_l, 4)}
    return {"bound": "mfma", "hand_written": hip_entry,
            "kernel": f"in_proj forward GEMM [{ntok}x{d_model}] x [{d_model}x{2 * d_inner}] bf16 ("
                      + ("aum_gemm_tn, hand-written MFMA kernel" if use_hip else "hipBLASLt via TunableOp") + ")",
            "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
            "sustained": MFMA_BF16_SUSTAINED_TFLOPS, "frac_of_sustained": round(tf / MFMA_BF16_SUSTAINED_TFLOPS, 4),       # tools/mfma_probe.hip's rate: the attainable bar (>= 0.75)
            "avg_launch_ms": round(ms, 4),
            "library_gemm": {"avg_launch_ms": round(ms_lib, 4), "achieved": round(flop / (ms_lib * 1e-3) / 1e12, 1),
                             "kernel": "hipBLASLt via TunableOp (one launch, as in the step), same operands"}}


def scan_alg_bytes(meta, backward):
    """Algorithmic bytes of one scan launch (DESIGN.md "scan"): activation tensors read/written once,
    B/C once, fp32 dB/dC once."""
    batch, dim, length, dstate, s, extra = meta
    T = batch * dim * length
    bc = 2 * batch * dstate * length * s
    if not backward:
        return (4 + (1 if extra else 0)) * T * s + bc              # u, delta, z, out (+ out_pre)
    return 8 * T * s + bc + 2 * batch * dstate * length * 4         # u, delta, z, dout, out_pre, du, ddelta, dz; dB, dC


def cpu_baseline(target_seconds=6.0):
    """The oracle (C restatement of the reference's *_ref arithmetic, oracle/) timed on the host cores: one AuM-Base
    Fo-Bi block (fused add+RMSNorm, in_proj, conv, x/dt proj, two scans, out_proj) forward + backward on a few clips,
    scaled by the 24 blocks of the model.  Reported, never a target."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import oracle as O
    import cases
    cores = O.num_threads()
    Dm, L = 768, 513
    Bc = max(2, min(cores, 8))
    r = np.random.default_rng(0)
    p = cases.inner_params(r, Dm)
    hidden = r.normal(0, 1, (Bc, L, Dm)).astype(np.float32)
    resid = r.normal(0, 1, (Bc, L, Dm)).astype(np.float32)
    w_norm = np.ones(Dm, np.float32)
    w_in = (r.normal(0, 1, (4 * Dm, Dm)) / np.sqrt(Dm)).astype(np.float32)
    dout = r.normal(0, 1, (Bc, L, Dm)).astype(np.float32)

    def block():
        n = O.rmsnorm_fwd(hidden, w_norm, None, resid, 1e-5, "f32")
        xz = (n["y"].reshape(Bc * L, Dm) @ w_in.T).reshape(Bc, L, 4 * Dm).transpose(0, 2, 1)
        xz = np.ascontiguousarray(xz)
        st = O.inner_fwd(xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"], None, p["A"],
                         p["D"], p["dt_bias"], p["A_b"], "f32")
        g = O.inner_full_bwd(st, dout, xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"],
                             None, p["A"], p["D"], p["dt_bias"], p["A_b"], "f32")
        dn = g["dxz"].transpose(0, 2, 1).reshape(Bc * L, 4 * Dm) @ w_in
        _ = n["y"].reshape(Bc * L, Dm).T @ g["dxz"].transpose(0, 2, 1).reshape(Bc * L, 4 * Dm)
        O.rmsnorm_bwd(dn.reshape(Bc, L, Dm), n["residual_out"], w_norm, n["rstd"], dout, False, "f32")

    t0 = time.time()
    block()
    t1 = time.time() - t0
    reps = max(1, min(5, int(target_seconds / max(t1, 1e-3)) - 1))
    ts = []
    for _ in range(reps):
        t0 = time.time()
        block()
        ts.append(time.time() - t0)
    t_block = sorted(ts)[len(ts) // 2]
    return {"value": round(Bc / (24 * t_block), 5), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle (C, OpenMP x{cores}, fp32): 1 of 24 AuM-Base Fo-Bi blocks fwd+bwd on {Bc} clips "
                      f"(L=513), median of {reps} runs = {t_block:.2f} s/block, scaled x24 blocks"}


def cpu_baseline_torch_ref(budget_s=30.0):
    """The baseline north_star names: the reference's pure-PyTorch `selective_scan_ref` (SSI:86-152: an O(L) Python loop over a
    materialised [B,E,L,N] tensor) -- here the package's restatement of it, same loop -- on the host cores: BOTH scan directions of one
    AuM-Base Fo-Bi block (all 1536 channels, L = 513, N = 16, one clip), forward + autograd backward, composed as SSI:499-507 composes
    them (flip, scan, flip, add); scaled by the 24 blocks.  The scans only (no projections, conv, norm).  16 host threads; one quarter-size
    warm-up, then the median of up to three full-size runs (as many as fit the time budget: said in `sample`)."""
    import torch as T
    from mamba_ssm.ops.selective_scan_interface import selective_scan_ref
    T.manual_seed(0)
    n_thr = T.get_num_threads()
    Bc, E, L, N = 1, 1536, 513, 16

    def make(e):
        mk = lambda *s: T.randn(*s, requires_grad=True)
        return dict(u=mk(Bc, e, L), delta=(0.5 * T.randn(Bc, e, L)).requires_grad_(True), z=mk(Bc, e, L), B=mk(Bc, 1, N, L), C=mk(Bc, 1, N, L),
                    A=(-T.arange(1, N + 1, dtype=T.float32).repeat(e, 1)).requires_grad_(True),
                    A_b=(-1.05 * T.arange(1, N + 1, dtype=T.float32).repeat(e, 1)).requires_grad_(True),
                    D=T.ones(e, requires_grad=True), bias=T.full((e,), -4.0, requires_grad=True))

    def block(d):
        t0 = time.time()
        fl = lambda t: t.flip(-1)
        o = selective_scan_ref(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["bias"], True)
        o = o + fl(selective_scan_ref(fl(d["u"]), fl(d["delta"]), d["A_b"], fl(d["B"]), fl(d["C"]), d["D"], fl(d["z"]), d["bias"], True))
        o.sum().backward()
        return time.time() - t0

    # Frozen since round 4: the same workload and thread rule every round.  16 threads (the loop is ~1500 small ops per
    # pass: beyond a few dozen threads they only add synchronisation -- 16 was the fastest of 16 / 32 / all on every box of rounds 3-4),
    # one quarter-size warm-up, then full-size runs while they fit the budget (one run is ~25 s: the default bench stays under a minute).
    thr = min(16, n_thr)
    T.set_num_threads(thr)
    warm = block(make(E // 4))
    d = make(E)
    runs, spent = [], warm
    while len(runs) < 3 and (not runs or spent + runs[-1] < budget_s):
        runs.append(block(d))
        spent += runs[-1]
    t_blk = sorted(runs)[len(runs) // 2]
    T.set_num_threads(n_thr)
    return {"value": round(Bc / (24 * t_blk), 5), "unit": "clips/s", "cores": thr, "kind": "port",
            "sample": f"torch selective_scan_ref loop (package restatement of SSI:86-152, fp32, torch {T.__version__}, {thr} of {os.cpu_count()} "
                      f"host threads): both scan directions of 1 of 24 AuM-Base Fo-Bi blocks, all {E} channels, "
                      f"forward + autograd backward on {Bc} clip (L=513, N=16): quarter-size warm-up ({warm:.1f} s) + median of {len(runs)} full-size "
                      f"run(s) = {t_blk:.2f} s, x24 blocks; scans only (no projections, conv, norm)"}


def step_alg_bytes(batch, length, d_model, depth, s=2):
    """SURVEY.md 8(d) whole-layer estimate under an op-boundary decomposition (norm, in_proj, conv, x_proj, dt_proj, bidirectional
    scan, out_proj): forward 26 H s + 8 H bytes, backward 47 H s + 12 H bytes per layer, H = batch * length * d_model."""
    H = batch * length * d_model
    return depth * ((26 + 47) * H * s + (8 + 12) * H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE.json config 3/4: 64)")
    ap.add_argument("--size", default="base")
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-compress", default="no", choices=["no", "bf16", "fp16"], help="16-bit gradient exchange (N > 1)")
    ap.add_argument("--no-frontend", action="store_true", help="start from spectrograms instead of waveforms")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU, RCCL over xGMI) the way the driver does
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("AUM_BENCH_PRINT_LAUNCH") == "1":      # tests: the command, not the run
            print(json.dumps(cmd))
            return
        raise SystemExit(subprocess.call(cmd))

    import aum_hip
    from aum.model import build_aum
    import mamba_ssm.ops.selective_scan_interface as ssi_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    dev_index = local_rank % torch.cuda.device_count()       # identity on a full node; lets 2 ranks share one GPU in a dry run
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # AUM_BENCH_FORCE_DDP=1: the data-parallel path (RCCL process group, DistributedDataParallel reducer with bucket views, the gradient
    # exchange hook) at world size 1 -- a one-GPU box then runs everything of the N > 1 step except the wire
    force_ddp = os.environ.get("AUM_BENCH_FORCE_DDP", "0") == "1"
    if world == 1 and force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:       # a port the kernel hands out as free: two forced-DDP runs on one box must not collide
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_ddp:
        import torch.distributed as dist
        backend = os.environ.get("AUM_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm (xGMI within the node); gloo = dry run
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    aum_hip.get()      # fail loudly now if the HIP extension is missing

    torch.manual_seed(3949 + rank)
    n_class = 527
    model = build_aum(args.size, depth=args.depth, num_classes=n_class, bimamba_type="v1").to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-7, betas=(0.95, 0.999), eps=1e-8,
                           fused=True)                                         # TT:32-34
    net, homes = model, None
    if dist is not None:
        # bucket size: torch's default (25 MB, and a first bucket of 1 MB).  A custom cap drops the small first bucket, and the odd-sized
        # head.bias (527 floats) then leaves every later view of a 64 MB bucket 12 bytes off a 16-byte boundary: 54 gradients the
        # kernels cannot write in place (ssi.grad_home) and the fused Adam reads unaligned (profiles/r06_ddp_overhead.txt)
        bucket_mb = int(os.environ["AUM_BENCH_BUCKET_MB"]) if os.environ.get("AUM_BENCH_BUCKET_MB") else None
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index], gradient_as_bucket_view=True,
                                                        bucket_cap_mb=bucket_mb, broadcast_buffers=False)
        from aum.train import compress_gradients
        homes = compress_gradients(net, args.grad_compress)
        if os.environ.get("AUM_BENCH_NO_GRAD_HOMES") == "1":          # A/B: the reducer copies every gradient into its bucket, as torch does by default
            homes = None
    loss_fn = torch.nn.BCEWithLogitsLoss()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if args.no_frontend:
        x = torch.randn(args.batch, 1024, 128, device=dev, generator=g) * 0.5   # SURVEY 8d synthetic spectrograms
        spec = lambda: x
        fe = None
    else:   # 10 s / 16 kHz synthetic waveforms; waveform -> tokens (aum_frontend_tokens_fwd) runs inside the timed step (config 2)
        from aum.frontend import FbankTables, WaveInput, prepare_wave
        tabs = FbankTables(dev)
        wave = (torch.randn(args.batch, 160000, device=dev, generator=g) * 0.1).clamp_(-1, 1)
        spec = lambda: prepare_wave(wave, None, tabs)[0]          # mean removal; log-mel + patch embedding run inside the model
        fe = WaveInput(tabs, 1024)
    y = torch.zeros(args.batch, n_class, device=dev)
    y.scatter_(1, torch.randint(0, n_class, (args.batch, 2), device=dev, generator=g), 1.0)

    def step():
        xin = spec()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = net(xin, frontend=fe)
            loss = loss_fn(logits.float(), y)
        loss.backward()
        if homes is not None:
            homes.after_backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    # Per-kernel table: every launch of the LAST warm-up step is bracketed with HIP events (~340 launches, ~3 ms of host
    # time per step -- kept out of the timed region).  Timed region: only the dominant kernel's launches are bracketed,
    # which is what the roofline object needs.
    table = None
    for i in range(args.warmup):
        if i == args.warmup - 1:
            aum_hip.timer.reset()
            aum_hip.timer.only, aum_hip.timer.enabled = None, True
        step()
    if args.warmup > 0:
        table = aum_hip.timer.summary()
        tot_w = {k: v["avg_ms"] * v["launches"] for k, v in table.items()}
        aum_hip.timer.only = {max(tot_w, key=tot_w.get)}
    aum_hip.timer.reset()
    aum_hip.timer.enabled = True
    # ... every 5th of them (24 launches per step: 4 or 5 a step, every layer position over five steps): the two events of a bracket are two
    # barrier packets on the stream, 2 x 5.8 us of idle GPU per bracketed launch -- 0.28 ms per step when all 24 are bracketed
    # (profiles/r06_step_timeline.txt); `launches_timed` in the roofline object says how many the average is over
    aum_hip.timer.every = int(os.environ.get("AUM_BENCH_TIMER_EVERY", "5"))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    aum_hip.timer.enabled = False
    aum_hip.timer.every = 1
    # the same step without the optimizer (SURVEY 8d asks for both figures); a few extra steps, not part of `value`
    def step_no_opt():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            l_ = loss_fn(net(spec(), frontend=fe).float(), y)
        l_.backward()
        opt.zero_grad(set_to_none=True)
    n_extra = max(2, min(5, args.steps))
    step_no_opt()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_extra):
        step_no_opt()
    torch.cuda.synchronize()
    ms_no_opt = (time.perf_counter() - t1) / n_extra * 1e3
    rank_ms, n_buckets = None, None
    if dist is not None:
        # the slowest rank's time is the job's; every rank's own figure rides along so that a first multi-GPU run can be read from one line
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t_.item()) / args.steps * 1e3 for t_ in every]
        rank_ms = {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3), "per_rank": [round(v, 3) for v in per_rank]}
        elapsed = max(float(t_.item()) for t_ in every)
        try:
            n_buckets = len(net.reducer._get_zeros_like_grad_buckets()) if hasattr(net.reducer, "_get_zeros_like_grad_buckets") else None
        except Exception:
            n_buckets = None
        if n_buckets is None:
            grad_bytes = sum(p.numel() * p.element_size() for p in model.parameters())
            n_buckets = -(-grad_bytes // ((bucket_mb or 25) << 20))
    final_loss = float(loss.item())
    ktimes = aum_hip.timer.summary()

    if rank == 0:
        clips = world * args.batch * args.steps
        # dominant kernel = largest share of summed launch time among the hand-written kernels
        tot = {k: v["avg_ms"] * v["launches"] for k, v in ktimes.items()}
        dom = max(tot, key=tot.get)
        rec = ktimes[dom]                                   # launches of the timed region
        per_step = ({k: v["avg_ms"] * v["launches"] for k, v in table.items()} if table is not None
                    else {k: v / args.steps for k, v in tot.items()})
        alg = scan_alg_bytes(rec["meta"], "bwd" in dom) if dom.startswith("scan") else None
        roof = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
                "traffic": None, "avg_launch_ms": round(rec["avg_ms"], 4), "launches_timed": rec["launches"]}
        if alg is not None:
            roof["alg_bytes_per_launch"] = alg
            roof["achieved"] = round(alg / (rec["avg_ms"] * 1e-3) / 1e9, 1)
            roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBPS, 4)
            if "bwd" in dom:
                # SURVEY.md 8(d)'s own figure for the backward also counts `out` and the state checkpoints (s (9T + 2BNL) + 4 (2BNL) + ckpt =
                # 926.6 MB at the bench shape); `achieved` above uses the tensors this design's kernel has to touch (no `out`: the gate is
                # recomputed from out_pre).  Both, so that either definition can be read off the line.
                batch, dim, length, dstate, es, _ = rec["meta"]
                alg8d = alg + batch * dim * length * es + batch * dim * ((length + 2047) // 2048) * 2 * dstate * 4
                roof["alg_bytes_per_launch_survey_8d"] = alg8d
                roof["frac_survey_8d"] = round(alg8d / (rec["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        roof["traffic"], roof["traffic_provenance"] = pmc_record("pmc_traffic", dom)      # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE passes
        if alg is not None:
            roof["frac_of_measured_copy_rate"] = round(roof["achieved"] / HBM_COPY_GBPS, 4)      # 6.29 TB/s float4 copy (MI355X_MICROARCH.md)
            if isinstance(roof["traffic"], (int, float)) and roof["traffic"]:
                roof["traffic_ratio"] = round(roof["traffic"] / alg, 3)                           # HBM bytes moved / algorithmic bytes
                roof["actual_GBps"] = round(roof["traffic"] / (rec["avg_ms"] * 1e-3) / 1e9, 1)
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        if dom.startswith("scan") and rec.get("meta") is not None:
            fl_ms = scan_valu_floor_ms(rec["meta"], "bwd" in dom, 2 if "bidir" in dom else 1, n_cu)
            roof["valu_floor_ms"] = round(fl_ms, 4)
            roof["valu_floor_frac"] = round(fl_ms / rec["avg_ms"], 4)         # the attainable bar for this kernel: >= 0.6
        roof["valu"], roof["valu_provenance"] = pmc_record("valu_busy", dom)              # vector-ALU occupancy of the same kernel (SQ pass)
        if isinstance(roof.get("valu"), dict) and "valu_busy_frac" in roof["valu"]:
            roof["valu_frac"] = roof["valu"]["valu_busy_frac"]
        if dom.startswith("scan"):
            roof["note"] = ("VALU-bound kernel (16 states x 2 directions, one v_exp_f32 per state and step at a quarter of the fp32 rate, "
                            "packed fp32 FMAs for the rest): the vector ALU is busy for `valu_frac` of the launch (SQ_ACTIVE_INST_VALU), so the "
                            "fraction of the HBM roofline is bounded near 0.1-0.3 by arithmetic; `traffic` above the algorithmic bytes is the "
                            "state checkpoint (every 8 steps; pairs of states in the activations' 16-bit type) and the per-wave dB/dC partial rows; see DESIGN.md 4")
        # the forward kernel of the same scan and the largest library GEMM, against their own roofs
        fwd_roof = None
        fk = "scan_tm_fwd_bidir" if "scan_tm_fwd_bidir" in ktimes or (table is not None and "scan_tm_fwd_bidir" in table) else "scan_fwd_bidir"
        frec = ktimes.get(fk) or (table or {}).get(fk)
        if frec is not None and frec.get("meta") is not None:
            falg = scan_alg_bytes(frec["meta"], False)
            fwd_roof = {"bound": "hbm", "kernel": fk, "alg_bytes_per_launch": falg, "avg_launch_ms": round(frec["avg_ms"], 4),
                        "achieved": round(falg / (frec["avg_ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
            fwd_roof["frac"] = round(fwd_roof["achieved"] / HBM_PEAK_GBPS, 4)
            fwd_roof["traffic"], fwd_roof["traffic_provenance"] = pmc_record("pmc_traffic", fk)
            fwd_roof["frac_of_measured_copy_rate"] = round(fwd_roof["achieved"] / HBM_COPY_GBPS, 4)
            if isinstance(fwd_roof["traffic"], (int, float)) and fwd_roof["traffic"]:
                fwd_roof["traffic_ratio"] = round(fwd_roof["traffic"] / falg, 3)
                fwd_roof["actual_GBps"] = round(fwd_roof["traffic"] / (frec["avg_ms"] * 1e-3) / 1e9, 1)     # what the kernel really moves per second
            ffl = scan_valu_floor_ms(frec["meta"], False, 2 if "bidir" in fk else 1, n_cu)
            fwd_roof["valu_floor_ms"], fwd_roof["valu_floor_frac"] = round(ffl, 4), round(ffl / frec["avg_ms"], 4)
            fv, _ = pmc_record("valu_busy", fk)
            if isinstance(fv, dict):
                fwd_roof["valu_frac"] = fv.get("valu_busy_frac")
        gemm_roof = gemm_probe(dev, args.batch * 513, model.embed_dim, 2 * model.embed_dim) if world == 1 else None
        out = {
            "metric": "clips/sec/node AuM-Base 128x1024 fwd+bwd", "value": round(clips / elapsed, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"AuM-{args.size.capitalize()} (d_model={model.embed_dim}, {args.depth} Fo-Bi blocks, "
                                   f"d_state=16, {n_params / 1e6:.1f}M params) 128-mel x 1024-frame clips, L=513 tokens, "
                                   "fwd+bwd+Adam, bf16 autocast / fp32 master weights"
                                   + ("" if args.no_frontend else ", input = 160000-sample waveforms through the one-launch HIP frontend (log-mel + 16x16 patch embedding + position rows)"),
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}" + (" (DDP, RCCL all-reduce overlapped with backward)" if dist is not None else "")},
            "roofline": roof,
            "roofline_forward_kernel": fwd_roof,
            "roofline_gemm": gemm_roof,
            # whole step against the HBM roofline: SURVEY 8(d)'s algorithmic bytes of the 24 layers / measured step time
            "step_roofline": {"bound": "hbm", "alg_bytes_per_step": step_alg_bytes(args.batch, 513, model.embed_dim, args.depth),
                              "achieved": round(step_alg_bytes(args.batch, 513, model.embed_dim, args.depth) / (elapsed / args.steps) / 1e9, 1),
                              "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(step_alg_bytes(args.batch, 513, model.embed_dim, args.depth) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, 4)},
            "ms_per_step_without_optimizer": round(ms_no_opt, 3),
            "dist": {"world_size": world, "backend": (dist.get_backend() if dist is not None else None),
                     "rccl": bool(dist is not None and dist.get_backend() == "nccl"), "rank_ms_per_step": rank_ms,
                     "ddp_buckets": n_buckets, "bucket_cap_mb": (bucket_mb or 25) if dist is not None else None,
                     "grads_written_in_bucket": (ssi_mod.HOME_HITS[0] if dist is not None else None),
                     "grad_exchange_dtype": ("fp32" if args.grad_compress == "no" else args.grad_compress) if dist is not None else None},
            "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(per_step.items())},
            "final_loss": round(final_loss, 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            # north_star's CPU baseline is the reference's pure-PyTorch selective_scan_ref loop; the C oracle (the checker of the parity
            # tests, all operators of a block, OpenMP on every core) is the faster CPU statement and is reported beside it
            out["cpu_baseline"] = cpu_baseline_torch_ref()
            out["cpu_baseline_oracle"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
