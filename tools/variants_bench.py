#!/usr/bin/env python3
"""Throughput of the configurations around the headline one (NOT the bench line): AuM sizes, block types, inference.
Same step as bench.py (bf16 autocast, BCE, fused Adam, GPU log-mel frontend), per-GPU batch 64 unless noted."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import torch  # noqa: E402
from aum.model import build_aum  # noqa: E402
from aum.frontend import FbankTables, wav2fbank  # noqa: E402


def run(size, btype, train, batch=64, steps=6, warm=3, frames=1024, ddp=None):
    """ddp: None, or the gradient-exchange kind ("no" / "bf16") of a DistributedDataParallel wrapper over a world-size-1 RCCL group (the
    data-parallel step on one GPU: reducer, bucket views, the stream-joining exchange hook -- everything but the wire)"""
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = build_aum(size, depth=24, num_classes=527, bimamba_type=btype, spectrogram_size=(128, frames)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    net = model
    if ddp is not None:
        import torch.distributed as dist
        from aum.train import compress_gradients
        if not dist.is_initialized():
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29547"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
            dist.init_process_group("nccl", device_id=dev)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True, broadcast_buffers=False)
        compress_gradients(net, ddp)
    tabs = FbankTables(dev)
    wave = (torch.randn(batch, 400 + (frames - 1) * 160 if frames != 1024 else 160000, device=dev) * 0.1).clamp_(-1, 1)
    y = torch.zeros(batch, 527, device=dev)
    y[:, :2] = 1
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def step():
        x = wav2fbank(wave, tabs, target_length=frames)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if train:
                loss = loss_fn(net(x).float(), y)
            else:
                with torch.no_grad():
                    return model(x)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    model.train(train)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r = {"size": size, "block": {"v1": "Fo-Bi", "v2": "Bi-Bi", "none": "Fo-Fo"}[btype], "mode": "train" if train else "inference",
         "frames": frames, "tokens": model.num_patches + 1, "batch": batch, "ddp": ddp, "ms_per_step": round(dt * 1e3, 2), "clips_per_s": round(batch / dt, 1),
         "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    print(json.dumps(r), flush=True)
    del model, opt, net
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return r


if __name__ == "__main__":
    # `--only long` runs just the long-form configuration (BASELINE config 5: B = 8, 128 x 8192 frames, L = 4097)
    long_form = lambda: run("base", "v1", True, batch=8, steps=4, warm=2, frames=8192)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only == "long":
        out, name = [long_form()], "variants_bench_long.json"
    elif only == "bibi":      # the Bi-Bi block (and Fo-Fo beside it); AUM_DEBUG=1 AUM_TM_MIN_WAVES=1000000000 keeps the channel-major block for A/B
        out, name = [run("base", "v2", True), run("base", "none", True)], "variants_bench_bibi.json"
    elif only == "bibi_ddp":  # Bi-Bi under DistributedDataParallel (RCCL group of one): two backward streams joined by the exchange hook vs in line
        import mamba_ssm.ops.selective_scan_interface as ssi
        out = [run("base", "v2", True), run("base", "v2", True, ddp="no")]
        ssi._V2_STREAMS = False
        out.append(run("base", "v2", True, ddp="no"))
        out[-1]["block"] = "Bi-Bi (one stream)"
        ssi._V2_STREAMS = True
        out.append(run("base", "v1", True, ddp="bf16"))
        name = "variants_bench_bibi_ddp.json"
    else:
        out = [run("base", "v1", True), run("base", "v1", False), run("base", "v2", True), run("base", "none", True),
               run("small", "v1", True), run("small", "v1", False), run("tiny", "v1", True), run("base", "v1", True, batch=256, steps=3, warm=2), long_form()]
        name = "variants_bench.json"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
