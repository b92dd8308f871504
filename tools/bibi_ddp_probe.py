#!/usr/bin/env python3
"""GPU: Bi-Bi step time with one / two streams, alone and under DistributedDataParallel (world-size-1 RCCL group), depth-6 AuM-Base."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from aum.model import build_aum  # noqa: E402
import mamba_ssm.ops.selective_scan_interface as ssi  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29549"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)


def run(tag, ddp, streams, join):
    torch.manual_seed(0)
    model = build_aum("base", depth=6, num_classes=527, bimamba_type="v2").to(dev)
    net = model
    ssi._V2_STREAMS = streams
    if ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True, broadcast_buffers=False)
        if join:
            net.register_comm_hook(None, ssi.ddp_join_streams_hook())
        else:
            ssi._DDP_STREAM_JOIN = True          # (timing only: two streams without the join)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    x = torch.randn(64, 1024, 128, device=dev) * 0.5
    y = torch.zeros(64, 527, device=dev)
    y[:, :2] = 1
    lf = torch.nn.BCEWithLogitsLoss()
    calls = {"two": 0, "one": 0}
    real = ssi.v2_two_streams
    ssi.v2_two_streams = lambda *a: (calls.__setitem__("two" if real(*a) else "one", calls["two" if real(*a) else "one"] + 1), real(*a))[1]

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = lf(net(x).float(), y)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    ssi.v2_two_streams = real
    print(f"{tag:34s} {(time.perf_counter() - t0) / 6 * 1e3:7.2f} ms/step   v2_two_streams said yes {calls['two']} / no {calls['one']} times", flush=True)
    del net, model, opt


run("alone, two streams", False, True, False)
run("alone, one stream", False, False, False)
dist.init_process_group("nccl", device_id=dev)
run("DDP, two streams + join hook", True, True, True)
run("DDP, one stream", True, False, True)
run("DDP, two streams, NO join (unsafe)", True, True, False)
dist.destroy_process_group()
