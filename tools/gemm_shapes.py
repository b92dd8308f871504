#!/usr/bin/env python3
"""Which library GEMMs one training step of the bench configuration still issues (shapes and counts): torch.profiler with record_shapes
over one step after warm-up.  python tools/gemm_shapes.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
from aum import tunable  # noqa: E402
tunable.enable(0)
import torch  # noqa: E402
from aum.model import build_aum  # noqa: E402
from aum.frontend import FbankTables, wav2fbank  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
model = build_aum("base", depth=24, num_classes=527, bimamba_type="v1", spectrogram_size=(128, 1024)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
tabs = FbankTables(dev)
wave = (torch.randn(64, 160000, device=dev) * 0.1).clamp_(-1, 1)
y = torch.zeros(64, 527, device=dev)
loss_fn = torch.nn.BCEWithLogitsLoss()


def step():
    x = wav2fbank(wave, tabs, target_length=1024)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = loss_fn(model(x).float(), y)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::linear", "aten::addmm_", "aten::baddbmm", "aten::mv"):
        if e.name in ("aten::matmul", "aten::linear"):
            continue
        k = (e.name, str(e.input_shapes))
        agg[k][0] += 1
        agg[k][1] += e.device_time_total
for (name, shapes), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {name:14s} {shapes:70s} {t / max(n, 1):8.1f} us each")
