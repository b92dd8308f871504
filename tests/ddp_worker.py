"""Worker for test_ddp_gloo.py: 2 ranks, gloo, CPU.  Data-parallel semantics of the training path:
disjoint minibatch shards, DDP mean all-reduce of the gradients during backward, identical parameters after the step.
The kernels run through the tests-only lane-array library (no GPU here); the collective path is torch.distributed."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-mamba-aum_amd"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import aum_hip  # noqa: E402
import build_emu  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    from aum.model import AudioMamba
    torch.manual_seed(0)                                   # same init on every rank
    model = AudioMamba(spectrogram_size=(128, 128), depth=2, embed_dim=32, num_classes=5)
    ref = AudioMamba(spectrogram_size=(128, 128), depth=2, embed_dim=32, num_classes=5)
    ref.load_state_dict(model.state_dict())
    ddp = torch.nn.parallel.DistributedDataParallel(model, gradient_as_bucket_view=True, bucket_cap_mb=1)
    if os.environ.get("AUM_TEST_COMPRESS"):               # the optional 16-bit gradient exchange of aum.train / bench.py
        from aum.train import compress_gradients
        compress_gradients(ddp, os.environ["AUM_TEST_COMPRESS"])
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(2 * world, 128, 128, generator=g) * 0.5
    y_all = (torch.rand(2 * world, 5, generator=g) > 0.7).float()
    xs, ys = x_all[rank * 2:(rank + 1) * 2], y_all[rank * 2:(rank + 1) * 2]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(ddp(xs), ys)
    loss.backward()
    # reference: single-process gradient over the GLOBAL batch (mean loss) == mean of the per-shard gradients
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(ref(x_all), y_all)
    loss_ref.backward()
    worst = 0.0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        scale = q.grad.abs().max().item() + 1e-12
        worst = max(worst, (p.grad - q.grad).abs().max().item() / scale)
    # every rank holds the same reduced gradient
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.95, 0.999), eps=1e-8, weight_decay=5e-7)
    opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    same_w = all(torch.equal(ws[0], t) for t in ws)
    if rank == 0:
        print(f"DDP_RESULT worst_rel_grad_err={worst:.3e} grads_identical={same} weights_identical={same_w}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
