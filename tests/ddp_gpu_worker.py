"""Worker for tests/test_gpu_ddp.py: ONE rank, backend "nccl" (= RCCL on ROCm) on the MI355X.  A one-GPU box has no wire, but everything
else of the N > 1 training step runs: RCCL communicator init, the DistributedDataParallel reducer, gradients written into bucket VIEWS
by this package's custom autograd Functions, the gradient-exchange hook (plain / 16-bit, behind ssi.ddp_join_streams_hook), Bi-Bi's two
backward streams under the reducer, the fused Adam step on bucket-view gradients.  Checked against the same model without the wrapper:
gradients and parameters bit-equal (TT:39, TT:168)."""
import copy
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-mamba-aum_amd")):
    sys.path.insert(0, p)
import aum_hip  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    lib = aum_hip.get()
    assert not lib.host
    from aum.model import build_aum
    from aum.train import compress_gradients
    from mamba_ssm.ops import selective_scan_interface as ssi
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    # a collective that really goes through RCCL
    t = torch.arange(8, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    out["all_reduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))
    B, n_class = 4, 527
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, 1024, 128, device=dev, generator=g) * 0.5
    y = torch.zeros(B, n_class, device=dev)
    y.scatter_(1, torch.randint(0, n_class, (B, 2), device=dev, generator=g), 1.0)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    assert not ssi.v2_two_streams(), "a process group without the joining hook: Bi-Bi stays on one stream"
    ssi._TM_MIN_WAVES = 1          # the token-major block (the bench's) at this small batch: its kernels are the ones that write gradients in place
    for name, bim, comp in (("v1_fp32", "v1", "no"), ("v1_bf16", "v1", "bf16"), ("v2_fp32", "v2", "no")):
        torch.manual_seed(11)
        model = build_aum("base", depth=2, num_classes=n_class, bimamba_type=bim).to(dev)
        ref = copy.deepcopy(model)
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True,
                                                        broadcast_buffers=False)
        homes = compress_gradients(ddp, comp)  # registers the exchange hook (Bi-Bi: ssi.ddp_join_streams_hook(...): from here on it may use its side stream)
        opts = [torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=5e-7, betas=(0.95, 0.999), eps=1e-8, fused=True) for m in (model, ref)]
        # the permission is per wrapped model: this wrapper's Bi-Bi blocks carry it, a block outside any hooked wrapper does not
        v2_blocks = [m_ for m_ in model.modules() if getattr(m_, "bimamba_type", None) == "v2"]
        rec = {"two_streams": bool(v2_blocks) and all(ssi.v2_two_streams(module=m_) for m_ in v2_blocks), "steps": []}
        assert not ssi.v2_two_streams(), "a block that no hooked wrapper marked stays in line under a process group"
        hits0 = ssi.HOME_HITS[0]
        for step in range(4):
            losses = []
            for net, streams in ((ddp, True), (ref, False)):
                ssi._V2_STREAMS = streams      # the un-wrapped comparison model runs Bi-Bi in line: two streams vs one, bit for bit
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = loss_fn(net(x).float(), y)
                loss.backward()
                if net is ddp:
                    adopted = homes.after_backward()       # from the next step on the kernels write into the reducer's buckets
                losses.append(float(loss))
            ssi._V2_STREAMS = True
            torch.cuda.synchronize()
            worst, equal, n = 0.0, True, 0
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                if q.grad is None:
                    assert p.grad is None, k
                    continue
                want = q.grad if comp == "no" else q.grad.to(torch.bfloat16).float()     # world size 1: the 16-bit hook is cast, reduce, cast back
                equal = equal and torch.equal(p.grad, want)
                worst = max(worst, float((p.grad - want).abs().max()) / (float(want.abs().max()) + 1e-30))
                n += 1
            views = sum(1 for p in model.parameters() if p.grad is not None and p.grad._is_view())
            # gradients the kernels wrote straight into the reducer's bucket (ssi.grad_home): none in step 0, the blocks' parameters afterwards
            in_home, hits0 = ssi.HOME_HITS[0] - hits0, ssi.HOME_HITS[0]
            for o in opts:
                o.step()
                o.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            same_w = all(torch.equal(p, q) for p, q in zip(model.parameters(), ref.parameters()))
            rec["steps"].append({"loss": losses, "grads_equal": bool(equal), "worst_rel": worst, "n_grads": n, "bucket_views": views, "in_home": in_home, "adopted": adopted,
                                 "params_equal": bool(same_w), "finite": bool(all(torch.isfinite(p).all() for p in model.parameters()))})
            if comp != "no":
                break              # after a rounded exchange the two models differ by construction
        out[name] = rec
        del ddp, model, ref, opts
    print("DDP_GPU_RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
