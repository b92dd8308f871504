"""SURVEY 8 a16 / 8(e) on the device without a second GPU: a world-size-1 RCCL process group runs the
DistributedDataParallel reducer, the bucket views, the gradient-exchange hooks and Bi-Bi's two backward streams against this package's
custom autograd Functions; the launcher's data-parallel loop (aum.train with AUM_FORCE_DDP=1) incl. the MIN-reduced skip-step branch;
bench.py's forced-DDP mode.  The two-rank semantics (disjoint shards, mean of the shard gradients) are tests/test_ddp_gloo.py's."""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_train_launcher import toy_dataset  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _env(**kw):
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)


def test_ddp_world1_rccl_reducer_bucket_views_hooks_two_streams():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"DDP_GPU_RESULT (.*)", r.stdout)
    assert m, r.stdout[-2000:]
    res = json.loads(m.group(1))
    assert res["backend"] == "nccl" and res["world"] == 1 and res["all_reduce_ok"]
    for name in ("v1_fp32", "v2_fp32"):
        steps = res[name]["steps"]
        assert len(steps) == 4
        for i, s in enumerate(steps):      # reducer + bucket views + hook leave exactly the un-wrapped model's gradients and parameters
            assert s["grads_equal"] and s["params_equal"] and s["finite"] and s["n_grads"] > 20, (name, s)
            # every gradient sits in a bucket: as the view the reducer made of it, or written there by the kernel that finished it (those the
            # accumulator keeps as plain aliases of the bucket's storage)
            assert s["bucket_views"] + s["in_home"] >= s["n_grads"] and (i > 1 or s["bucket_views"] == s["n_grads"]), (name, s)
            assert s["loss"][0] == s["loss"][1]
            # round 6: from the second step on the blocks' kernels write their parameter gradients straight into the reducer's buckets
            # (ssi.grad_home): 2 blocks x 11 parameters (+ the final norm's weight) -- and the gradients are still the un-wrapped model's, bit for bit
            assert (s["in_home"] == 0) if i == 0 else (s["in_home"] >= 22), (name, i, s)
    assert res["v2_fp32"]["two_streams"], "Bi-Bi's side stream stays on under DDP once the joining hook is registered"
    s = res["v1_bf16"]["steps"][0]
    assert s["grads_equal"] and s["finite"], s           # == the un-wrapped gradient rounded to bf16, bit for bit


def test_launcher_forced_ddp_rccl_nan_steps(toy_dataset, tmp_path):  # noqa: F811
    """aum.train.main on the GPU with AUM_FORCE_DDP=1 (RCCL group of one), --if_nan2num False --if_continue_inf True and a non-finite
    loss at steps 0 and 2: the finite flag goes through dist.all_reduce(MIN) on the device, the steps are skipped without a backward
    (legal for the reducer: static_graph is off on this path), the run finishes with finite weights and metrics (TT:153-164)."""
    exp = str(tmp_path / "exp_ddp")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "launcher_worker.py"),
           "--model_type", "tiny", "--depth", "1", "--n_class", "4", "--label-csv", str(toy_dataset / "labels.csv"),
           "--data-train", str(toy_dataset / "train.json"), "--data-val", str(toy_dataset / "val.json"),
           "--audio_length", "64", "--num-workers", "0", "-b", "2", "--mixed_precision", "bf16", "--exp-dir", exp,
           "--n-epochs", "1", "--metrics", "acc", "--loss", "CE", "--if_nan2num", "False", "--if_continue_inf", "True"]
    env = _env(AUM_FORCE_DDP="1", AUM_TEST_PRODUCT_LIB="1", AUM_TEST_NAN_STEPS="0,2", AUM_TEST_NAN_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("Loss is not finite on some rank, continuing training") == 2, r.stdout[-2000:]
    assert np.isfinite(np.loadtxt(exp + "/result.csv", delimiter=",").reshape(1, 8)).all()
    sd = torch.load(exp + "/models/best_audio_model.pth")
    assert all(k.startswith("module.") for k in sd)                          # the DistributedDataParallel wrapper was there
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    hs = json.load(open(exp + "/host_syncs_rank0.json"))
    assert hs["steps"] == 3, hs                                             # 10 clips / batch 2 = 5 steps, two of them skipped


def test_bench_forced_ddp_reports_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--depth", "2", "--batch", "8",
                        "--no-cpu-baseline", "--grad-compress", "bf16"], capture_output=True, text=True, timeout=900,
                       env=_env(AUM_BENCH_FORCE_DDP="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])      # (RCCL prints its library path at shutdown)
    assert line["dist"]["rccl"] is True and line["dist"]["world_size"] == 1 and line["dist"]["grad_exchange_dtype"] == "bf16"
    assert line["n_gpus"] == 1 and np.isfinite(line["final_loss"]) and line["dist"]["ddp_buckets"] >= 1
