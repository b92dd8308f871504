"""Worker for test_train_launcher.py::test_launcher_two_ranks_gloo: runs aum.train.main under torch.distributed.run
with the tests-only lane-array library where libaum_hip.so would be (no GPU here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-mamba-aum_amd"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import aum_hip  # noqa: E402
import build_emu  # noqa: E402

if __name__ == "__main__":
    if os.environ.get("AUM_TEST_PRODUCT_LIB") != "1":      # tests/test_gpu_ddp.py runs the same worker on the product library
        aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    import json
    from aum import train
    # tests only: make the loss of rank AUM_TEST_NAN_RANK non-finite at the listed training steps (test_launcher_two_ranks_gloo_nan_steps)
    nan_steps = {int(x) for x in os.environ.get("AUM_TEST_NAN_STEPS", "").split(",") if x}
    if nan_steps and os.environ.get("RANK", "0") == os.environ.get("AUM_TEST_NAN_RANK", "1"):
        real_loss, calls = train._loss, {"n": 0}

        def poisoned(loss_fn, out, labels):
            import torch
            loss = real_loss(loss_fn, out, labels)
            if torch.is_grad_enabled():                      # training steps only (validation runs under no_grad)
                if calls["n"] in nan_steps:
                    loss = loss * float("nan")
                calls["n"] += 1
            return loss
        train._loss = poisoned
    train.main(sys.argv[1:])
    exp = sys.argv[sys.argv.index("--exp-dir") + 1]
    with open(os.path.join(exp, f"host_syncs_rank{os.environ.get('RANK', '0')}.json"), "w") as f:
        json.dump(train.HOST_SYNCS, f)
