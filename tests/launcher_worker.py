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
    aum_hip._product = aum_hip.Lib(build_emu.build(), host=True)
    import json
    from aum import train
    train.main(sys.argv[1:])
    exp = sys.argv[sys.argv.index("--exp-dir") + 1]
    with open(os.path.join(exp, f"host_syncs_rank{os.environ.get('RANK', '0')}.json"), "w") as f:
        json.dump(train.HOST_SYNCS, f)
