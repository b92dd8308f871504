"""hipBLASLt/rocBLAS solution selection for the projection GEMMs (PyTorch TunableOp).

The default heuristic picks poor kernels for this model's GEMM shapes on gfx950 (in_proj 268 us vs 134 us tuned,
x_proj 145 vs 29 us: profiles/r01_v2_bench_kernel_stats.txt).  `enable()` must run BEFORE `import torch`-side GEMMs are
issued (env vars are read lazily at first use): it points TunableOp at the solutions recorded on an MI355X
(aum/tunableop_gfx950.csv, one copy per local rank because TunableOp appends the device ordinal to the file name) and
leaves online tuning ON, so a shape that is missing -- or a library version whose validators reject the file -- is tuned
during the first (warm-up) step instead of silently running the slow default.
"""
import atexit
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))


def enable(local_rank=0, tuning=True):
    if os.environ.get("AUM_NO_TUNABLEOP") == "1":
        return None
    src = os.environ.get("AUM_TUNABLEOP_CSV", os.path.join(_HERE, "tunableop_gfx950.csv"))
    d = os.path.join(tempfile.gettempdir(), f"aum_tunableop_{os.getuid()}_{os.getpid()}")
    os.makedirs(d, exist_ok=True)
    atexit.register(shutil.rmtree, d, ignore_errors=True)       # per-process scratch copy of the solution file
    if os.path.exists(src):
        # ordinal = local rank on a full node, 0 when the launcher masks each rank to one visible device
        for ordinal in {0, local_rank}:
            shutil.copyfile(src, os.path.join(d, f"results{ordinal}.csv"))
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1" if tuning else "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(d, "results.csv"))
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "10")
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    dump = os.environ.get("AUM_TUNABLEOP_DUMP")
    if dump:        # tools/tune_job.sh: everything this process looked up or tuned, written out for merging into the recorded file
        def _dump():
            import torch.cuda.tunable as t
            with open(dump, "w") as f:          # the layout of TunableOp's own result files
                for v in t.get_validators():
                    f.write("Validator," + ",".join(str(x) for x in v) + "\n")
                for r in t.get_results():
                    f.write(",".join(str(x) for x in r) + "\n")
        atexit.register(_dump)          # (registered after the scratch directory's removal: runs before it)
    return d
