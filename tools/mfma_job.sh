# MFMA-busy counters of the GEMM and projection kernels (run through gpurun); counter pass = --kernel-trace only.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/run -o pmc -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_mfma/run.log 2>&1
find gpurun_out/pmc_mfma/run -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_mfma/mfma.csv \;
python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("gpurun_out/pmc_mfma/mfma.csv")):
    acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-72s %8s %14s %14s %8s" % ("kernel", "calls", "MFMA_BUSY", "GUI_ACTIVE", "MFMA%"))
rows = []
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    g = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(c.get("GRBM_GUI_ACTIVE", [0])))
    rows.append((m, k, len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), g))
for m, k, n, g in sorted(rows, reverse=True)[:16]:
    # MFMA_BUSY is summed over the 1024 SIMDs of the chip, GRBM_GUI_ACTIVE over its 8 XCDs (checked against the kernel-trace
    # durations: in_proj forward 153 us = 367k cycles at 2.4 GHz = GUI_ACTIVE / 8)
    print("%-72s %8d %14.0f %14.0f %7.1f%%" % (k, n, m, g, 100.0 * m / (1024.0 * g / 8.0) if g else 0.0))
PY
