# VALU-busy counters of the scan kernels (run through gpurun); counter pass = --kernel-trace only.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/pmc_valu
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_valu/run -o pmc -- python tools/kbench.py --only scan_fwd,scan_bwd > gpurun_out/pmc_valu/run.log 2>&1
find gpurun_out/pmc_valu/run -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_valu/valu.csv \;
python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("gpurun_out/pmc_valu/valu.csv")):
    if "scan" in row["Kernel_Name"]:
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs")
print("%-62s %6s %16s %14s %14s %10s %10s" % ("kernel", "calls", "ACTIVE_INST_VALU", "INSTS_VALU", "GUI_ACTIVE", "cyc/inst", "VALU busy"))
for k, c in sorted(acc.items()):
    m = lambda n: sum(c[n]) / max(1, len(c[n]))
    a, i, g = m("SQ_ACTIVE_INST_VALU"), m("SQ_INSTS_VALU"), m("GRBM_GUI_ACTIVE")
    print("%-62s %6d %16.0f %14.0f %14.0f %10.2f %9.1f%%" % (k, len(c["SQ_INSTS_VALU"]), a, i, g, 4.0 * a / i if i else 0, 100.0 * 4.0 * a / (1024.0 * g / 8.0) if g else 0))
PY
