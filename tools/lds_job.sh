# LDS / wait-state counters of the scan kernels (run through gpurun); counter passes = --kernel-trace only, one group each.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/pmc_lds
rocprofv3 --list-avail > gpurun_out/pmc_lds/list_avail.txt 2>&1
i=0
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_lds/run$i -o pmc -- python tools/kbench.py --only scan_fwd,scan_bwd > gpurun_out/pmc_lds/run$i.log 2>&1
  find gpurun_out/pmc_lds/run$i -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_lds/pass$i.csv \;
  rm -rf gpurun_out/pmc_lds/run$i
done
python - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_lds/pass*.csv")):
    for row in csv.DictReader(open(f)):
        if "scan" in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:52]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/pmc_lds/summary.txt", "w") as out:
    for k, c in sorted(acc.items()):
        out.write(k + "\n")
        for n, v in sorted(c.items()):
            out.write("    %-26s %18.0f  (%d launches)\n" % (n, sum(v) / len(v), len(v)))
print(open("gpurun_out/pmc_lds/summary.txt").read())
PY
