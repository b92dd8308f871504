# wait-state / occupancy counters of the scan row kernels (separate --pmc passes, --kernel-trace only)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
OUT=gpurun_out/${1:-pmc_r2}
ONLY=${2:-scan_ck}
mkdir -p $OUT
i=0
for grp in "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/run$i -o pmc -- python tools/kbench.py --only $ONLY > $OUT/run$i.log 2>&1
  find $OUT/run$i -name "*counter_collection.csv" -exec cp {} $OUT/pass$i.csv \;
  rm -rf $OUT/run$i
done
python - $OUT <<'PY'
import csv, collections, glob, sys
out_dir = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out_dir + "/pass*.csv")):
    for row in csv.DictReader(open(f)):
        if "scan" in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:52]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out_dir + "/summary.txt", "w") as out:
    for k, c in sorted(acc.items()):
        out.write(k + "\n")
        for n, v in sorted(c.items()):
            out.write("    %-26s %18.0f  (%d launches)\n" % (n, sum(v) / len(v), len(v)))
print(open(out_dir + "/summary.txt").read())
PY
rm -f $OUT/pass*.csv
