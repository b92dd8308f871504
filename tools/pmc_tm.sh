# Issue / wait counters of the time-serial scan kernels (run through gpurun).  Counter passes use --kernel-trace only.
# usage: tools/pmc_tm.sh <kbench --only filter> <output tag>
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
FILTER=${1:-scan_tm}
TAG=${2:-tm}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python tools/kbench.py --only $FILTER > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python tools/kbench.py --only $FILTER > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "scan" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
         "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_INSTS_VMEM_RD"]
with open(out + "/summary.txt", "w") as fo:
    for k, c in sorted(acc.items()):
        m = {n: (sum(c[n]) / len(c[n]) if c[n] else 0.0) for n in names}
        g = m["GRBM_GUI_ACTIVE"] / 8.0
        line = [k, "calls=%d" % len(c["SQ_INSTS_VALU"])] + ["%s=%.4g" % (n, m[n]) for n in names]
        if g and m["SQ_INSTS_VALU"]:
            line.append("valu_busy=%.1f%%" % (100.0 * 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * g)))
            line.append("cyc_per_valu=%.2f" % (4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]))
            line.append("kernel_cycles=%.0f" % g)
        if m["SQ_WAVE_CYCLES"]:
            line.append("wait_any=%.1f%% wait_inst=%.1f%% active_any=%.1f%%" % (100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
                                                                           100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
        fo.write("  ".join(line) + "\n")
print(open(out + "/summary.txt").read())
PY
