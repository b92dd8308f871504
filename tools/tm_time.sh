# kernel durations of tools/tm_time.py from a kernel trace (run through gpurun).  usage: tools/tm_time.sh [variant ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
for v in ${@:-default}; do
  rm -rf gpurun_out/tm_time_$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tm_time_$v -o t -- python tools/tm_time.py $v > gpurun_out/tm_time_$v.log 2>&1
  python - gpurun_out/tm_time_$v $v <<'PY'
import csv, collections, glob, sys
csv.field_size_limit(10**9)
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_scan" in n or "k_convt" in n or "k_sum_rows" in n:
            acc[n[:110]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print("%-8s %-112s calls %3d  median %8.1f us  min %8.1f" % (sys.argv[2], k, len(v), v[len(v) // 2], v[0]))
PY
done
