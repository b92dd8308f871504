# HBM traffic and kernel durations of the skinny-gradient kernels (tools/skinny_probe.py launches): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
OUT=/tmp/skinny_pmc; rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/skinny_probe.py > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python tools/skinny_probe.py > $OUT/$c.log 2>&1
done
python - <<'PY' | tee gpurun_out/skinny_pmc.txt
import csv, glob, collections
st = glob.glob('/tmp/skinny_pmc/trace/**/*kernel_stats.csv', recursive=True)[0]
dur = {r['Name']: (float(r['AverageNs']), int(r['Calls'])) for r in csv.DictReader(open(st))}
def pmc(c):
    f = glob.glob(f'/tmp/skinny_pmc/{c}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = pmc('FETCH_SIZE'), pmc('WRITE_SIZE')
print("kernel | avg us | calls | read MB (FETCH_SIZE x2 KiB, gfx950 correction) | write MB | GB/s")
for k, (ns, n) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if not any(t in k for t in ('xdt', 'wgrad', 'Cijk', 'sum_rows')): continue
    rd, w = 2 * 1024 * fe.get(k, 0) / 1e6, 1024 * wr.get(k, 0) / 1e6
    print(f"{k[:90]:90s} {ns / 1e3:8.1f} {n:5d} {rd:9.1f} {w:9.1f} {(rd + w) * 1e6 / ns / 1e3:8.0f}" if ns else k)
PY
