set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/prof_r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2/run -o bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_r2/bench.json 2> gpurun_out/prof_r2/bench.err
tail -c 600 gpurun_out/prof_r2/bench.json
db=$(find gpurun_out/prof_r2/run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/prof_r2/kernel_stats.txt | head -75
python - <<'PY'
import sqlite3, glob, re
db = glob.glob('gpurun_out/prof_r2/run/**/*.db', recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {namec}, count(*), sum(end-start) from kernels group by {namec} order by 3 desc").fetchall()
with open('gpurun_out/prof_r2/kernel_stats_all.txt', 'w') as f:
    for n, c, s in rows:
        f.write(f"{s/1e6:10.3f} ms {c:7d}  {n[:300]}\n")
PY
find gpurun_out/prof_r2/run -name "*.db" -delete
