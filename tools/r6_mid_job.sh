#!/bin/bash
# round 6, mid-session: the default bench under rocprofv3 --kernel-trace --stats (kernel table of the current tree) + a plain bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/mid
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/mid_run -o bench -- python bench.py > gpurun_out/mid/bench.json 2> gpurun_out/mid/bench.err
tail -c 600 gpurun_out/mid/bench.json
db=$(find /tmp/mid_run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/mid/kernel_stats.txt | head -60
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/mid/bench_plain.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/mid/bench_plain.json'));print('plain',d['ms_per_step'],d['value'])"
