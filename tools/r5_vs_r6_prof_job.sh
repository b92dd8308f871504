#!/bin/bash
# kernel tables of the round-5 tree (_r5tree/, see tools/r5_vs_r6_job.sh) and of this tree from rocprofv3 runs on ONE box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R6=$PWD
cd /tmp && export TMPDIR=/tmp
for t in r5 r6; do
  if [ $t = r5 ]; then cd $R6/_r5tree; else cd $R6; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rr_$t -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /tmp/b_$t.json 2>/dev/null
  tail -1 /tmp/b_$t.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"
  db=$(find /tmp/rr_$t -name "*.db" | head -1)
  python $R6/tools/rocpd_stats.py $db $R6/gpurun_out/ks_$t.txt > /dev/null
done
