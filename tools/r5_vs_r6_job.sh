#!/bin/bash
# same-box A/B of the round-5 tree (97f33ba, extracted and built under _r5tree/ by `git archive 97f33ba | tar -x -C _r5tree` +
# `python _r5tree/audio-mamba-aum_amd/csrc/build.py`) against this tree: alternating plain bench.py runs, ms per step and clips/s
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R6=$PWD
for i in 1 2 3; do
  for t in r5 r6; do
    if [ $t = r5 ]; then cd $R6/_r5tree; else cd $R6; fi
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"
  done
done
