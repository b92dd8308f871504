#!/bin/bash
# Where does DistributedDataParallel's time go at world size 1?  Same box: alternating un-profiled bench.py runs (plain,
# forced DDP with the fp32 / bf16 exchange), then one rocprofv3 kernel trace of each and the kernel-by-kernel difference
# (tools/ddp_overhead_diff.py) -> gpurun_out/ddp_overhead.txt.   gpurun -- 'bash tools/ddp_overhead_job.sh [rounds] [extra specs ...]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD/audio-mamba-aum_amd:${PYTHONPATH:-}
mkdir -p gpurun_out
OUT=gpurun_out/ddp_overhead.txt
: > $OUT
rounds=${1:-2}; shift || true
one() {   # label, env words ...
    local label=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      python bench.py --steps 12 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-} 2> gpurun_out/b.err | grep '^{"metric"' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', d['ms_per_step'], d['value'], 'no-opt', d['ms_per_step_without_optimizer'], {k: v for k, v in d['dist'].items() if k in ('ddp_buckets', 'bucket_cap_mb', 'grad_exchange_dtype')})" ) | tee -a $OUT
}
for ((i = 0; i < rounds; ++i)); do
    one plain
    one ddp_fp32 AUM_BENCH_FORCE_DDP=1
    BENCH_ARGS="--grad-compress bf16" one ddp_bf16 AUM_BENCH_FORCE_DDP=1
    for s in "$@"; do one "$s" AUM_BENCH_FORCE_DDP=1 ${s//,/ }; done
done
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
trace() {  # label, bench args, env words
    local label=$1 bargs=$2; shift 2
    rm -rf /tmp/prof_$label
    ( for kv in "$@"; do export "$kv"; done
      timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$label -o bench -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline $bargs > gpurun_out/prof_$label.json 2> gpurun_out/prof_$label.err )
    tail -c 200 gpurun_out/prof_$label.json; echo
}
trace plain ""
trace ddp_fp32 "" AUM_BENCH_FORCE_DDP=1
trace ddp_bf16 "--grad-compress bf16" AUM_BENCH_FORCE_DDP=1
db() { find /tmp/prof_$1 -name '*.db' | head -1; }
{ echo "=== plain vs forced DDP, fp32 exchange"; python tools/ddp_overhead_diff.py "$(db plain)" "$(db ddp_fp32)" 5 6
  echo "=== plain vs forced DDP, bf16 exchange"; python tools/ddp_overhead_diff.py "$(db plain)" "$(db ddp_bf16)" 5 6; } | tee -a $OUT
