#!/bin/bash
# Kernel-by-kernel difference of the bench step under two configurations (same box, one rocprofv3 kernel trace each):
#   gpurun -- 'bash tools/step_diff_job.sh <specA> <specB> [bench args]'     spec = "-" or ENV=V,ENV=V,... (AUM_DEBUG=1 implied), lib:<variant>
# -> gpurun_out/step_diff.txt (tools/ddp_overhead_diff.py: launches and ms per step per kernel, busy / idle time, the largest gaps)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD/audio-mamba-aum_amd:${PYTHONPATH:-}
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
mkdir -p gpurun_out
A=$1; B=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
trace() {
    local label=$1 spec=$2; shift 2
    rm -rf /tmp/prof_$label
    ( if [ "$spec" != "-" ]; then export AUM_DEBUG=1; IFS=',' read -ra items <<< "$spec"; for it in "${items[@]}"; do case $it in lib:*) export AUM_HIP_LIB=$V/libaum_hip_${it#lib:}.so ;; *) export "$it" ;; esac; done; fi
      timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$label -o bench -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/prof_$label.json 2> gpurun_out/prof_$label.err )
    grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$label.json | head -1
}
trace a "$A" "$@"
trace b "$B" "$@"
db() { find /tmp/prof_$1 -name '*.db' | head -1; }
{ echo "=== A = $A   B = $B"; python tools/ddp_overhead_diff.py "$(db a)" "$(db b)" 5 6; } | tee gpurun_out/step_diff.txt | cut -c1-170
