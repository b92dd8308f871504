#!/bin/bash
# One parameterised GPU job script (the 28 one-off tools/r4_job*.sh are folded in here).  Run through gpurun:
#   gpurun -- 'bash tools/ab_job.sh <recipe> [args ...]'          several recipes: separate them with ' -- '
# Recipes
#   pytest <-k expression> [files ...]      GPU parity tests selected by expression (default files: tests)
#   bench_ab <key,key,...> <label>=<spec> [<label>=<spec> ...] [x<rounds>]
#                                            alternating same-box bench.py runs (--no-cpu-baseline), one line per run: label, ms/step,
#                                            clips/s and kernel_ms_per_step[key] for every key.  <spec> is a comma list of
#                                            ENV=VALUE settings (AUM_DEBUG=1 is implied) and/or lib:<variant> (a library built by
#                                            tools/build_variant.sh); "-" = the default build and dispatch.
#   tm_ab <fwd|bwd> <variant> [...]          tools/tm_ab.py: the time-serial scan kernels, default library against build variants
#   variants <only>                          tools/variants_bench.py --only <only>   (bibi | long | ...), optionally after ENV=VALUE words
#   profile [bench.py args]                  bench.py under rocprofv3 --kernel-trace --stats -> gpurun_out/prof/kernel_stats.txt
#   probe <script.py> [args]                 any tools/*.py probe (gemm_probe.py, wgrad_probe.py, seg_time.py, skinny_probe.py ...)
# Round-4 measurements and the invocation that repeats them (files under profiles/):
#   r04_ab_msum.txt / r04_ab_tail2.txt       tm_ab bwd msum0   |  tm_ab bwd tail0          (+ bench_ab scan_tm_bwd_bidir -=- m=lib:msum0)
#   r04_gemm_split_tail.txt (step A/B)       bench_ab gemm_tn d=- p=AUM_GEMM_SHAPES=1536x768,3072x768,768x3072 a=AUM_GEMM=hip x2
#   r04_ab_bibi_token_major.txt              variants bibi  --  variants AUM_V2_STREAMS=0 bibi  --  variants AUM_TM_MIN_WAVES=1000000000 bibi
#   r04_ab_longform_segments.txt             variants long  --  variants AUM_TM_SEGMENTS=0 long  --  probe seg_time.py
#   r04_ab_skinny_gradients.txt              bench_ab xdt_tm_bwd,gemm_wgrad d=- l=AUM_XDT_BWD_LIB=1 x2
#   r04_ab_conv_tm.txt                       bench_ab conv_tm_fwd,conv_tm_bwd d=- o=lib:convold x2
#   wgrad pipelining / AUM_WGRAD=lib         bench_ab gemm_wgrad d=- w=lib:wpipe0 l=AUM_WGRAD=lib x2  --  probe wgrad_probe.py
# Round-5 measurements (build the named variants first: tools/build_variant.sh <name> <flags>; finish with csrc/build.py --force for the default):
#   r05_ab_lsum.txt          l0 -DAUM_SCANT_LSUM=0 | l1 -DAUM_SCANT_LSUM=1 | l1a + -DAUM_LSUM_PUT_ASM=1 | babl1 / babl2 -DAUM_SCANT_BABL=1 / 2
#                            tm_ab bwd l0 l1 l1a babl1 babl2  --  pytest scan_tm  --  bench_ab scan_tm_bwd_bidir d=- l0=lib:l0 l1a=lib:l1a x2
#   r05_ab_dbc_merge.txt     nomerge -DAUM_SCANT_DBC_MERGE=0:  tm_ab bwd nomerge  --  bench_ab scan_tm_bwd_bidir d=- n=lib:nomerge x2
#   r05_ab_bwd_prio.txt      bprio0 -DAUM_SCANT_BPRIO=0 (others: -DAUM_SCANT_BPRIO_SHIFT=1|2, _HI=1, BPRIO=2|3):  tm_ab bwd bprio0  --  bench_ab scan_tm_bwd_bidir d=- p0=lib:bprio0 x2
#                            --  variants long  (and the same after AUM_HIP_LIB=.../libaum_hip_bprio0.so)
#   r05_gemm_w4_ring.txt     GEMM_PROBE_FLAGS=4,64,128 probe gemm_probe.py      (flags 64 / 128: the four-wave / ring forms of aum_gemm_tn)
#   r05_gemm_dispatch_ab.txt bench_ab gemm_tn d=- nofwd=AUM_GEMM_SHAPES=1536x768 none=AUM_GEMM_SHAPES=1x1 fwdonly=AUM_GEMM_SHAPES=3072x768 x3
#                            bench_ab gemm_tn d=- m13=AUM_GEMM_TOKEN_SPLIT=13 m1=AUM_GEMM_TOKEN_SPLIT=1 m8=AUM_GEMM_TOKEN_SPLIT=8 x3      (d = 0 now)
#   r05_ab_sum_multi.txt     bench_ab "" d=- o=AUM_SUMS_ONE_BY_ONE=1 x3
#   r05_ab_conv_tc.txt       cv16 -DAUM_CONVT_NB16_BWD=16 -DAUM_CONVT_TC_BWD=64 (the round-4 backward) | cvb2 -DAUM_CONVT_BWD_BLOCKS=2:
#                            bench_ab conv_tm_fwd,conv_tm_bwd d=- o=lib:cv16 b=lib:cvb2 x2
#   r05_ab_bibi_ddp_streams.txt   variants bibi_ddp
#   GEMM solutions (aum/tunableop_gfx950.csv)   tools/tune_job.sh, then tools/merge_tunable.py gpurun_out/tunable_*.csv
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD/audio-mamba-aum_amd:${PYTHONPATH:-}
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
mkdir -p gpurun_out

apply_spec() {          # sets the environment of ONE run from a spec
    local spec=$1 item
    [ "$spec" = "-" ] && return
    export AUM_DEBUG=1
    IFS=',' read -ra items <<< "$spec"
    for item in "${items[@]}"; do
        case $item in
            lib:*) export AUM_HIP_LIB=$V/libaum_hip_${item#lib:}.so ;;
            *=*) export "$item" ;;
        esac
    done
}

run_recipe() {
    local r=$1; shift
    case $r in
        pytest)
            local expr=$1; shift
            timeout 1500 python -m pytest "${@:-tests}" -m gpu -q -x -k "$expr" 2>&1 | tail -6 | cut -c1-250 ;;
        bench_ab)
            local keys=$1; shift
            local rounds=1 specs=()
            for a in "$@"; do case $a in x[0-9]*) rounds=${a#x} ;; *) specs+=("$a") ;; esac; done
            for ((i = 0; i < rounds; ++i)); do for s in "${specs[@]}"; do
                ( apply_spec "${s#*=}"; python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2> gpurun_out/b.err | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']
print('${s%%=*}', d['ms_per_step'], d['value'], *[(n, k.get(n)) for n in '$keys'.split(',') if n])" )
            done; done ;;
        tm_ab) timeout 600 python tools/tm_ab.py "$@" 2>&1 | grep -v amdgpu.ids ;;
        variants)
            ( while [[ ${1:-} == *=* ]]; do export AUM_DEBUG=1 "$1"; shift; done
              timeout 600 python tools/variants_bench.py --only "$1" 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300 ) ;;
        profile)
            cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
            rm -rf /tmp/prof_run; mkdir -p gpurun_out/prof
            timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_run -o bench -- python bench.py "$@" > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
            tail -c 300 gpurun_out/prof/bench.json; echo
            python tools/rocpd_stats.py "$(find /tmp/prof_run -name '*.db' | head -1)" gpurun_out/prof/kernel_stats.txt | head -40 | cut -c1-190 ;;
        probe) local script=$1; shift; timeout 900 python tools/$script "$@" 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-300 ;;
        *) echo "unknown recipe $r"; return 2 ;;
    esac
}

args=()
for a in "$@" --; do
    if [ "$a" = "--" ]; then
        [ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_recipe "${args[@]}"; }
        args=()
    else
        args+=("$a")
    fi
done
