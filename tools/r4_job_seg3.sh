cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/long_prof -o t -- python tools/variants_bench.py --only long > gpurun_out/long_prof.log 2>&1
tail -1 gpurun_out/long_prof.log
python tools/trace_tail_stats.py /tmp/long_prof 4 45 | tee gpurun_out/long_prof_tail_stats.txt | cut -c1-230
