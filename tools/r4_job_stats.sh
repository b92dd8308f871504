cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/mid
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/mid_run -o bench -- python bench.py --no-cpu-baseline > gpurun_out/mid/bench.json 2> gpurun_out/mid/bench.err
tail -c 300 gpurun_out/mid/bench.json
db=$(find /tmp/mid_run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/mid/kernel_stats.txt | head -48 | cut -c1-190
