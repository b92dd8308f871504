# rocprofv3 kernel trace of the long-form step (BASELINE config 5: B = 8, L = 4097), summarised per kernel.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/prof_long
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_long/run -o long -- python tools/variants_bench.py --only long > gpurun_out/prof_long/run.log 2>&1
tail -2 gpurun_out/prof_long/run.log
db=$(find gpurun_out/prof_long/run -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" gpurun_out/prof_long/kernel_stats.txt | head -30
rm -rf gpurun_out/prof_long/run
