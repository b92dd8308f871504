# round 2, call 1: GPU suite on the RMW_BATCH-default build, default bench line, per-kernel bench
set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest1.log
tail -3 gpurun_out/r2_pytest1.log
timeout 300 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; tail -1 gpurun_out/r2_bench1.json
timeout 200 python tools/kbench.py > gpurun_out/r2_kbench1.json 2>&1; tail -20 gpurun_out/r2_kbench1.json
