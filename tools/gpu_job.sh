# Round-end measurement recipe (run through gpurun): GPU parity tests, the default bench line, a per-kernel profile.
set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err; tail -1 gpurun_out/bench_latest.json
