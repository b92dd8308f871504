set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "scan" > gpurun_out/r2_pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest2.log
tail -5 gpurun_out/r2_pytest2.log
timeout 300 python tools/kbench.py --only scan > gpurun_out/r2_kbench2.json 2>&1; cat gpurun_out/r2_kbench2.json
