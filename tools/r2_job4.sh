set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "row_kernels" > gpurun_out/r2_pytest4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest4.log
tail -15 gpurun_out/r2_pytest4.log
timeout 300 python tools/kbench.py --only scan_ck > gpurun_out/r2_kbench4.json 2>&1; cat gpurun_out/r2_kbench4.json
