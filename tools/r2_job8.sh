set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "state_kernel or row_kernels" > gpurun_out/r2_pytest8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest8.log
tail -6 gpurun_out/r2_pytest8.log
timeout 300 python tools/kbench.py --only scan_ck > gpurun_out/r2_kbench8.json 2>&1; grep -v amdgpu gpurun_out/r2_kbench8.json | cut -c1-120
