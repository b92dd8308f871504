mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" > gpurun_out/r4_pytest_wgrad.log 2>&1; tail -3 gpurun_out/r4_pytest_wgrad.log | cut -c1-250
timeout 600 python tools/wgrad_probe.py > gpurun_out/r4_wgrad_probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_wgrad_probe.txt | tail -6
