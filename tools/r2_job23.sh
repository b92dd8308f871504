export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 600 python tools/sweep_proj_splits.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_sweep_proj_splits.txt
