export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 1500 python tools/sweep_wgrad_splits.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_sweep_wgrad_splits.txt
ls -la gpurun_out/tunableop_wgrad_sweep*.csv
