export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=150
timeout 1200 python tools/sweep_gemm_tokens.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_sweep_gemm_tokens.txt
