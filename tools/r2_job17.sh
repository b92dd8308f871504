set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_frontend.py -m gpu -x -q -k "headline_grid or fbank_gpu" 2>&1 | tail -12
