set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | tail -15
timeout 300 tools/_bin/hbm_probe | tee gpurun_out/r2_hbm_probe.txt
