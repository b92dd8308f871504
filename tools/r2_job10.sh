set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/kbench.py --only frontend,hbm_copy 2>&1 | grep -v amdgpu | tee gpurun_out/r2_kbench10.txt
timeout 600 python bench.py > gpurun_out/r2_bench10.json 2> gpurun_out/r2_bench10.err; tail -c 3000 gpurun_out/r2_bench10.json; tail -5 gpurun_out/r2_bench10.err
