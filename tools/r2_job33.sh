export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python tools/variants_bench.py --only long 2>&1 | grep -v amdgpu | tail -5
timeout 900 python tools/variants_bench.py 2>&1 | grep -v amdgpu | tail -14
ls gpurun_out | grep -i variant
