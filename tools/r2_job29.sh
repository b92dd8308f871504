export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/kbench.py --only norm,proj 2>&1 | grep '"rmsnorm_bwd\|"proj_bwd_weight'
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b29_$i.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b29_$i.json'));print('bench',d['ms_per_step'],d['value'])"; done
