export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
AUM_GEMM_REMAINDER_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/r2_b40.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b40.json'));print('one stream ',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/r2_b40.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b40.json'));print('side stream',d['ms_per_step'],d['value'])"
AUM_GEMM_TOKEN_SPLIT=15 timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/r2_b40.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b40.json'));print('side, mask 15',d['ms_per_step'],d['value'])"
done | tee gpurun_out/r2_ab_remainder_stream.txt
