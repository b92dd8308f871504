export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_r2b.csv
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
for m in 0 15 1 2 4 8 13 0 15 13; do
AUM_GEMM_TOKEN_SPLIT=$m timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/r2_b39.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b39.json'));print('mask $m',d['ms_per_step'],d['value'])"
done | tee gpurun_out/r2_ab_token_split.txt
