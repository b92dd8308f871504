export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_r2.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=200
timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 4 > gpurun_out/r2_b37.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b37.json'));print('bench',d['ms_per_step'],d['value'])"
ls -la gpurun_out/tunableop_r2*
grep "B_6\|B_9" gpurun_out/tunableop_r2*.csv
