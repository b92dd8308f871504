export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for rep in 1 2; do
for sp in 4,8 19,9 6,9 4,9 19,27; do
AUM_WGRAD_SPLITS=$sp timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/r2_b36.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b36.json'));print('splits $sp',d['ms_per_step'],d['value'])"
done; done | tee gpurun_out/r2_ab_wgrad_splits.txt
