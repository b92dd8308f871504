export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
AUM_WGRAD_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b34_off$i.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b34_off$i.json'));print('one stream',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b34_on$i.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b34_on$i.json'));print('overlap   ',d['ms_per_step'],d['value'],d['final_loss'])"
done
