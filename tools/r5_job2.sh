# round 5 job 2: the four-wave GEMM (AUM_GEMM_W4) -- bit-equality with the 8-wave kernel and timing on the four projection shapes;
# the full GPU suite with the round's new tests; a default bench
set -x
mkdir -p gpurun_out/r5
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
O=gpurun_out/r5
GEMM_PROBE_FLAGS=4,64 timeout 600 python tools/gemm_probe.py > $O/gemm_probe_w4.txt 2>&1; grep -v amdgpu.ids $O/gemm_probe_w4.txt | tail -8 | cut -c1-400
cp gpurun_out/gemm_probe.json $O/gemm_probe_w4.json
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1; tail -8 $O/pytest_gpu_full.log | cut -c1-300
python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_default_j2.json; python -c "import json;d=json.load(open('$O/bench_default_j2.json'));print('default',d['ms_per_step'],d['value'])"
