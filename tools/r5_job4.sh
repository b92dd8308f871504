# round 5 job 4: dB | dC rows of the two directions merged in the scan backward: parity, kernel A/B, step A/B
set -x
mkdir -p gpurun_out/r5
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
O=gpurun_out/r5
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "scan_tm or headline or longform or inner or repeatable or vs_reference_model" > $O/pytest_merge.log 2>&1; tail -4 $O/pytest_merge.log | cut -c1-300
timeout 300 python tools/tm_ab.py bwd nomerge > $O/ab_bwd_merge.txt 2>&1; grep -v amdgpu $O/ab_bwd_merge.txt
for i in 1 2; do
python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_merge_$i.json; python -c "import json;d=json.load(open('$O/bench_merge_$i.json'));print('merge',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_nomerge.so python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_nomerge_$i.json; python -c "import json;d=json.load(open('$O/bench_nomerge_$i.json'));print('nomerge',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
done
