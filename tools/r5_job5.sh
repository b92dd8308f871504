# round 5 job 5: the scan backward's two waves per SIMD taking turns at priority: parity of the scan / model tests, kernel and step A/B
set -x
mkdir -p gpurun_out/r5
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
O=gpurun_out/r5
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "scan_tm or headline or longform or inner or repeatable or vs_reference_model" > $O/pytest_bprio.log 2>&1; tail -3 $O/pytest_bprio.log | cut -c1-300
timeout 300 python tools/tm_ab.py bwd bprio0 > $O/ab_bwd_prio.txt 2>&1; grep -v amdgpu $O/ab_bwd_prio.txt
for i in 1 2; do
python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_bprio_$i.json; python -c "import json;d=json.load(open('$O/bench_bprio_$i.json'));print('turns',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_bprio0.so python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_bprio0_$i.json; python -c "import json;d=json.load(open('$O/bench_bprio0_$i.json'));print('no priorities',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
done
timeout 300 python tools/variants_bench.py --only long 2>&1 | grep '"size"' | cut -c1-200
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_bprio0.so timeout 300 python tools/variants_bench.py --only long 2>&1 | grep '"size"' | cut -c1-200
