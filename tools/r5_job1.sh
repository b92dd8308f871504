# round 5 job 1: scan-backward channel sums through LDS (A/B + ablation bounds), parity of the new default and of the asm-put build,
# the new RCCL world-size-1 tests and module-level dispatch tests, bench A/B default vs round-4 butterflies on the same box
set -x
mkdir -p gpurun_out/r5
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
O=gpurun_out/r5
timeout 300 python tools/tm_ab.py bwd l0 l1 l1a l2 l2a babl1 babl2 > $O/ab_bwd_lsum.txt 2>&1; cat $O/ab_bwd_lsum.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan_tm" > $O/pytest_scan_tm_default.log 2>&1; tail -3 $O/pytest_scan_tm_default.log
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_l1a.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan_tm" > $O/pytest_scan_tm_l1a.log 2>&1; tail -3 $O/pytest_scan_tm_l1a.log
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q > $O/pytest_ddp.log 2>&1; tail -15 $O/pytest_ddp.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "longform_block or two_streams" > $O/pytest_module.log 2>&1; tail -15 $O/pytest_module.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_default_$i.json; python -c "import json;d=json.load(open('$O/bench_default_$i.json'));print('default',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_l0.so python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_l0_$i.json; python -c "import json;d=json.load(open('$O/bench_l0_$i.json'));print('l0',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_l1a.so python bench.py --no-cpu-baseline 2> $O/b.err | tail -1 > $O/bench_l1a_$i.json; python -c "import json;d=json.load(open('$O/bench_l1a_$i.json'));print('l1a',d['ms_per_step'],d['value'],d['roofline']['avg_launch_ms'])"
done
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "headline_bench_batch" > $O/pytest_headline_modes.log 2>&1; tail -5 $O/pytest_headline_modes.log
