set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -k "scan_tm or wave_sum or repeatable or headline" > gpurun_out/r4_pytest_tail2.log 2>&1; tail -3 gpurun_out/r4_pytest_tail2.log
bash tools/tm_time.sh default tail0 > gpurun_out/r4_tm_time.txt 2>&1
bash tools/tm_time.sh default tail0 > gpurun_out/r4_tm_time_b.txt 2>&1
cat gpurun_out/r4_tm_time.txt gpurun_out/r4_tm_time_b.txt | grep 'scant_bwd<'
python bench.py > gpurun_out/r4_bench_tail2.json 2> gpurun_out/r4_bench_tail2.err; tail -1 gpurun_out/r4_bench_tail2.json | cut -c1-200
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_tail0.so python bench.py > gpurun_out/r4_bench_tail0.json 2> gpurun_out/r4_bench_tail0.err; tail -1 gpurun_out/r4_bench_tail0.json | cut -c1-200
python bench.py > gpurun_out/r4_bench_tail2b.json 2> gpurun_out/r4_bench_tail2b.err; tail -1 gpurun_out/r4_bench_tail2b.json | cut -c1-200
