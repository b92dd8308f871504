set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
rm -f gpurun_out/autocast_errors.json
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest_gpu.log
tail -8 gpurun_out/r4_pytest_gpu.log | cut -c1-300
for i in 1 2 3; do python tools/r4_dbg_msum.py 2>&1 | grep -v amdgpu.ids | head -2; done
python tools/r4_dbg_msum.py trace1 2>&1 | grep -v amdgpu.ids | head -2
python tools/tm_trace.py trace1 bwd > gpurun_out/r4_trace1.txt 2>&1; cat gpurun_out/r4_trace1.txt
bash tools/tm_time.sh default msum0 > gpurun_out/r4_tm_time.txt 2>&1
bash tools/tm_time.sh default msum0 > gpurun_out/r4_tm_time_b.txt 2>&1
cat gpurun_out/r4_tm_time.txt gpurun_out/r4_tm_time_b.txt | grep 'scant_bwd<'
python bench.py > gpurun_out/r4_bench_v2.json 2> gpurun_out/r4_bench_v2.err; tail -1 gpurun_out/r4_bench_v2.json | cut -c1-200
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_msum0.so python bench.py > gpurun_out/r4_bench_msum0.json 2> gpurun_out/r4_bench_msum0.err; tail -1 gpurun_out/r4_bench_msum0.json | cut -c1-200
python bench.py > gpurun_out/r4_bench_v2b.json 2> gpurun_out/r4_bench_v2b.err; tail -1 gpurun_out/r4_bench_v2b.json | cut -c1-200
