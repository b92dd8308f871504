# round-4 GPU call 1: full GPU suite (new oracle grid tests, same-precision headline pin), scan timing A/B of the matrix-pipe channel
# sums (default = MSUM 1, variants msum0 = round-3 butterflies, msum2 = half tiles issued inside the sweeps), instruction-rate probe, bench.
set -x
mkdir -p gpurun_out tools/_bin
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest_gpu.log
tail -15 gpurun_out/r4_pytest_gpu.log
bash tools/tm_time.sh default msum0 msum2 > gpurun_out/r4_tm_time.txt 2>&1
bash tools/tm_time.sh default msum0 msum2 > gpurun_out/r4_tm_time_b.txt 2>&1
cat gpurun_out/r4_tm_time.txt gpurun_out/r4_tm_time_b.txt | grep 'scant'
hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/_bin/valu_probe && timeout 300 tools/_bin/valu_probe > gpurun_out/r4_valu_probe.txt 2>&1
python bench.py > gpurun_out/r4_bench_v1.json 2> gpurun_out/r4_bench_v1.err; tail -1 gpurun_out/r4_bench_v1.json | cut -c1-600
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_msum0.so python bench.py > gpurun_out/r4_bench_msum0.json 2> gpurun_out/r4_bench_msum0.err; tail -1 gpurun_out/r4_bench_msum0.json | cut -c1-300
