mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest_gpu.log; tail -4 gpurun_out/r4_pytest_gpu.log | cut -c1-300
for i in 1 2; do
python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
AUM_DEBUG=1 AUM_WGRAD=lib python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
done
AUM_DEBUG=1 AUM_GEMM=hip python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
AUM_DEBUG=1 AUM_GEMM=lib python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
