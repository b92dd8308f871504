mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
for i in 1 2; do
python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
AUM_DEBUG=1 AUM_WGRAD=lib python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
done
AUM_DEBUG=1 AUM_GEMM=hip python bench.py 2> gpurun_out/b.err | tail -1 | cut -c1-160
