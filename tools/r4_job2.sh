set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
python tools/r4_dbg_msum.py > gpurun_out/r4_dbg_msum.txt 2>&1; cat gpurun_out/r4_dbg_msum.txt
python tools/tm_trace.py trace1 bwd > gpurun_out/r4_trace1.txt 2>&1; cat gpurun_out/r4_trace1.txt
python tools/tm_trace.py trace0 bwd > gpurun_out/r4_trace0.txt 2>&1; cat gpurun_out/r4_trace0.txt
