set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
{ python tools/scans_trace.py; python tools/scans_trace.py --bidir; } > gpurun_out/r2_trace9.txt 2>&1
cat gpurun_out/r2_trace9.txt | grep -v amdgpu
