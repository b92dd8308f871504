# round 5 job 3: the ring GEMM (AUM_GEMM_RING): bit-equality with the 8-wave kernel, timing on the four projection shapes
set -x
mkdir -p gpurun_out/r5
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
O=gpurun_out/r5
GEMM_PROBE_FLAGS=4,64,128 timeout 600 python tools/gemm_probe.py > $O/gemm_probe_ring.txt 2>&1; grep -v amdgpu.ids $O/gemm_probe_ring.txt | tail -12 | cut -c1-500
cp gpurun_out/gemm_probe.json $O/gemm_probe_ring.json
