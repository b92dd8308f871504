set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
export AUM_DEBUG=1
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
{
echo "== default"; python tools/kbench.py --only scan_fwd,scan_bwd 2>&1 | grep -v amdgpu.ids
echo "== fnw10"; AUM_HIP_LIB=$V/libaum_hip_fnw10.so python tools/kbench.py --only scan_fwd 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_kbench5.txt 2>&1
cat gpurun_out/r2_kbench5.txt | cut -c1-150
