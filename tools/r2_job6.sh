set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "scan" > gpurun_out/r2_pytest6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest6.log
tail -4 gpurun_out/r2_pytest6.log
export AUM_DEBUG=1
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
{
echo "== default (no reload)"; python tools/kbench.py --only scan_bwd,ablate 2>&1 | grep -v amdgpu.ids
echo "== reload"; AUM_HIP_LIB=$V/libaum_hip_reload.so python tools/kbench.py --only scan_bwd,ablate 2>&1 | grep -v amdgpu.ids
echo "== default again"; python tools/kbench.py --only scan_bwd 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_kbench6.txt 2>&1
cat gpurun_out/r2_kbench6.txt | cut -c1-120
