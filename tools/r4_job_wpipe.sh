cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad" 2>&1 | tail -2
echo "--- pipelined (default build)"; timeout 200 python tools/wgrad_probe.py 2>&1 | grep -v amdgpu | tail -4
echo "--- as first built (-DAUM_WGRAD_PIPE=0)"; AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_wpipe0.so timeout 200 python tools/wgrad_probe.py 2>&1 | grep -v amdgpu | tail -4
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', d['ms_per_step'], d['value'], 'gemm_wgrad', k.get('gemm_wgrad'))"; }
run pipelined
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_wpipe0.so run first_built
run pipelined
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_wpipe0.so run first_built
