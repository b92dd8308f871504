#!/bin/bash
# round 6, job 2: dead-wave path of the paced GEMM kernel (A/B builds), GEMM parity tests, the fp16 scan grid tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or grid_b64_fp16" 2>&1 | tail -8 | cut -c1-400 > gpurun_out/r6_pytest_job2.txt
cat gpurun_out/r6_pytest_job2.txt
timeout 600 python tools/gemm_abl_probe.py --variants nosplit,nodead,deadonly --check nosplit,nodead,deadonly 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_gemm_dead.txt
