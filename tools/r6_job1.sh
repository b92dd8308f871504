#!/bin/bash
# round 6, job 1: full GPU suite after the ABI-12 cleanup + the split last round of aum_gemm_tn; GEMM probe; step A/B of the dispatch
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | cut -c1-250 > gpurun_out/r6_pytest_full.txt
cat gpurun_out/r6_pytest_full.txt
timeout 600 python tools/gemm_abl_probe.py --variants nosplit --check nosplit 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_gemm_split.txt
bash tools/ab_job.sh bench_ab gemm_tn d=- hip=AUM_GEMM=hip lib=AUM_GEMM=lib x3 2>&1 | tee gpurun_out/r6_gemm_step_ab.txt
