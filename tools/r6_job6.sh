#!/bin/bash
# round 6, job 6: k_scant_fwd with the per-step values paired over steps: parity, bit-equality / timing against the step-by-step build; order bias of tm_ab
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "scan_tm" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r6_pytest_job6.txt
cat gpurun_out/r6_pytest_job6.txt
bash tools/ab_job.sh tm_ab fwd fwdold same 2>&1 | tee gpurun_out/r6_tm_ab_fwd_pairs.txt
bash tools/ab_job.sh tm_ab bwd same scanold 2>&1 | tee -a gpurun_out/r6_tm_ab_fwd_pairs.txt
bash tools/ab_job.sh bench_ab scan_tm_bwd_bidir,scan_tm_fwd_bidir d=- o=lib:fwdold x3 2>&1 | tee -a gpurun_out/r6_tm_ab_fwd_pairs.txt
