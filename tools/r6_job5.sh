#!/bin/bash
# round 6, job 5: k_scant_bwd with the per-step values paired over steps: parity, bit-equality with and timing against the round-5 kernel, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "scan_tm" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r6_pytest_job5.txt
cat gpurun_out/r6_pytest_job5.txt
bash tools/ab_job.sh tm_ab bwd scanold 2>&1 | tee gpurun_out/r6_tm_ab_pairs.txt
bash tools/ab_job.sh bench_ab scan_tm_bwd_bidir,scan_tm_fwd_bidir d=- o=lib:scanold x3 2>&1 | tee -a gpurun_out/r6_tm_ab_pairs.txt
