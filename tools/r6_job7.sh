#!/bin/bash
# round 6, job 7: k_scant_bwd with one dependent link per step in both sweeps: parity (oracle), timing against the two-link build, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "scan_tm" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r6_pytest_job7.txt
cat gpurun_out/r6_pytest_job7.txt
bash tools/ab_job.sh tm_ab bwd chainold 2>&1 | tee gpurun_out/r6_tm_ab_chain.txt
bash tools/ab_job.sh bench_ab scan_tm_bwd_bidir,scan_tm_fwd_bidir d=- o=lib:chainold x3 2>&1 | tee -a gpurun_out/r6_tm_ab_chain.txt
