#!/bin/bash
# round 6, job 8: dB | dC partial rows in two planes (the reduce kernel reads one contiguous plane): parity, bit-equality, timing, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "scan_tm or inner or longform" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r6_pytest_job8.txt
cat gpurun_out/r6_pytest_job8.txt
bash tools/ab_job.sh tm_ab bwd planesold 2>&1 | tee gpurun_out/r6_tm_ab_planes.txt
bash tools/ab_job.sh bench_ab scan_tm_bwd_bidir,scan_tm_fwd_bidir d=- o=lib:planesold x3 2>&1 | tee -a gpurun_out/r6_tm_ab_planes.txt
