#!/bin/bash
# round 6, job 4: conv_tm waves grouped into workgroups of whole token rows: parity + step A/B (default / one wave per workgroup / round-5 kernels)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or inner" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r6_pytest_job4.txt
cat gpurun_out/r6_pytest_job4.txt
bash tools/ab_job.sh bench_ab conv_tm_fwd,conv_tm_bwd d=- w1=lib:convw1 o=lib:convold x3 2>&1 | tee gpurun_out/r6_conv_ab2.txt
