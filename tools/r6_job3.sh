#!/bin/bash
# round 6, job 3: conv_tm kernels on channel pairs + ring windows (parity, step A/B against the round-5 kernels), the fp16 model pin
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or fp16_vs_reference or inner" 2>&1 | tail -8 | cut -c1-600 > gpurun_out/r6_pytest_job3.txt
cat gpurun_out/r6_pytest_job3.txt
bash tools/ab_job.sh bench_ab conv_tm_fwd,conv_tm_bwd d=- o=lib:convold x3 2>&1 | tee gpurun_out/r6_conv_ab.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/autocast_errors.json"))
for k, v in d.items():
    if "fp16" in k:
        print(k, json.dumps(v)[:900])
PY
