# GEMM round-3 measurement job (through gpurun): parity tests of aum_gemm_tn, the probe (all schedules vs the library GEMM, interleaved
# rounds), the headline-size model goldens, and the step with the MFMA kernel vs with the library GEMMs (same box, back to back).
set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/gemm
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k gemm 2>&1 | tail -3
timeout 200 python tools/gemm_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/gemm/probe.txt
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "headline" 2>&1 | tail -5
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/gemm/bench_hip_$i.json 2>gpurun_out/gemm/bench_hip_$i.err
  AUM_DEBUG=1 AUM_GEMM_LIB=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/gemm/bench_lib_$i.json 2>gpurun_out/gemm/bench_lib_$i.err
done
python - <<'PY'
import json
for k in ("hip_1", "lib_1", "hip_2", "lib_2"):
    try:
        d = json.load(open(f"gpurun_out/gemm/bench_{k}.json"))
        print(k, d["ms_per_step"], d["value"], d.get("final_loss"))
    except Exception as e:
        print(k, "failed", e); print(open(f"gpurun_out/gemm/bench_{k}.err").read()[-1500:])
PY
