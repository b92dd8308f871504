# The end-of-round measurement run (through gpurun): full GPU suite, smoke(), the default bench under rocprofv3 --kernel-trace --stats
# (-> profiles/rNN_final_bench_n1.json + rNN_final_bench_kernel_stats.txt via tools/rocpd_stats.py), the micro-benchmark, a plain bench.
set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -5 gpurun_out/final/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final/run -o bench -- python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 400 gpurun_out/final/bench.json
db=$(find gpurun_out/final/run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/final/kernel_stats.txt | head -30
find gpurun_out/final/run -name "*.db" -delete
timeout 600 python tools/kbench.py 2>&1 | grep -v amdgpu > gpurun_out/final/kbench.txt; cp gpurun_out/kbench_bf16_B64.json gpurun_out/final/kbench.json
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_plain.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/final/bench_plain.json'));print('plain',d['ms_per_step'],d['value'])"
# round 3: the MFMA projection GEMM against the library GEMM (probe, interleaved rounds) and the step with the default dispatch, with the
# MFMA kernel on all four K-contiguous GEMMs, and with library GEMMs only (same box, back to back, twice)
timeout 200 python tools/gemm_probe.py 2>&1 | grep -v amdgpu > gpurun_out/final/gemm_probe.txt; cat gpurun_out/final/gemm_probe.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_gemm_auto_$i.json 2>/dev/null
  AUM_DEBUG=1 AUM_GEMM=hip timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_gemm_hip_$i.json 2>/dev/null
  AUM_DEBUG=1 AUM_GEMM=lib timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_gemm_lib_$i.json 2>/dev/null
done
python -c "
import json
for k in ('auto_1','hip_1','lib_1','auto_2','hip_2','lib_2'):
    d=json.load(open('gpurun_out/final/bench_gemm_%s.json'%k)); print(k, d['ms_per_step'], d['value'], d['kernel_ms_per_step'].get('gemm_tn'))
" | tee gpurun_out/final/gemm_step_ab.txt
