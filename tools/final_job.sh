# The end-of-round measurement run (through gpurun): full GPU suite, smoke(), the default bench under rocprofv3 --kernel-trace --stats
# (-> profiles/rNN_final_bench_n1.json + rNN_final_bench_kernel_stats.txt via tools/rocpd_stats.py), PMC counters on the bench's own launches,
# a plain bench, the other configurations.
set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -5 gpurun_out/final/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/final_run -o bench -- python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 400 gpurun_out/final/bench.json
db=$(find /tmp/final_run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/final/kernel_stats.txt | head -30
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_plain.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/final/bench_plain.json'));print('plain',d['ms_per_step'],d['value'])"
AUM_PMC_OUT=/tmp/pmc_tmb AUM_COMMIT=${AUM_COMMIT:-unknown} timeout 900 bash tools/pmc_tm_bench.sh bench > gpurun_out/final/pmc.log 2>&1; tail -30 gpurun_out/final/pmc.log
cp /tmp/pmc_tmb/pmc_traffic_tm.json /tmp/pmc_tmb/valu_busy_tm.json gpurun_out/final/ 2>/dev/null
timeout 900 python tools/variants_bench.py > gpurun_out/final/variants.log 2>&1; grep '"size"' gpurun_out/final/variants.log
cp gpurun_out/autocast_errors.json gpurun_out/final/ 2>/dev/null
