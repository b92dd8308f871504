# round 6: the end-of-round measurement run (tools/final_job.sh + the round's extras)
export AUM_COMMIT=${AUM_COMMIT:-71fb870}
bash tools/final_job.sh
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python tools/variants_bench.py --only bibi_ddp > gpurun_out/final/variants_bibi_ddp.log 2>&1; grep '"size"' gpurun_out/final/variants_bibi_ddp.log
cp gpurun_out/variants_bench.json gpurun_out/variants_bench_bibi_ddp.json gpurun_out/variants_bench_long.json gpurun_out/final/ 2>/dev/null
AUM_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_forced_ddp.json 2>/dev/null; grep '^{' gpurun_out/final/bench_forced_ddp.json | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('forced ddp',d['ms_per_step'],d['value'])"
