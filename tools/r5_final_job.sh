# round 5: the end-of-round measurement run (tools/final_job.sh + the round's extras)
export AUM_COMMIT=d7a790c
bash tools/final_job.sh
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 600 python tools/variants_bench.py --only bibi_ddp > gpurun_out/final/variants_bibi_ddp.log 2>&1; grep '"size"' gpurun_out/final/variants_bibi_ddp.log
cp gpurun_out/variants_bench.json gpurun_out/variants_bench_bibi_ddp.json gpurun_out/variants_bench_long.json gpurun_out/final/ 2>/dev/null
tools/_bin/mfma_probe 2>&1 | grep -v amdgpu > gpurun_out/final/mfma_probe.txt; cat gpurun_out/final/mfma_probe.txt
