# GEMM solution recording (through gpurun): runs the bench and the other configurations with TunableOp's online tuning at a longer trial time and
# writes every (shape -> solution) pair the processes used to gpurun_out/tunable_*.csv; tools/merge_tunable.py adds the ones the recorded file
# (audio-mamba-aum_amd/aum/tunableop_gfx950.csv) lacks.  Needed whenever a dispatch default changes which library GEMMs a step issues.
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${AUM_TUNE_MS:-150}
mkdir -p gpurun_out
AUM_TUNABLEOP_DUMP=$PWD/gpurun_out/tunable_bench.csv timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline | tail -c 200; echo
AUM_TUNABLEOP_DUMP=$PWD/gpurun_out/tunable_variants.csv timeout 900 python tools/variants_bench.py 2>&1 | grep '"size"' | cut -c1-200
AUM_TUNABLEOP_DUMP=$PWD/gpurun_out/tunable_bibi_ddp.csv timeout 600 python tools/variants_bench.py --only bibi_ddp 2>&1 | grep '"size"' | cut -c1-200
wc -l gpurun_out/tunable_*.csv
