# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and vector-ALU occupancy of the token-major kernels at the bench shape
# (run through gpurun; counter passes use --kernel-trace only, as the pool's rocprofv3 policy requires).
# usage: AUM_COMMIT=<hash> bash tools/pmc_tm_bench.sh [bench]     -> gpurun_out/pmc_tmb/{pmc_traffic_tm.json, valu_busy_tm.json}
#   `bench` (round 4): the counters are taken on bench.py's OWN launches (python bench.py --steps 2 --warmup 1), not on the stand-alone
#   harness tools/tm_time.py -- roofline.traffic in the bench line is then a measurement of the timed kernels themselves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
OUT=${AUM_PMC_OUT:-gpurun_out/pmc_tmb}
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/tm_time.py default"
if [ "$1" = "bench" ]; then CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"; export AUM_PMC_SOURCE="python bench.py --steps 2 --warmup 1 (the bench's own launches)"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- $CMD > $OUT/$c.log 2>&1
  find $OUT/$c -name "*counter_collection.csv" -exec cp {} $OUT/${c}.csv \;
done
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/valu -o pmc -- $CMD > $OUT/valu.log 2>&1
find $OUT/valu -name "*counter_collection.csv" -exec cp {} $OUT/valu.csv \;
python tools/pmc_tm_bench_summary.py $OUT
