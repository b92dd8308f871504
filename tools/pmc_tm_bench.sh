# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and vector-ALU occupancy of the token-major kernels at the bench shape
# (run through gpurun; counter passes use --kernel-trace only, as the pool's rocprofv3 policy requires).
# usage: AUM_COMMIT=<hash> bash tools/pmc_tm_bench.sh      -> gpurun_out/pmc_tmb/{pmc_traffic_tm.json, valu_busy_tm.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
OUT=gpurun_out/pmc_tmb
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python tools/tm_time.py default > $OUT/$c.log 2>&1
  find $OUT/$c -name "*counter_collection.csv" -exec cp {} $OUT/${c}.csv \;
done
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/valu -o pmc -- python tools/tm_time.py default > $OUT/valu.log 2>&1
find $OUT/valu -name "*counter_collection.csv" -exec cp {} $OUT/valu.csv \;
python tools/pmc_tm_bench_summary.py $OUT
