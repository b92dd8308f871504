# Issue / wait counters of the time-serial scan kernels (run through gpurun).  Counter passes use --kernel-trace only.
# usage: tools/pmc_tm.sh <kbench --only filter> <output tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
FILTER=${1:-scan_tm}
TAG=${2:-tm}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python tools/kbench.py --only $FILTER > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python tools/kbench.py --only $FILTER > $OUT/p2.log 2>&1
python tools/pmc_tm_summary.py $OUT | tee $OUT/summary.txt
