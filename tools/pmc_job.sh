# HBM traffic counters of the hand-written kernels (run through gpurun).  Counter passes are separate runs and use
# --kernel-trace only, as the pool's rocprofv3 policy requires.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$c -o pmc -- python tools/kbench.py --only "$1" > gpurun_out/pmc/$c.log 2>&1
  find gpurun_out/pmc/$c -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc/${c}.csv \;
done
python tools/pmc_summary.py gpurun_out/pmc/FETCH_SIZE.csv gpurun_out/pmc/WRITE_SIZE.csv gpurun_out/pmc/pmc_traffic.json
