cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
AUM_SEGS=9,10,16,20,21,22 timeout 300 python tools/seg_time.py 2>&1 | grep batch | cut -c1-150
AUM_SEGS=16 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/seg_prof -o t -- python tools/seg_time.py > /dev/null 2>&1
find gpurun_out/seg_prof -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-200
