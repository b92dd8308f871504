cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "scan_tm_segments or longform" 2>&1 | tail -3
AUM_SEGS=8,12,16,24 timeout 300 python tools/seg_time.py 2>&1 | grep batch | cut -c1-150
timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
AUM_DEBUG=1 AUM_TM_SEGMENTS=0 timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
