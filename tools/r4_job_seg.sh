# round 4: time-segmented token-major scan -- parity on the GPU, the segment-count sweep, and the long-form step A/B on one box
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "scan_tm_segments or longform" 2>&1 | tail -6
timeout 300 python tools/seg_time.py > gpurun_out/seg_time.log 2>&1; tail -12 gpurun_out/seg_time.log | cut -c1-260
echo "--- long-form step: channel-major (AUM_TM_SEGMENTS=0)"
AUM_DEBUG=1 AUM_TM_SEGMENTS=0 timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
cp gpurun_out/variants_bench_long.json gpurun_out/variants_bench_long_channel_major.json
echo "--- long-form step: time segments (default dispatch)"
timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
echo "--- channel-major again"
AUM_DEBUG=1 AUM_TM_SEGMENTS=0 timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
timeout 300 python tools/variants_bench.py --only long 2>&1 | tail -1
