cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
echo "--- token-major (default now)"; timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"size"'
echo "--- channel-major for one-direction training (AUM_TM_MIN_WAVES=2048)"; AUM_DEBUG=1 AUM_TM_MIN_WAVES=2048 timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"size"'
echo "--- token-major again"; timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"size"'
