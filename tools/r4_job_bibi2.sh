cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
echo "--- two streams"; timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"Bi-Bi"'
echo "--- in line"; AUM_DEBUG=1 AUM_V2_STREAMS=0 timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"Bi-Bi"'
echo "--- two streams"; timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"Bi-Bi"'
echo "--- in line"; AUM_DEBUG=1 AUM_V2_STREAMS=0 timeout 300 python tools/variants_bench.py --only bibi 2>&1 | grep '"Bi-Bi"'
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "v2 or bibi or Bi or model" 2>&1 | tail -2
