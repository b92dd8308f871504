cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "xdt or wgrad" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
AUM_DEBUG=1 AUM_XDT_BWD_LIB=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
AUM_DEBUG=1 AUM_XDT_BWD_LIB=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
