cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv_tm" 2>&1 | tail -2
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', d['ms_per_step'], d['value'], 'conv fwd/bwd', k.get('conv_tm_fwd'), k.get('conv_tm_bwd'))"; }
run two_blocks_in_flight
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_convold.so run one_block
run two_blocks_in_flight
AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_convold.so run one_block
