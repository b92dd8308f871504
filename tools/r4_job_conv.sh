cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', d['ms_per_step'], d['value'], 'conv fwd/bwd', k.get('conv_tm_fwd'), k.get('conv_tm_bwd'))"; }
run tc64
for tc in 32 57 128; do AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_convtc$tc.so run tc$tc; done
run tc64
