cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_tn" 2>&1 | tail -2
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', d['ms_per_step'], d['value'], 'gemm_tn', k.get('gemm_tn'))"; }
run default
AUM_DEBUG=1 AUM_GEMM_SHAPES=1536x768,3072x768,768x3072 run plus_in_proj_dgrad
AUM_DEBUG=1 AUM_GEMM=hip run all_four
run default
AUM_DEBUG=1 AUM_GEMM_SHAPES=1536x768,3072x768,768x3072 run plus_in_proj_dgrad
AUM_DEBUG=1 AUM_GEMM=hip run all_four
