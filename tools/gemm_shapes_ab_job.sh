# same-box A/B of the step with different (N, K) sets of the projection-GEMM dispatch on aum_gemm_tn (AUM_DEBUG=1 AUM_GEMM_SHAPES=...)
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
for i in 1 2; do
for sh in "1536x768" "1536x768,3072x768" "1536x768,3072x768,768x1536" "1x1"; do
  AUM_DEBUG=1 AUM_GEMM_SHAPES=$sh timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sh', d['ms_per_step'], d['value'], d['kernel_ms_per_step'].get('gemm_tn'))"
done
done
