# same-box A/B of the token-major / channel-major block for AuM-Small at batch 64 (1536 scan waves): training and forward-only steps with the
# forward threshold at its default (1536: token-major) and at 4000 (channel-major).  -> the training-aware threshold of token_major_preferred
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PWD/tools
for m in 1536 4000; do
AUM_DEBUG=1 AUM_TM_MIN_WAVES=$m python - <<PY 2>&1 | grep -v amdgpu | grep size
import sys; sys.argv=["x"]
import variants_bench as v
v.run("small", "v1", True); v.run("small", "v1", False)
PY
done
