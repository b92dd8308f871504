export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PWD/tools
for m in 1536 4000; do
AUM_DEBUG=1 AUM_TM_MIN_WAVES=$m python - <<PY 2>&1 | grep -v amdgpu | grep size
import sys; sys.argv=["x"]
import variants_bench as v
v.run("small", "v1", True); v.run("small", "v1", False)
PY
done
