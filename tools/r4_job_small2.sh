cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
timeout 600 python -m pytest tests -m gpu -q -x -k "xdt or small" 2>&1 | tail -3
cat > /tmp/small_inf.py <<'PY'
import sys, importlib.util
sys.argv = ["x"]
spec = importlib.util.spec_from_file_location("vb", "tools/variants_bench.py"); vb = importlib.util.module_from_spec(spec); spec.loader.exec_module(vb)
vb.run("small", "v1", False, steps=20, warm=5)
PY
python /tmp/small_inf.py 2>&1 | grep '"size"'
AUM_DEBUG=1 AUM_XDT_LIB=1 python /tmp/small_inf.py 2>&1 | grep '"size"'
python /tmp/small_inf.py 2>&1 | grep '"size"'
AUM_DEBUG=1 AUM_XDT_LIB=1 python /tmp/small_inf.py 2>&1 | grep '"size"'
