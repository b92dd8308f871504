cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
cat > /tmp/small_inf.py <<'PY'
import sys, importlib.util
sys.argv = ["x"]
spec = importlib.util.spec_from_file_location("vb", "tools/variants_bench.py"); vb = importlib.util.module_from_spec(spec); spec.loader.exec_module(vb)
vb.run("small", "v1", False, steps=20, warm=5)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/si -o t -- python /tmp/small_inf.py 2>&1 | grep '"size"'
python tools/trace_tail_stats.py /tmp/si 25 30 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
