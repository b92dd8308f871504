# same-box A/B of the launch-bound tiny configuration: round-3 end tree (_r3/) against HEAD
cd $GRAFT_REPO_ROOT
for tree in . _r3 . _r3; do
  ( cd $tree; export PYTHONPATH=$PWD/audio-mamba-aum_amd; python - <<'PY'
import sys, os
sys.argv = ["x"]
sys.path.insert(0, "tools")
import importlib.util
spec = importlib.util.spec_from_file_location("vb", "tools/variants_bench.py"); vb = importlib.util.module_from_spec(spec); spec.loader.exec_module(vb)
print(os.getcwd()); vb.run("tiny", "v1", True); vb.run("small", "v1", True)
PY
  ) 2>&1 | grep -v amdgpu | tail -3
done
