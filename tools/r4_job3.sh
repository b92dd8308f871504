export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
echo "== default (z1)"; python tools/r4_dbg_msum.py 2>&1 | grep -v amdgpu.ids | head -3
echo "== z2"; python tools/r4_dbg_msum.py z2 2>&1 | grep -v amdgpu.ids | head -3
