set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
export AUM_DEBUG=1
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
{
echo "== default dmajor"; python tools/kbench.py --only scan_ck,hbm 2>&1 | grep -v amdgpu.ids
echo "== default bmajor"; python tools/kbench.py --only scan_ck,scan_bwd,scan_fwd --layout bmajor 2>&1 | grep -v amdgpu.ids
for v in noload nw8 nw16; do echo "== $v dmajor"; AUM_HIP_LIB=$V/libaum_hip_$v.so python tools/kbench.py --only scan_ck 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r2_kbench3.txt 2>&1
cat gpurun_out/r2_kbench3.txt | cut -c1-150
