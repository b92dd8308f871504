set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for L in 513 512 528 544; do
echo "== L=$L"
timeout 300 python tools/kbench.py --only proj,conv,norm --len $L 2>&1 | grep -v amdgpu | grep -v lib_
done | tee gpurun_out/r2_kbench14.txt
