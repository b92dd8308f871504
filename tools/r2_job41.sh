export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "scan" 2>&1 | tail -3
for rep in 1 2 3; do
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_noflags.so timeout 300 python tools/kbench.py --only scan_bwd 2>&1 | grep '"scan_bwd' | sed "s/^/barrier /"
timeout 300 python tools/kbench.py --only scan_bwd 2>&1 | grep '"scan_bwd' | sed "s/^/flags   /"
done | tee gpurun_out/r2_ab_flags.txt
