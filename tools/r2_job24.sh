export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "proj" 2>&1 | tail -2
for rep in 1 2; do
for v in nw10 nw12; do
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --only scan_fwd 2>&1 | grep '"scan_fwd_bidir_train"\|"scan_fwd_uni"' | sed "s/^/$v /"
done
timeout 300 python tools/kbench.py --only scan_fwd 2>&1 | grep '"scan_fwd_bidir_train"\|"scan_fwd_uni"' | sed "s/^/nw8 /"
done | tee gpurun_out/r2_sweep_fwd_nw.txt
timeout 300 python tools/kbench.py --only proj 2>&1 | grep '"proj_bwd_weight' 
