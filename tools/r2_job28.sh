export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for rep in 1 2; do
for v in ch16 ch32 ch64; do AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --batch 8 --len 4097 --only scan_bwd 2>&1 | grep '"scan_bwd' | sed "s/^/$v /"; done
timeout 300 python tools/kbench.py --batch 8 --len 4097 --only scan_bwd,scan_fwd 2>&1 | grep '"scan_bwd\|"scan_fwd' | sed "s/^/default(48) /"
done | tee gpurun_out/r2_sweep_ch_rows.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "long" 2>&1 | tail -2
