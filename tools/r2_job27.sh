export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for rep in 1 2; do
for v in ct16 ct48 ct64; do AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --batch 8 --len 4097 --only scan_fwd 2>&1 | grep '"scan_fwd' | sed "s/^/$v /"; done
timeout 300 python tools/kbench.py --batch 8 --len 4097 --only scan_fwd 2>&1 | grep '"scan_fwd' | sed "s/^/ct32(default) /"
done | tee gpurun_out/r2_sweep_ct_rows.txt
