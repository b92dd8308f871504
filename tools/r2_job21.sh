set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for rep in 1 2; do
for v in fwd16 fwd32 fwd96; do
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --only scan_fwd 2>&1 | grep '"scan_fwd_bidir_train"\|"scan_fwd_uni"' | sed "s/^/$v /"
done
timeout 300 python tools/kbench.py --only scan_fwd 2>&1 | grep '"scan_fwd_bidir_train"\|"scan_fwd_uni"' | sed "s/^/rows64 /"
done | tee gpurun_out/r2_sweep_fwd_rows.txt
