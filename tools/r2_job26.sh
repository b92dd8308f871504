export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
for rep in 1 2; do
for v in convr4 convr16; do AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --only conv 2>&1 | grep '"conv' | sed "s/^/$v /"; done
for v in norm2048 norm1024; do AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$v.so timeout 300 python tools/kbench.py --only norm 2>&1 | grep '"rmsnorm' | sed "s/^/$v /"; done
timeout 300 python tools/kbench.py --only conv,norm 2>&1 | grep '"conv\|"rmsnorm' | sed "s/^/default /"
done | tee gpurun_out/r2_sweep_conv_norm.txt
