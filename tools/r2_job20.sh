set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "scan" 2>&1 | tail -4
for B in 64 64 8 16; do
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_fwdfixed.so timeout 300 python tools/kbench.py --only scan_fwd --batch $B 2>&1 | grep '"scan_fwd' | sed "s/^/fixed B$B /"
timeout 300 python tools/kbench.py --only scan_fwd --batch $B 2>&1 | grep '"scan_fwd' | sed "s/^/sized B$B /"
done | tee gpurun_out/r2_ab_rows_fwd.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b20.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b20.json'));print('bench',d['ms_per_step'],d['value'],d['kernel_ms_per_step'])"
