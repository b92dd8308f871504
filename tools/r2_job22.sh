set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "scan" 2>&1 | tail -3
for rep in 1 2; do
AUM_DEBUG=1 AUM_HIP_LIB=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_bwd192.so timeout 300 python tools/kbench.py --only scan_bwd 2>&1 | grep '"scan_bwd' | sed "s/^/bwd192 /"
timeout 300 python tools/kbench.py --only scan_bwd,scan_fwd 2>&1 | grep '"scan_bwd\|"scan_fwd' | sed "s/^/default /"
done | tee gpurun_out/r2_sweep_bwd_rows.txt
for B in 8 16; do timeout 300 python tools/kbench.py --only scan_fwd --batch $B 2>&1 | grep '"scan_fwd' | sed "s/^/default B$B /"; done | tee -a gpurun_out/r2_sweep_bwd_rows.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b22.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b22.json'));print('bench',d['ms_per_step'],d['value'],d['kernel_ms_per_step'])"
