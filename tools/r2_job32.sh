export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/prof32
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof32/run -o bench -- python bench.py --no-cpu-baseline > gpurun_out/prof32/bench.json 2> gpurun_out/prof32/bench.err
db=$(find gpurun_out/prof32/run -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/prof32/kernel_stats.txt | grep "k_scan_reduce\|Fill\|TOTAL\|k_sum_rows\|k_scanh\|k_scanr"
find gpurun_out/prof32/run -name "*.db" -delete
python -c "import json;d=json.load(open('gpurun_out/prof32/bench.json'));print('bench',d['ms_per_step'],d['value'])"
