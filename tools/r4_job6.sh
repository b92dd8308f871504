set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "vs_reference_model or inner_fns" > gpurun_out/r4_pytest_bibi.log 2>&1; tail -5 gpurun_out/r4_pytest_bibi.log | cut -c1-300
python tools/variants_bench.py --only bibi 2>&1 | grep clips_per_s
AUM_DEBUG=1 AUM_TM_MIN_WAVES=1000000000 python tools/variants_bench.py --only bibi 2>&1 | grep clips_per_s
