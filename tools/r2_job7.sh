set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
rm -f gpurun_out/autocast_errors.json
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest7.log
tail -25 gpurun_out/r2_pytest7.log
cat gpurun_out/autocast_errors.json
timeout 400 python bench.py > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err; tail -1 gpurun_out/r2_bench7.json; tail -3 gpurun_out/r2_bench7.err
timeout 100 python tools/kbench.py --only hbm > gpurun_out/r2_kbench7.json 2>&1; cat gpurun_out/r2_kbench7.json
