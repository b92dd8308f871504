set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_frontend.py -m gpu -x -q 2>&1 | tail -3
AUM_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r2_dry2.json 2> gpurun_out/r2_dry2.err; tail -c 1200 gpurun_out/r2_dry2.json; tail -5 gpurun_out/r2_dry2.err
AUM_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --grad-compress bf16 > gpurun_out/r2_dry2c.json 2> gpurun_out/r2_dry2c.err; tail -c 600 gpurun_out/r2_dry2c.json; tail -5 gpurun_out/r2_dry2c.err
