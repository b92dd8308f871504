export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sum_rows or norm" 2>&1 | tail -2
python - <<'PY'
import torch
import aum_hip
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
base = timeit(lambda: big.zero_())
t = torch.randn(4096, 768, device="cuda")
print("4096x768: sum_rows (two-stage) %.1f us, torch %.1f us (cold)" % (timeit(lambda: (big.zero_(), aum_hip.sum_rows(t))) - base, timeit(lambda: (big.zero_(), t.sum(0))) - base))
PY
for i in 1 2; do
AUM_DEBUG=1 AUM_TORCH_SUMS=1 timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b42.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b42.json'));print('torch sums',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b42.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b42.json'));print('sum_rows  ',d['ms_per_step'],d['value'])"
done
