export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out
python - <<'PY'
import torch, sys
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
for shape, dt in (((4096, 768), torch.float32), ((2048, 768), torch.float32), ((42, 1536, 80), torch.float32), ((42, 48, 1536), torch.float32), ((4, 3072, 768), torch.bfloat16), ((8, 768, 1536), torch.bfloat16)):
    t = torch.randn(shape, device="cuda").to(dt)
    a = timeit(lambda: (big.zero_(), aum_hip.sum_rows(t))) - base
    b = timeit(lambda: (big.zero_(), t.sum(0, dtype=torch.float32))) - base
    print(shape, dt, f"sum_rows {a:.1f} us   torch {b:.1f} us (cold)", flush=True)
PY
for i in 1 2; do
AUM_DEBUG=1 AUM_TORCH_SUMS=1 timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b31_t$i.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b31_t$i.json'));print('torch sums',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 15 > gpurun_out/r2_b31_s$i.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r2_b31_s$i.json'));print('sum_rows  ',d['ms_per_step'],d['value'])"
done
