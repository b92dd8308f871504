# same-box A/B of the dt-projection MFMA kernel (aum_dtproj_tm_fwd) against the library GEMM: parity tests, the kernel alone, the step
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "dtproj or headline or repeatable or inner_fns or token_major" 2>&1 | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, "audio-mamba-aum_amd")
import aum_hip
from aum import tunable
tunable.enable()
x = torch.randn(64*513, 80, device="cuda").bfloat16(); w = (torch.randn(1536, 48, device="cuda")/7).bfloat16()
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for r in range(3):
    print("dtproj hip %.1f us   lib %.1f us" % (t(lambda: aum_hip.dtproj_tm_fwd(x, 48, w)), t(lambda: torch.matmul(x[:, :48], w.t()))))
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hip', d['ms_per_step'], d['value'], d['kernel_ms_per_step'].get('dtproj_tm_fwd'))"
  AUM_DEBUG=1 AUM_DTPROJ_LIB=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib', d['ms_per_step'], d['value'])"
done
