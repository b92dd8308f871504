# same-box A/B of the fused x_proj + dt_proj kernel (aum_xdt_tm_fwd) against library x_proj + aum_dtproj_tm_fwd: parity tests, alone, the step
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "xdt or headline or repeatable or inner_fns or token_major" 2>&1 | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, "audio-mamba-aum_amd")
import aum_hip
from aum import tunable
tunable.enable()
u = torch.randn(64*513, 1536, device="cuda").bfloat16(); wx = (torch.randn(80, 1536, device="cuda")/39).bfloat16(); w = (torch.randn(1536, 48, device="cuda")/7).bfloat16()
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def two():
    x = torch.matmul(u, wx.t()); return aum_hip.dtproj_tm_fwd(x, 48, w)
for r in range(3):
    print("xdt fused %.1f us   library x_proj + dtproj kernel %.1f us" % (t(lambda: aum_hip.xdt_tm_fwd(u, wx, w)), t(two)))
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['ms_per_step'], d['value'], d['kernel_ms_per_step'].get('xdt_tm_fwd'))"
  AUM_DEBUG=1 AUM_XDT_LIB=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split', d['ms_per_step'], d['value'], d['kernel_ms_per_step'].get('dtproj_tm_fwd'))"
done
