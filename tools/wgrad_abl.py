import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip
t, n, k = 64 * 513, 3072, 768
ys = [(torch.randn(t, n, device="cuda") * 0.1).bfloat16() for _ in range(4)]
xs = [torch.randn(t, k, device="cuda").bfloat16() for _ in range(4)]
for v in ("default", "wabl1", "wabl2", "wabl3"):
    lib = aum_hip.get() if v == "default" else aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", f"libaum_hip_{v}.so"))
    f = lambda i: aum_hip.gemm_wgrad(ys[i % 4], xs[i % 4], lib=lib, partials=True)
    for i in range(3):
        f(i)
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8):
            f(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8 * 1e3)
    print(v, "%.1f us" % sorted(ts)[2], flush=True)
