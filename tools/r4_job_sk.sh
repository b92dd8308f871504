cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd
V=$PWD/audio-mamba-aum_amd/aum_hip/variants
cat > /tmp/skt.py <<'PY'
import torch, statistics, sys
import aum_hip
lib = aum_hip.get()
M = 64 * 513
for (N, K) in ((768, 1536), (768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); r = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize(); r.append(a.elapsed_time(b) * 100)
        return statistics.median(r)
    print(N, K, "split", round(t(lambda: aum_hip.gemm_tn(x, w, out=o, split_tail=True)), 1), "whole", round(t(lambda: aum_hip.gemm_tn(x, w, out=o, split_tail=False)), 1))
PY
for v in default skabl1 skabl2 skabl3; do
  echo "--- $v"
  if [ $v = default ]; then python /tmp/skt.py 2>&1 | grep -v amdgpu; else AUM_DEBUG=1 AUM_HIP_LIB=$V/libaum_hip_$v.so python /tmp/skt.py 2>&1 | grep -v amdgpu; fi
done
