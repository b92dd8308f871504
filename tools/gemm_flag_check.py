import sys, os
sys.path.insert(0, "audio-mamba-aum_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import torch, aum_hip, cases, kernel_checks as KC
lib = aum_hip.get()
FL = int(sys.argv[1]) if len(sys.argv) > 1 else aum_hip.GEMM_PACED
for c in cases.GEMM_CASES:
    if c[3] >= 512:
        for dt in (torch.bfloat16, torch.float16):
            KC.check_gemm(lib, "cuda", c, dt, flags=FL)
        print("ok", c[0])
# more shapes: several tiles per workgroup, ragged rows, odd nk
for (m, n, k) in [(70000, 768, 768), (33000, 512, 512), (32832, 3072, 768), (5000, 1536, 1536), (257, 256, 704)]:
    KC.check_gemm(lib, "cuda", (f"m{m}_n{n}_k{k}", m, n, k, 0, 0), torch.bfloat16, flags=FL)
    print("ok", m, n, k)
