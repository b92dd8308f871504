#!/usr/bin/env python3
"""launches of the projection kernels of the token-major block at the bench shape (for rocprofv3 counter passes: tools/xdt_pmc.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-mamba-aum_amd"))
import aum_hip  # noqa: E402

u = torch.randn(64 * 513, 1536, device="cuda").bfloat16()
wx = (torch.randn(80, 1536, device="cuda") / 39).bfloat16()
w = (torch.randn(1536, 48, device="cuda") / 7).bfloat16()
for _ in range(6):
    x, d = aum_hip.xdt_tm_fwd(u, wx, w)
    aum_hip.dtproj_tm_fwd(x, 48, w)
torch.cuda.synchronize()
