"""debug: where do the matrix-pipe channel sums differ from the butterfly build at the bench launch?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
import aum_hip
ref = aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", "libaum_hip_msum0.so"))
new = aum_hip.get() if len(sys.argv) < 2 else aum_hip.Lib(os.path.join(ROOT, "audio-mamba-aum_amd", "aum_hip", "variants", f"libaum_hip_{sys.argv[1]}.so"))
for (Bsz, E) in ((64, 1536),):
    torch.manual_seed(5)
    L, N, R, dev = 513, 16, 48, "cuda"
    bf = lambda t: t.bfloat16()
    xz = bf(torch.randn(Bsz, L, 2 * E, device=dev)); u, z = bf(torch.randn(Bsz, L, E, device=dev)), xz[:, :, E:]
    dl = bf(0.5 * torch.randn(Bsz, L, E, device=dev)); x_dbl = bf(torch.randn(Bsz, L, R + 2 * N, device=dev))
    Bm, Cm = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(E, 1) * (1 + 0.1 * torch.rand(E, N, device=dev))
    A_b = A * (1 + 0.1 * torch.rand(E, N, device=dev))
    D, bias = torch.rand(E, device=dev) + 0.5, torch.full((E,), -4.0, device=dev) + torch.rand(E, device=dev)
    dout = bf(torch.randn(Bsz, L, E, device=dev))
    ck = aum_hip.scan_tm_ckpt(Bsz, L, E, N, True, dev, dtype=torch.bfloat16)
    out, pre = aum_hip.scan_tm_fwd(u, dl, A, Bm, Cm, D, z, bias, True, A_b=A_b, want_out_pre=True, ckpt=ck, lib=ref)
    g0 = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=ref)
    g1 = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=new)
    g2 = aum_hip.scan_tm_bwd(u, dl, A, Bm, Cm, D, z, bias, dout, pre, ck, True, A_b=A_b, lib=new)
    a, b, c = g0["dBC"].cpu().numpy(), g1["dBC"].cpu().numpy(), g2["dBC"].cpu().numpy()
    sc = np.abs(a).max()
    bad = np.abs(a - b) > 0.02 * sc
    print(f"B={Bsz} E={E}: max|new-ref|/max = {np.abs(a-b).max()/sc:.3g}; bad elements {bad.sum()} of {bad.size}; run-to-run differing {(b != c).sum()}")
    if bad.any():
        bi, ti, ci = np.nonzero(bad)
        print("  bad by column (0-15 dB, 16-31 dC):", np.bincount(ci, minlength=32).tolist())
        print("  bad by t mod 8:", np.bincount(ti % 8, minlength=8).tolist())
        print("  bad by t // 8 (first 20 blocks with any):", [(int(k), int(v)) for k, v in enumerate(np.bincount(ti // 8, minlength=65)) if v][:20])
        print("  bad by batch entry (first 10):", [(int(k), int(v)) for k, v in enumerate(np.bincount(bi, minlength=Bsz)) if v][:10])
        k = np.argmax(np.abs(a - b)); idx = np.unravel_index(k, a.shape)
        print("  worst", idx, a[idx], b[idx], c[idx])
    for k in ("du", "ddelta", "dz", "dA", "dA_b", "dD", "ddelta_bias"):
        e = (g0[k].float() - g1[k].float()).abs().max().item() / g0[k].float().abs().max().item()
        if e > 1e-3:
            print("  ", k, e)
    # which waves: partial rows [batch][L][nparts = 2 * groups][32] at the start of the workspace
    G = E // 64
    p0 = g0["_ws"][:Bsz * L * 2 * G * 32].view(Bsz, L, 2 * G, 32).cpu().numpy()
    p1 = g1["_ws"][:Bsz * L * 2 * G * 32].view(Bsz, L, 2 * G, 32).cpu().numpy()
    badp = np.abs(p0 - p1) > 0.05 * np.abs(p0).max()
    bi, ti, pi, ci = np.nonzero(badp)
    print("  partials: bad", badp.sum(), "of", badp.size)
    if badp.any():
        grp, d = pi // 2, pi % 2
        unit = bi * G + grp
        slot = unit % 3
        it = np.where(d == 0, ti, L - 1 - ti)
        fh = np.where(d == 0, L // 2, L - L // 2)
        phase = np.where(it >= fh, 1, 2)
        wave = np.where(slot == 0, d, np.where(slot == 2, 2 + d, np.where(phase == 1, 2 + d, d)))
        import collections
        print("  by (slot XYZ, dir, phase):", sorted(collections.Counter(zip(slot.tolist(), d.tolist(), phase.tolist())).items()))
        print("  by wave in workgroup:", sorted(collections.Counter(wave.tolist()).items()))
        print("  by wg parity / wg//256:", sorted(collections.Counter(((unit // 3) // 256).tolist()).items()))
        print("  by iteration mod 8:", np.bincount(it % 8, minlength=8).tolist(), " by column:", np.bincount(ci, minlength=32).tolist())
        blk = it // 8
        print("  by block (it//8) first 12:", [(int(k), int(v)) for k, v in enumerate(np.bincount(blk, minlength=65)) if v][:12], "...last:", [(int(k), int(v)) for k, v in enumerate(np.bincount(blk, minlength=65)) if v][-6:])
    if badp.any():
        k = 0
        b_, t_, p_, c_ = bi[k], ti[k], pi[k], ci[k]
        d_ = p_ % 2
        it_ = t_ if d_ == 0 else L - 1 - t_
        blk_ = it_ // 8
        its = np.arange(blk_ * 8, blk_ * 8 + 8)
        ts = its if d_ == 0 else L - 1 - its
        np.set_printoptions(precision=4, suppress=True, linewidth=250)
        print(f"  one bad wave: batch {b_} part {p_} (dir {d_}) block {blk_}: rows = iterations {its[0]}..{its[-1]}; dC columns (states 0..15)")
        print("  ref:\n", p0[b_, ts, p_, 16:]); print("  new:\n", p1[b_, ts, p_, 16:])
        print("  ref dB:\n", p0[b_, ts, p_, :16]); print("  new dB:\n", p1[b_, ts, p_, :16])
        # next block too
        its2 = its + 8
        if its2[-1] < L:
            ts2 = its2 if d_ == 0 else L - 1 - its2
            print("  next block ref dC:\n", p0[b_, ts2, p_, 16:]); print("  next block new dC:\n", p1[b_, ts2, p_, 16:])
