#!/usr/bin/env python3
"""Instruction mix of the token-major scan kernels' inner loops by ROLE ("account for every VALU slot").
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAUM_API_PART=6 -DAUM_DTYPE_ONLY=1 [-DAUM_SCANT_MSUM=0] \
        --cuda-device-only -S audio-mamba-aum_amd/csrc/aum_hip.hip -o /tmp/p6.s        (part 5 = the forward kernels)
  python tools/isa_mix_tm.py /tmp/p6.s 'k_scant_bwd.*Lb1ELb1ELb1E'
For every loop body at the deepest nesting level (the pass over one state pair of the backward: 8 steps x 2 states; the unrolled 8-step
block of the forward) the instructions are grouped as
  exp        v_exp_f32 and the packed multiply that forms its argument (delta * A log2 e)
  pk-arith   the other packed fp32 multiply / fma / add: recurrence, adjoint, dB / dC products, S1 / S2 / dA sums
  dpp        DPP adds + permlane swaps of the channel sums
  cvt        v_cvt_pk_bf16_f32 (terms of the matrix-pipe sums), 16-bit unpack shifts / ands
  mfma       v_mfma (matrix pipe: not a vector-ALU slot)
  mov        v_mov (register-indexed carries through s_set_gpr_idx, operand copies for the in-place DPP levels)
  valu-other address arithmetic, selects, scalar-operand fp32 ops
  lds / vmem / salu / nop / wait
and printed per loop with the per-(state, step) figure for the backward (16 state-steps per pass)."""
import collections
import re
import sys


def role(t):
    op = t.split()[0]
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp"):
        return "exp"
    if op.startswith("v_mfma"):
        return "mfma"
    if "dpp" in t or "row_" in t or "quad_perm" in t or op.startswith("v_permlane"):
        return "dpp"
    if op.startswith("v_cvt_pk") or op in ("v_lshlrev_b32_e32", "v_and_b32_e32") and ("16," in t or "0xffff0000" in t):
        return "cvt"
    if op.startswith("v_pk_"):
        return "pk-arith"
    if op.startswith("v_mov"):
        return "mov"
    if op.startswith("v_"):
        return "valu-other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_")):
        return "vmem"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    denom = int(sys.argv[3]) if len(sys.argv) > 3 else 16        # state-steps one trip of the loop covers (backward pass: 16; forward block: 128)
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^(_Z\w+):", ln) and pat.search(ln))
    print(lines[start].split(":")[0])
    blocks, depth, cur = collections.OrderedDict(), {}, None
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):(.*)", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            d = re.findall(r"Depth=(\d+)", m.group(2))
            depth[cur] = max(map(int, d)) if d else 0
            continue
        t = ln.strip()
        if cur and not blocks[cur] and t.startswith(";"):        # the comment lines under a loop header's label carry its own depth
            d = re.findall(r"Depth=(\d+)", t)
            if d:
                depth[cur] = max(depth[cur], max(map(int, d)))
        if cur and t and not t.startswith((";", ".", "//")):
            blocks[cur].append(t)
    for ln in lines[start:]:
        if re.search(r"; (NumVgprs|ScratchSize|Occupancy):", ln):
            print("  ", ln.strip())
        if "; -- End function" in ln:
            break
    deepest = max(depth.values())
    # merge the blocks of one innermost loop (consecutive labels at the deepest level)
    loops, run = [], []
    for name, body in blocks.items():
        if depth[name] == deepest:
            run.append(name)
        elif run:
            loops.append(run)
            run = []
    if run:
        loops.append(run)
    keys = ["exp", "pk-arith", "dpp", "cvt", "mfma", "mov", "valu-other", "lds", "vmem", "salu", "nop", "wait"]
    print("loop (blocks)".ljust(34) + "".join(k.rjust(11) for k in keys) + "   VALU   all")
    for run in loops:
        c = collections.Counter()
        for name in run:
            for t in blocks[name]:
                c[role(t)] += 1
        # the packed multiplies that feed v_exp: one per two exponentials
        nexp_arg = c["exp"] // 2
        c["exp"] += nexp_arg
        c["pk-arith"] -= nexp_arg
        valu = sum(c[k] for k in ("exp", "pk-arith", "dpp", "cvt", "mov", "valu-other"))
        if valu < 60:
            continue
        print((run[0] + ".." + run[-1]).ljust(34) + "".join(str(c[k]).rjust(11) for k in keys) + f"{valu:7d}{sum(c.values()):6d}")
        print((" per (state, step) of %d" % denom).ljust(34) + "".join(("%.2f" % (c[k] / denom)).rjust(11) for k in keys) + f"{valu / denom:7.2f}")


if __name__ == "__main__":
    main()
