#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc `-S --cuda-device-only` listing (VALU-bound kernels: the count of
vector-ALU issue slots in the state loop is the time model).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAUM_API_PART=2 -DAUM_DTYPE_ONLY=1 \
        --cuda-device-only -S audio-mamba-aum_amd/csrc/aum_hip.hip -o /tmp/p2.s
  python tools/isa_count.py /tmp/p2.s k_scanh_bwd.*Li1ELi2E
Prints per basic block (label) the number of VALU / transcendental / DPP / LDS / VMEM / SALU / scratch instructions, the
largest blocks first, plus the kernel's register and scratch footprint."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_waitcnt", "s_barrier")):
        return "sync"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and pat.search(m.group(1)):
            start, name = i, m.group(1)
            break
    if start is None:
        raise SystemExit("kernel not found")
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = collections.Counter()
    total = collections.Counter()
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end") or ln.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        blocks[cur][c] += 1
        total[c] += 1
        if "dpp" in t or "row_" in t or "quad_perm" in t:
            blocks[cur]["(dpp)"] += 1
            total["(dpp)"] += 1
        if op.startswith("v_pk_"):
            blocks[cur]["(pk)"] += 1
            total["(pk)"] += 1
    print(name)
    for ln in lines[start:]:
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)", ln):
            print("  ", ln.strip())
        if ln.startswith(".Lfunc_end"):
            pass
        if "; -- End function" in ln:
            break
    keys = ["valu", "(pk)", "(dpp)", "trans", "lane", "lds", "vmem", "scratch", "salu", "nop", "sync"]
    print("block".ljust(14) + "".join(k.rjust(9) for k in keys))
    big = sorted(blocks.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if not k.startswith("(")))[:12]
    for lab, c in big:
        print(lab.ljust(14) + "".join(str(c.get(k, 0)).rjust(9) for k in keys))
    print("TOTAL".ljust(14) + "".join(str(total.get(k, 0)).rjust(9) for k in keys))


if __name__ == "__main__":
    main()
