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
and printed per loop with the per-(state, step) figure for the backward (16 state-steps per pass).

  python tools/isa_mix_tm.py --marks /tmp/p6m.s 'k_scant_bwd.*Lb1ELb1ELb1E'        (listing built with -DAUM_SCANT_MARKS=1)
itemises the WHOLE 8-step block of the backward, not only its pass loop (VERDICT r5 #2a): the kernel's stage stamps become comment lines in
the listing, every basic block inherits the stage it is entered in along the control-flow graph, and the instructions are counted per
stage and role.  Stages inside the pass loop run eight times per block, the others once: the last columns give vector-ALU instructions per
block and per (state, step) (128 per block), which is what the SQ_INSTS_VALU counter sees."""
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


STAGES = {0: "phase prologue (A, first block's requests)", 9: "wait for the block's tensors; B / C pairs -> fp32 rows",
          10: "per-step values: unpack, softplus, gate, dy, dz (P, Q)", 1: "S1 / S2 = 0; requests of the next block / partials",
          2: "PASS  A, entry state, B / C of the pair from LDS", 3: "PASS  exponentials + forward sweep (x, w, dC terms)", 4: "PASS  dC butterfly, head",
          5: "PASS  reverse sweep (g, dB terms, S1, S2, dA, h)", 6: "PASS  dB head + both tails, dB | dC slots, carries", 7: "du, ddelta (+ partials, softplus'), dD, dbias -> tiles",
          8: "stores: du, ddelta, dz, dB | dC rows"}
# which stamp opens the stage that ENDS at stamp k is fixed by the source order: 9 10 1 [2 3 4 5 6]x8 7 8; a stamp names the stage that ends at it
ORDER = [9, 10, 1, 2, 3, 4, 5, 6, 7, 8]


def marks_main(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^(_Z\w+):", ln) and pat.search(ln))
    print(lines[start].split(":")[0])
    # basic blocks in listing order
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        t = ln.strip()
        mk = re.match(r"; AUM_MARK (\d+)", t)
        if mk:
            blocks[cur].append(("mark", int(mk.group(1))))
        elif t and not t.startswith((";", ".", "//")):
            blocks[cur].append(("ins", t))
    # control-flow edges
    succ = {b: [] for b in order}
    for i, b in enumerate(order):
        ins = [t for k, t in blocks[b] if k == "ins"]
        fall = True
        for t in ins:
            op = t.split()[0]
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = t.split()[1]
                if tgt in succ:
                    succ[b].append(tgt)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                fall = op != "s_branch" and False
        last = ins[-1].split()[0] if ins else ""
        if last not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < len(order):
            succ[b].append(order[i + 1])
    # the stage a block is entered in: propagate the last stamp seen along the edges (a stamp k means "stage k has just ended")
    entry = {b: None for b in order}
    exit_ = {}
    def block_exit(b):
        last = entry[b]
        for k, t in blocks[b]:
            if k == "mark":
                last = t
        return last
    changed = True
    while changed:
        changed = False
        for b in order:
            e = block_exit(b)
            for s in succ[b]:
                if entry[s] is None and e is not None:
                    entry[s] = e
                    changed = True
    # group = instantiation of the block body (phase x FULL / ragged), by listing position: a new one starts at every stamp 9
    group_of, g = {}, -1
    for b in order:
        group_of[b] = g
        for k, t in blocks[b]:
            if k == "mark" and t == 9:
                g += 1
        # a block that contains stamp 9 belongs, from the stamp on, to the new group: handled per instruction below
    keys = ["exp", "pk-arith", "dpp", "cvt", "mov", "valu-other", "lds", "vmem", "salu", "nop", "wait"]
    counts = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))     # group -> stage (the stamp that ENDS it) -> role
    nxt = {ORDER[i]: ORDER[i + 1] for i in range(len(ORDER) - 1)}
    nxt[6] = 2          # inside the pass loop; the epilogue (stage 7) is entered from the loop exit, where the last stamp seen is 6
    g = -1
    for b in order:
        last, gg = entry[b], group_of[b]
        for k, t in blocks[b]:
            if k == "mark":
                last = t
                if t == 9:
                    g += 1
                    gg = g
                continue
            if last is None or gg < 0 and last not in ORDER:
                continue
            counts[gg][last][role(t)] += 1
    for gg in sorted(counts):
        if gg < 0:
            continue
        print(f"\n## block body {gg} (phase {'1 (first half of a direction pair)' if gg < 2 else '2 (second half: finishes du / ddelta / dz)'}, "
              f"{'as laid out first' if gg % 2 == 0 else 'as laid out second'})")
        print("code after stamp".ljust(64) + "".join(k.rjust(11) for k in keys) + "   VALU  x/blk  VALU/(state,step)")
        tot = 0.0
        for st in [9, 10, 1, 2, 3, 4, 5, 6, 7, 8]:
            # instructions counted under "last stamp = st" are the stage that FOLLOWS stamp st
            c = counts[gg].get(st)
            if not c:
                continue
            follow = {9: 10, 10: 1, 1: 2, 2: 3, 3: 4, 4: 5, 5: 6, 6: "6+", 7: 8, 8: 9}[st]
            name = {10: STAGES[10], 1: STAGES[1], 2: STAGES[2] + " (+ loop entry)", 3: STAGES[3], 4: STAGES[4], 5: STAGES[5], 6: STAGES[6],
                    "6+": "PASS  loop control / after the last pass: " + STAGES[7], 8: STAGES[8], 9: STAGES[9]}[follow]
            nexp_arg = c["exp"] // 2 if follow == 3 else 0
            c = collections.Counter(c)
            c["exp"] += nexp_arg
            c["pk-arith"] -= nexp_arg
            valu = sum(c[k] for k in ("exp", "pk-arith", "dpp", "cvt", "mov", "valu-other"))
            reps = 8 if st in (1, 2, 3, 4, 5) and follow != 2 or follow in (3, 4, 5, 6) else 1
            if follow == 2:
                reps = 1        # the code after stamp 1 is the request stage plus the first pass's loop entry: counted once
            tot += valu * reps / 128.0
            print(f"{st:2d} -> {name[:58]:58s}" + "".join(str(c[k]).rjust(11) for k in keys) + f"{valu:7d}{reps:7d}{valu * reps / 128.0:10.2f}")
        print(" " * 64 + " " * (11 * len(keys)) + f"   total VALU / (state, step): {tot:.2f}")


def main():
    if sys.argv[1] == "--marks":
        return marks_main(sys.argv[2], re.compile(sys.argv[3]))
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
