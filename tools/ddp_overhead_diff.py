#!/usr/bin/env python3
"""Kernel-by-kernel difference of two rocprofv3 kernel traces of bench.py (rocpd sqlite): the plain step against the forced-DDP step
(AUM_BENCH_FORCE_DDP=1, world size 1).  Per kernel name: launches and total ms PER STEP in each run and the difference; then the time
the GPU was busy (union of kernel intervals), the time two kernels overlapped, and the idle time inside the span of the traced steps.
usage: ddp_overhead_diff.py <plain.db> <ddp.db> <steps> <skip> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:96]


def load(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    return [(short(n), s, e) for n, s, e in cur.execute(f"select {namec}, start, end from kernels order by start")]


def busy(rows):
    """union of intervals, sum of intervals, span -- in ms"""
    tot = sum(e - s for _, s, e in rows)
    union, cur_s, cur_e = 0, None, None
    for _, s, e in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    return union / 1e6, tot / 1e6, (rows[-1][2] - rows[0][1]) / 1e6


def tail_steps(rows, steps, skip, marker="k_frontend_tokens"):
    """the launches of `steps` steps in front of the last `skip` ones (bench.py ends with 1 + min(5, --steps) steps without the optimizer
    and the GEMM probe): a step starts at the frontend kernel (one launch per step)"""
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < steps + skip + 1:
        return rows, len(idx)
    return rows[idx[-1 - skip - steps]:idx[-1 - skip]], steps


def gaps(rows, steps, top=14):
    """idle intervals (no kernel running) by the kernels on either side, summed over the steps"""
    g, cur_e, prev = {}, None, None
    for n, s, e in rows:
        if cur_e is not None and s > cur_e:
            c = g.setdefault((prev, n), [0, 0])
            c[0] += 1
            c[1] += s - cur_e
        if cur_e is None or e > cur_e:
            cur_e, prev = e, n
    out = sorted(g.items(), key=lambda kv: -kv[1][1])[:top]
    return [f"    {c[1] / 1e6 / steps:7.3f} ms/step in {c[0] / steps:6.1f} gaps/step   {a[:52]:52s} -> {b[:52]}" for (a, b), c in out]


def table(rows, steps):
    t = {}
    for n, s, e in rows:
        c = t.setdefault(n, [0, 0])
        c[0] += 1
        c[1] += e - s
    return {n: (c[0] / steps, c[1] / 1e6 / steps) for n, c in t.items()}


def main():
    a, b, steps, skip = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    a, sa = tail_steps(a, steps, skip)
    b, sb = tail_steps(b, steps, skip)
    ta, tb = table(a, sa), table(b, sb)
    names = sorted(set(ta) | set(tb), key=lambda n: -abs(tb.get(n, (0, 0))[1] - ta.get(n, (0, 0))[1]))
    out = [f"per step, over {sa} / {sb} timed steps of each trace (step = frontend launch to frontend launch)",
           f"{'kernel':96s} {'n plain':>8s} {'ms plain':>9s} {'n ddp':>8s} {'ms ddp':>9s} {'d ms':>8s}"]
    for n in names[:45]:
        pa, pb = ta.get(n, (0, 0)), tb.get(n, (0, 0))
        if abs(pb[1] - pa[1]) < 0.004 and pa[0] == pb[0]:
            continue
        out.append(f"{n:96s} {pa[0]:8.1f} {pa[1]:9.3f} {pb[0]:8.1f} {pb[1]:9.3f} {pb[1] - pa[1]:+8.3f}")
    for label, rows, st in (("plain", a, sa), ("ddp", b, sb)):
        u, t, sp = busy(rows)
        out.append(f"{label:6s} span {sp / st:8.3f} ms/step   busy (union) {u / st:8.3f}   sum of kernels {t / st:8.3f}   "
                   f"overlapped {(t - u) / st:7.3f}   idle {(sp - u) / st:7.3f}   launches {len(rows) / st:7.1f}")
        out += gaps(rows, st)
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
