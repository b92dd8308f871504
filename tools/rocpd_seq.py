#!/usr/bin/env python3
"""One step of a rocprofv3 (rocpd sqlite) kernel trace as a time line: start offset, duration, gap to the previous kernel's end, name.
usage: rocpd_seq.py <results.db> <out.txt> [anchor-substring] [occurrence]   -- the slice starts at the n-th dispatch whose name contains the
anchor (default: k_frontend_tokens, the first kernel of a bench step; occurrence 4 = a timed step) and ends at the next one."""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")[:100]


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namec}, start, end from kernels order by start").fetchall()
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_frontend_tokens"
    occ = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    a, b = idx[occ], idx[occ + 1]
    t0 = rows[a][1]
    out, prev_end, gaps, busy = [], rows[a][1], 0, 0
    for n, s, e in rows[a:b]:
        gap = s - prev_end
        gaps += max(gap, 0)
        busy += e - s
        out.append(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {gap / 1e3:8.2f}  {short(n)}")
        prev_end = max(prev_end, e)
    hdr = f"# step slice: {b - a} dispatches, wall {(prev_end - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, positive gaps {gaps / 1e6:.3f} ms\n# start_us   dur_us   gap_us  kernel\n"
    open(sys.argv[2], "w").write(hdr + "\n".join(out) + "\n")
    print(hdr)


if __name__ == "__main__":
    main()
