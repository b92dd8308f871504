#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total, average, share.
usage: rocpd_stats.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namec}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {namec} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}"]
    for n, c, s, a, mn, mx in rows[:60]:
        lines.append(f"{short(n):110s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:5.1f}%")
    lines.append(f"TOTAL kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
