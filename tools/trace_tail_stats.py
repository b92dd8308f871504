#!/usr/bin/env python3
"""Per-kernel totals of the STEADY part of a rocprofv3 kernel trace: everything after the last TunableOp benchmark launch
(at::cuda::flush_icache_kernel -- shapes missing from aum/tunableop_gfx950.csv are tuned during warm-up and would drown the step).
  python tools/trace_tail_stats.py <dir with *_kernel_trace.csv> <timed steps> [top]"""
import collections
import csv
import glob
import os
import sys

d, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
last = max((int(r["End_Timestamp"]) for r in rows if "flush_icache" in r["Kernel_Name"]), default=0)
tail = [r for r in rows if int(r["Start_Timestamp"]) > last]
# the timed steps are the last `steps` of (warm + steps): keep the trailing steps/(all steps after tuning) share by time markers is
# not possible without markers, so report per-step averages over everything after tuning
agg = collections.defaultdict(lambda: [0, 0])
for r in tail:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) if tail else 0
print(f"kernels after tuning: {len(tail)} launches, busy {tot / 1e6:.2f} ms over a span of {span / 1e6:.2f} ms; per step (/{steps}): busy {tot / 1e6 / steps:.2f} ms")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e3 / steps:10.1f} us/step  {100 * t / tot:5.1f} %  calls/step {n / steps:7.1f}  avg {t / n / 1e3:8.1f} us  {name[:150]}")
