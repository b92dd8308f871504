#!/usr/bin/env python3
"""Per-kernel means of the counters tools/pmc_tm.sh collected, with the derived VALU occupancy and wait shares."""
import collections
import csv
import glob
import sys

csv.field_size_limit(10 ** 9)
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_scan" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(out + "/p1/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "k_scan" in row["Kernel_Name"]:
            dur[row["Kernel_Name"][:100]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    g = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    d = dur.get(k, [0.0])
    print(k, "calls=%d" % len(c.get("SQ_INSTS_VALU", [])), "kernel_us(mean under the counter pass)=%.1f" % (sum(d) / len(d)))
    print("   ", "  ".join("%s=%.4g" % (n, v) for n, v in sorted(m.items())))
    if g and m.get("SQ_INSTS_VALU"):
        print("    valu_busy=%.1f%%  quadcycles_per_valu=%.2f  kernel_cycles=%.0f  valu_per_wave=%.0f  lds_per_wave=%.0f" % (
            100.0 * 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * g), m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"], g, m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
            m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"]))
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        print("    of wave cycles: wait_any=%.1f%% wait_inst=%.1f%% (lds %.1f%%) active_any=%.1f%% active_lds=%.1f%%   lds_idx_active=%.4g bank_conflict=%.4g" % (
            100 * m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_INST_ANY"] / w, 100 * m.get("SQ_WAIT_INST_LDS", 0) / w, 100 * m["SQ_ACTIVE_INST_ANY"] / w,
            100 * m.get("SQ_ACTIVE_INST_LDS", 0) / w, m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)))
