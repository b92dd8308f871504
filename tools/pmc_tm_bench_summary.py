#!/usr/bin/env python3
"""pmc_traffic_tm.json / valu_busy_tm.json from the counter passes of tools/pmc_tm_bench.sh: HBM bytes per launch (FETCH_SIZE doubled
for wide coalesced reads as MI355X_MICROARCH.md prescribes, WRITE_SIZE as counted; units KiB) and the vector ALU's busy share of the
token-major kernels at the bench shape (B = 64, E = 1536, L = 513, bf16; tools/tm_time.py launches).  Stamped with commit and date."""
import collections
import csv
import datetime
import json
import os
import sys

out = sys.argv[1]
NAMES = [("k_scant_bwd<aum::bf16_t, true, true, true>", "scan_tm_bwd_bidir"), ("k_scant_fwd<aum::bf16_t, true, true, true, true>", "scan_tm_fwd_bidir"),
         ("k_scant_bwd<aum::bf16_t, false, true, true>", "scan_tm_bwd_bidir_nosp"), ("k_scant_fwd<aum::bf16_t, false, true, true, true>", "scan_tm_fwd_bidir_nosp"),
         ("k_scant_fwd<aum::bf16_t, false, true, false, true>", "scan_tm_fwd_bidir_inference"),
         ("k_scant_bwd_reduce", "scan_tm_bwd_reduce"), ("k_convt_fwd<aum::bf16_t", "conv_tm_fwd"), ("k_convt_bwd<aum::bf16_t", "conv_tm_bwd"),
         ("k_xdt_tm_bwd<true", "xdt_tm_bwd"), ("k_xdt_tm_fwd<true", "xdt_tm_fwd"), ("k_gemm_wgrad_skinny<true, 3>", "gemm_wgrad_k48"),
         ("k_gemm_wgrad_skinny<true, 5>", "gemm_wgrad_k80"), ("k_gemm_wgrad<true>", "gemm_wgrad"), ("k_gemm_tn_persistent<true>", "gemm_tn")]
stamp = {"_commit": os.environ.get("AUM_COMMIT", "unknown"), "_date": datetime.date.today().isoformat()}
SRC = os.environ.get("AUM_PMC_SOURCE", "tools/tm_time.py")


def collect(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path, newline="")):
        for pat, key in NAMES:
            if pat in row["Kernel_Name"]:
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                break
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in acc.items()}


fetch, write, valu = collect(out + "/FETCH_SIZE.csv"), collect(out + "/WRITE_SIZE.csv"), collect(out + "/valu.csv")
tr = dict(stamp, _method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over " + SRC + " (B=64, E=1536, L=513, bf16, "
          "[x | z] rows); units KiB, mean over the launches of each kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a "
          "wide coalesced read: calibrated in round 2 with a 1 GiB copy, profiles/pmc_traffic.json hbm_copy)")
for k in sorted(set(fetch) | set(write)):
    rd, wr = 2 * 1024 * fetch.get(k, {}).get("FETCH_SIZE", 0.0), 1024 * write.get(k, {}).get("WRITE_SIZE", 0.0)
    tr[k] = int(rd + wr)
    tr[k + "_detail"] = {"read_bytes_corrected": int(rd), "write_bytes": int(wr)}
json.dump(tr, open(out + "/pmc_traffic_tm.json", "w"), indent=1)
vb = dict(stamp, _method="rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over " + SRC + "; "
          "busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)")
for k, m in sorted(valu.items()):
    a, i, g = m.get("SQ_ACTIVE_INST_VALU", 0.0), m.get("SQ_INSTS_VALU", 0.0), m.get("GRBM_GUI_ACTIVE", 0.0)
    if g and i:
        vb[k] = {"valu_busy_frac": round(4.0 * a / (1024.0 * g / 8.0), 3), "valu_insts_per_launch": int(i), "cycles_per_inst": round(4.0 * a / i, 2)}
json.dump(vb, open(out + "/valu_busy_tm.json", "w"), indent=1)
print(json.dumps({k: v for k, v in tr.items() if not k.endswith("_detail") and not k.startswith("_")}, indent=1))
print(json.dumps({k: v for k, v in vb.items() if not k.startswith("_")}, indent=1))
