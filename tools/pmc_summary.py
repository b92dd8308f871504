#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv passes (tools/pmc_job.sh): HBM bytes per
launch of the hand-written kernels.  Units KiB; FETCH_SIZE is doubled for wide coalesced reads as
MI355X_MICROARCH.md prescribes (calibrated on this box with a 1 GiB float4 copy: FETCH_SIZE reads 0.50 GiB)."""
import csv
import json
import sys
from collections import defaultdict

NAMES = [("k_scanh_bwd<aum::bf16_t, 1, 2>", "scan_bwd_bidir"), ("k_scanh_bwd<aum::bf16_t, 1, 0>", "scan_bwd"),
         ("k_scanr_fwd<aum::bf16_t, 2>", "scan_fwd_bidir"), ("k_scanr_fwd<aum::bf16_t, 0>", "scan_fwd"),
         ("k_frontend_tokens<", "frontend_tokens"), ("k_fbank_w", "fbank_fwd"), ("k_sum_rows<", "sum_rows"), ("k_scan_reduce", "scan_reduce"),
         ("k_conv4_rows_fwd<", "conv_fwd"), ("k_conv4_rows_bwd<", "conv_bwd"),
         ("k_scanwg_bwd<aum::bf16_t, 8, 1, 2>", "scan_bwd_bidir_rowpair"), ("k_scanwg_fwd<aum::bf16_t, 8, 1, 2>", "scan_fwd_bidir"),
         ("k_scanwg_bwd<aum::bf16_t, 8, 1, 0>", "scan_bwd_rowpair"), ("k_scanwg_fwd<aum::bf16_t, 8, 1, 0>", "scan_fwd"),
         ("k_proj_fwd<", "proj_fwd"), ("k_proj_bwd_data<", "proj_bwd_data"), ("k_proj_bwd_weight<", "proj_bwd_weight"),
         ("k_conv4_fwd<", "conv_fwd"), ("k_conv4_bwd<", "conv_bwd"), ("k_norm_fwd_vec<", "rmsnorm_fwd"), ("k_norm_bwd_vec<", "rmsnorm_bwd"),
         ("k_hbm_copy", "hbm_copy")]


def collect(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for pat, key in NAMES:
                if pat in row["Kernel_Name"]:
                    acc[key].append(float(row["Counter_Value"]))
                    break
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write, out = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE"), {}
    out["_method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/kbench.py (B=64, E=1536, L=513, "
                      "bf16); units KiB, mean over the launches of each kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                      "reports 1/2 of a wide coalesced read; the hbm_copy entry is the calibration: 1 GiB read + 1 GiB written); "
                      "WRITE_SIZE of element-aligned 16-byte stores (rows of 513 elements) reads about 2x the stored bytes")
    for key in sorted(set(fetch) | set(write)):
        rd, wr = 2 * 1024 * fetch.get(key, 0.0), 1024 * write.get(key, 0.0)
        out[key] = int(rd + wr)
        out[key + "_detail"] = {"FETCH_SIZE_KiB": round(fetch.get(key, 0.0)), "WRITE_SIZE_KiB": round(write.get(key, 0.0)),
                                "read_bytes_corrected": int(rd), "write_bytes": int(wr)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.endswith("_detail") and k != "_method"}, indent=1))


if __name__ == "__main__":
    main()
