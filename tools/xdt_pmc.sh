# counter passes (each --kernel-trace only) of aum_xdt_tm_fwd and aum_dtproj_tm_fwd at the bench shape: matrix pipe, LDS conflicts, HBM bytes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=/tmp/pmc_xdt; rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o pmc -- python tools/xdt_time.py > $OUT/$n.log 2>&1; find $OUT/$n -name "*counter_collection.csv" -exec cp {} $OUT/$n.csv \; ; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import collections, csv
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("mfma", "lds", "fetch", "write"):
    for row in csv.DictReader(open(f"/tmp/pmc_xdt/{f}.csv")):
        k = row["Kernel_Name"]
        if "k_xdt" in k or "k_dtproj" in k:
            acc[k[:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else float("nan")
print("%-50s %7s %10s %10s %10s %12s" % ("kernel", "MFMA%", "conflict%", "fetch_MB", "write_MB", "GUI_ACTIVE/8"))
for k, c in sorted(acc.items()):
    m, g = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["GRBM_GUI_ACTIVE"])
    print("%-50s %6.1f%% %9.2f%% %10.1f %10.1f %12.0f" % (k, 100 * m / (1024 * g / 8), 100 * mean(c["SQ_LDS_BANK_CONFLICT"]) / max(mean(c["SQ_LDS_IDX_ACTIVE"]), 1.0),
          mean(c["FETCH_SIZE"]) * 2048 / 1e6, mean(c["WRITE_SIZE"]) * 1024 / 1e6, g / 8))
PY
