# Counter passes of the projection GEMM kernels on the in_proj forward shape (run through gpurun; counter passes use --kernel-trace only, as
# the pool's rocprofv3 policy requires): matrix-pipe occupancy, LDS bank conflicts, HBM traffic -- aum_gemm_tn's schedules next to the
# library GEMM on the same operands.     -> gpurun_out/pmc_gemm/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
SHAPE=${1:-in_proj_fwd}
run() {   # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o pmc -- python tools/gemm_abl_probe.py --shapes $SHAPE --rounds 2 --iters 5 > $OUT/$n.log 2>&1
  find $OUT/$n -name "*counter_collection.csv" -exec cp {} $OUT/$n.csv \;
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python - "$OUT" "$SHAPE" <<'PY' | tee $OUT/summary.txt
import collections, csv, sys
out, shape = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("mfma", "lds", "fetch", "write"):
    try:
        for row in csv.DictReader(open(f"{out}/{f}.csv")):
            k = row["Kernel_Name"]
            if "gemm_tn" in k or (k.startswith("Cijk") or k.startswith("Custom_Cijk")) and "MT256" in k:
                acc[k[:64]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    except FileNotFoundError:
        pass
print(f"# tools/gemm_pmc.sh {shape}: rocprofv3 --pmc passes over tools/gemm_abl_probe.py --shapes {shape} (bf16, 32832 tokens); means over launches")
print("# MFMA% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); conflict% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;")
print("# HBM MB per launch: FETCH_SIZE x 2 KiB (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md), WRITE_SIZE x 1 KiB; separate passes")
print("%-66s %6s %7s %10s %10s %10s" % ("kernel", "calls", "MFMA%", "conflict%", "fetch_MB", "write_MB"))
mean = lambda v: sum(v) / len(v) if v else float("nan")
for k, c in sorted(acc.items()):
    m, g = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(c.get("GRBM_GUI_ACTIVE", []))
    bc, ia = mean(c.get("SQ_LDS_BANK_CONFLICT", [])), mean(c.get("SQ_LDS_IDX_ACTIVE", []))
    print("%-66s %6d %6.1f%% %9.2f%% %10.1f %10.1f" % (k, len(c.get("GRBM_GUI_ACTIVE", c.get("FETCH_SIZE", []))), 100 * m / (1024 * g / 8) if g == g and g else float("nan"),
          100 * bc / ia if ia == ia and ia else float("nan"), mean(c.get("FETCH_SIZE", [])) * 2048 / 1e6, mean(c.get("WRITE_SIZE", [])) * 1024 / 1e6))
PY
