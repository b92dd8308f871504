#!/usr/bin/env python3
"""Adds the (operation, shape) -> solution lines of TunableOp result files (tools/tune_job.sh) that the recorded file lacks.
python tools/merge_tunable.py gpurun_out/tunable_*.csv [--replace]     (--replace: a new line also replaces a recorded one for the same shape)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "audio-mamba-aum_amd", "aum", "tunableop_gfx950.csv")
replace = "--replace" in sys.argv
lines = open(DST).read().splitlines()
validators = [ln for ln in lines if ln.startswith("Validator,")]
have = {tuple(ln.split(",")[:2]): i for i, ln in enumerate(lines) if not ln.startswith("Validator,")}
added = replaced = 0
for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
    src = open(path).read().splitlines()
    if [ln for ln in src if ln.startswith("Validator,")] != validators:
        sys.exit(f"{path}: validators differ from the recorded file's -- re-record the whole file instead")
    for ln in src:
        if ln.startswith("Validator,") or not ln.strip():
            continue
        key = tuple(ln.split(",")[:2])
        if key not in have:
            have[key] = len(lines)
            lines.append(ln)
            added += 1
        elif replace and lines[have[key]] != ln:
            lines[have[key]] = ln
            replaced += 1
open(DST, "w").write("\n".join(lines) + "\n")
print(f"{DST}: {added} added, {replaced} replaced, {len(lines) - len(validators)} shapes")
