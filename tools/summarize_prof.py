#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:90]
        print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (name, row.get("Calls"), row.get("TotalDurationNs"),
                                                              row.get("AverageNs"), row.get("Percentage")))
import json
pmc_json = {}
for d in find("pmc*/"):
    for f in find(os.path.relpath(d, out) + "/**/*counter_collection.csv"):
        print("== counters (%s)" % os.path.relpath(f, out))
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:70]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        for k, ctrs in acc.items():
            if not any(tag in k for tag in ("kws::", "gru_", "octbit", "mel_frontend", "window_step", "vad_kernel")):
                continue
            print(k)
            for name, v in sorted(ctrs.items()):
                n = cnt[(k, name)]
                print("    %-28s per-dispatch %.6g  (dispatches %d)" % (name, v / n, n))
                pmc_json.setdefault(k, {})[name] = v / n
# HBM traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB-units of
# 1024 B; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2.
for k, c in pmc_json.items():
    if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["SQ_WAVES"] > 0:
        # SQ_WAVE_CYCLES counts quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs
        # (one wave per SIMD in the resident kernels): the matrix pipe's share of the waves' lifetime
        c["mfma_busy_frac_of_wave_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"])
        c["wave_parked_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        # SQ_ACTIVE_INST_VALU: quad-cycles in which a SIMD's VALU was executing, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE:
        # busy cycles summed over the 8 XCDs -> the share of all SIMD-cycles of the launch with the VALU busy (the binding
        # number of an issue-bound kernel such as the FFT front-end; the same ratio for the matrix pipe beside it)
        simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
        c["valu_busy_frac_of_simd_cycles"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            c["mfma_busy_frac_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
        c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        c["hbm_bytes_per_launch"] = c["hbm_read_bytes_corrected"] + c["hbm_write_bytes"]
json.dump(pmc_json, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
