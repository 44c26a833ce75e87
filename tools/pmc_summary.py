#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel stats + per-kernel mean of each PMC counter).
Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_encode"
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"),
              row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    if rows:
        # the launches of the full shape = the largest grid; the others (cpu_baseline sample, small_launch) are listed apart
        gmax = max(int(r["Grid_Size_X"]) for r in rows)
        dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        small = [dur(r) for r in rows if int(r["Grid_Size_X"]) < gmax]
        d = [dur(r) for r in rows if int(r["Grid_Size_X"]) == gmax]
        if small:
            print("   (%d launch(es) on a smaller input left out of the means: %s ms)" % (len(small), " ".join("%.2f" % x for x in small)))
        k = min(10, len(d))
        print("== %s: %d launches matching %s, ms each: %s" % (os.path.relpath(f, root), len(d), sub, " ".join("%.2f" % x for x in d)))
        print("   mean of all %.3f ms; mean of the last %d (the timed steps of the default bench) %.3f ms" % (sum(d) / len(d), k, sum(d[-k:]) / k))
# counters per (kernel, grid size): bench.py also launches the kernels on smaller inputs (its cpu_baseline sample, its
# 62-block small_launch), and a mean over launches of different sizes says nothing
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if sub in row.get("Kernel_Name", ""):
            acc[int(row.get("Grid_Size") or 0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for grid in sorted(acc, reverse=True):
    print("== counters for kernels matching %s, grid of %d threads%s" % (sub, grid, "" if grid == max(acc) else "  (a smaller launch of the same kernel)"))
    for k in sorted(acc[grid]):
        v = acc[grid][k]
        print("  %-24s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
