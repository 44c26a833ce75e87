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
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    if d:
        med = sorted(d)[len(d) // 2]
        small = [x for x in d if x < med / 2]  # bench.py's cpu_baseline leg launches the kernel once on a 256 MiB sample
        d = [x for x in d if x >= med / 2]
        if small:
            print("   (%d launch(es) on a smaller input left out of the means: %s ms)" % (len(small), " ".join("%.2f" % x for x in small)))
        k = min(10, len(d))
        print("== %s: %d launches matching %s, ms each: %s" % (os.path.relpath(f, root), len(d), sub, " ".join("%.2f" % x for x in d)))
        print("   mean of all %.3f ms; mean of the last %d (the timed steps of the default bench) %.3f ms" % (sum(d) / len(d), k, sum(d[-k:]) / k))
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if sub in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== counters for kernels matching", sub)
for k in sorted(acc):
    v = acc[k]
    print("  %-24s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
