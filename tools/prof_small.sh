#!/bin/bash
# Kernel trace of tools/small_grid.py (the small-launch path: k_coop_model, k_coop_chain, scan, compaction).  GPU box only.
# Usage: tools/prof_small.sh <tag>   -> gpurun_out/prof_small_<tag>/
set -e
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_small_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf $OUT/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/small_grid.py > $OUT/trace.log 2>&1 || true
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/per_launch.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per kernel name and grid size: median duration
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    d[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(d.items()):
    v.sort()
    print(f"{name[:60]:60s} grid {grid:>8s}  n={len(v):3d}  median {v[len(v)//2]:10.1f} us")
PY
cat $OUT/per_launch.txt
