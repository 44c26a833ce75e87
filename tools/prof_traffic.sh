#!/bin/bash
# HBM traffic counters only (two PMC passes).  Usage: [WORKLOAD=zipf] tools/prof_traffic.sh <tag>
TAG=${1:-x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --workload ${WORKLOAD:-iid}"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1 || true
python3 tools/pmc_summary.py $OUT
python3 tools/profile_json.py --encode $OUT --workload ${WORKLOAD:-iid}
