#!/bin/bash
# Vector-L1 side of the pair kernel (the mask table goes through it): TA / TCP counters of the default bench workload.
# Usage: tools/prof_l1.sh <tag> -> gpurun_out/l1_<tag>/
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/l1_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode"
timeout -k 10 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || true
timeout -k 10 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || true
timeout -k 10 150 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || true
python3 tools/pmc_summary.py $OUT k_encode_pair 2>&1 | tail -20
