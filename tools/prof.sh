#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + PMC passes (each in its own run, as the
# MI355X guide prescribes).  Usage: tools/prof.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
# the kernel trace is taken over the DEFAULT bench command (10 warm-up + 10 timed steps), so that its
# per-kernel average can be held against the bench line's live HIP-event figure
rm -rf $OUT/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py > $OUT/trace.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc5 -- python3 $ARGS > $OUT/pmc5.log 2>&1 || true
find $OUT -name "*.csv" | head -30
