#!/bin/bash
TAG=${1:-x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --decode"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || true
python3 tools/pmc_summary.py $OUT k_decode
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || true
python3 tools/pmc_summary.py $OUT/pmc3 k_decode
# L2 <-> fabric traffic of the decoder: one pass per counter (the TCC block cannot hold both)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc5 -- python3 $ARGS > $OUT/pmc5.log 2>&1 || true
# what the wave waits for: the SQ's wait and level counters
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc6 -- python3 $ARGS > $OUT/pmc6.log 2>&1 || true
# read requests by size class: FETCH_SIZE's expression prices 128-byte requests through TCC_BUBBLE, which gfx950 does not count
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc7 -- python3 $ARGS > $OUT/pmc7.log 2>&1 || true
for i in 4 5 6 7; do python3 tools/pmc_summary.py $OUT/pmc$i k_decode; done
