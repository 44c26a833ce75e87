#!/bin/bash
# usage: tools/wr_ab.sh lib1.so ...  -> L2 (TCC) write / eviction / fabric-write mix of the encode kernel per build (GPU box)
# (counter names from `rocprofv3 --list-avail`: an unknown name aborts this rocprofv3 with a core dump)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
A="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode"
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  tag=$(basename $lib .so)
  timeout -k 10 100 rocprofv3 --pmc TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum --output-format csv -d gpurun_out/wr_$tag/a -- python3 $A > gpurun_out/wr_$tag.log 2>&1
  timeout -k 10 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum --output-format csv -d gpurun_out/wr_$tag/c -- python3 $A >> gpurun_out/wr_$tag.log 2>&1
  timeout -k 10 100 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_SECTORS_sum --output-format csv -d gpurun_out/wr_$tag/b -- python3 $A >> gpurun_out/wr_$tag.log 2>&1
  echo "== $lib"; python3 tools/pmc_summary.py gpurun_out/wr_$tag k_encode | grep "TCC"
done
