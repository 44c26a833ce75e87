#!/bin/bash
# FETCH_SIZE (L2 <- fabric reads) of the encode kernel for alternative builds, one PMC pass each.  Usage: tools/fetch_ab.sh a.so b.so ...
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  OUT=gpurun_out/fetch_ab/$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $OUT/log.txt 2>&1
  echo "$lib: $(python3 tools/pmc_summary.py $OUT k_encode_pair | grep FETCH_SIZE)"
done
