#!/bin/bash
# usage: tools/kernel_ab.sh lib1.so ...  -> rocprofv3 average kernel times per build (GPU box)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  tag=$(basename $lib .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kab_$tag -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== $lib"; python3 tools/pmc_summary.py gpurun_out/kab_$tag | grep "k_compact\|k_encode\|k_scan" | cut -c1-130
done
