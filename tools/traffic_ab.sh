#!/bin/bash
# usage: tools/traffic_ab.sh lib1.so lib2.so ...  -> kernel ms + FETCH/WRITE_SIZE of the encode kernel per build (GPU box)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  tag=$(basename $lib .so)
  echo "== $lib: $(timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o 'kernel_ms[^}]*')"
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/tab_$tag/f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/tab_$tag/w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/tab_$tag k_encode | grep "SIZE"
done
