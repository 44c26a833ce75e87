#!/bin/bash
# usage: tools/ab/dec_traffic_ab.sh lib1.so ...  -> decode ms (bench.py) + FETCH_SIZE / WRITE_SIZE of k_decode_lock per build (GPU box; REDUX_LIB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
A="bench.py --steps 1 --warmup 0 --no-cpu-baseline --decode"
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib); tag=$(basename $lib .so)
  echo "== $lib: $(timeout -k 10 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 2>&1 | tail -1 | grep -o '"decode": {"kernel": "[^"]*", "ms": [0-9.]*')"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/dab_$tag/f -- python3 $A > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/dab_$tag/w -- python3 $A > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/dab_$tag k_decode_lock | grep -A2 "grid of 65536" | grep SIZE
done
