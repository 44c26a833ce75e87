#!/bin/bash
# kernel time + FETCH/WRITE counters for each build variant (GPU box only)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  t=$(timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o 'kernel_ms[^}]*')
  tr=$(tools/prof_traffic.sh m_$(basename $lib .so) 2>&1 | grep -E "FETCH|WRITE" | tr '\n' ' ')
  echo "$lib: $t | $tr"
done
