#!/bin/bash
# usage: tools/zipf_variants.sh lib1.so lib2.so ...   (A/B timing on the Zipf workload; GPU box only)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  echo "$lib: $(timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-10} --no-cpu-baseline --no-decode --workload zipf 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
done
