#!/bin/bash
# usage: tools/run_variants.sh lib1.so lib2.so ...   (A/B timing of alternative builds; GPU box only)
# STEPS / WARMUP env vars: the chip's clock settles only after a few hundred ms of load, so use
# a long warm-up when the difference to resolve is below ~3 %.
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  echo "$lib: $(timeout -k 10 200 python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-1} --no-cpu-baseline --no-decode 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
done
