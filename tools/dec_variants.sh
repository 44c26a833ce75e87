#!/bin/bash
# usage: tools/dec_variants.sh lib1.so lib2.so ...   (A/B timing of alternative decode builds; GPU box only)
# (a variant is loaded through REDUX_LIB: the in-tree product library is never overwritten)
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  echo "$lib: $(timeout -k 10 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode ${DEC_ARGS} 2>&1 | tail -1 | grep -o '"decode": {[^}]*}')"
done
