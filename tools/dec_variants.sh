#!/bin/bash
# usage: tools/dec_variants.sh lib1.so lib2.so ...   (A/B timing of alternative decode builds; GPU box only)
cp redux_amd/libredux_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so redux_amd/libredux_hip.so' EXIT  # an interrupted run must not leave a variant build as the product library
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  echo "$lib: $(timeout -k 10 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode ${DEC_ARGS} 2>&1 | tail -1 | grep -o '"decode": {[^}]*}')"
done
cp /tmp/keep.so redux_amd/libredux_hip.so
