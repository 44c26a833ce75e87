#!/bin/bash
# usage: tools/run_variants.sh lib1.so lib2.so ...   (A/B timing of alternative builds; GPU box only)
cp redux_amd/libredux_hip.so /tmp/keep.so
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  echo "$lib: $(timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
done
cp /tmp/keep.so redux_amd/libredux_hip.so
