#!/bin/bash
# usage: tools/run_variants.sh lib1.so lib2.so ...   (A/B timing of alternative builds; GPU box only)
# STEPS / WARMUP env vars: the chip's clock settles only after a few hundred ms of load, so use
# a long warm-up when the difference to resolve is below ~3 %.
cp redux_amd/libredux_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so redux_amd/libredux_hip.so' EXIT  # an interrupted run must not leave a variant build as the product library
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  echo "$lib: $(timeout -k 10 200 python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-1} --no-cpu-baseline --no-decode 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
done
cp /tmp/keep.so redux_amd/libredux_hip.so
