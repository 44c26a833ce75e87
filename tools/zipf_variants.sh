#!/bin/bash
# usage: tools/zipf_variants.sh lib1.so lib2.so ...   (A/B timing on the Zipf workload; GPU box only)
cp redux_amd/libredux_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so redux_amd/libredux_hip.so' EXIT
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  echo "$lib: $(timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-10} --no-cpu-baseline --no-decode --workload zipf 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
done
