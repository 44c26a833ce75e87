#!/bin/bash
# kernel time + FETCH/WRITE counters for each build variant (GPU box only)
cp redux_amd/libredux_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so redux_amd/libredux_hip.so' EXIT  # an interrupted run must not leave a variant build as the product library
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  t=$(timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o 'kernel_ms[^}]*')
  tr=$(tools/prof_traffic.sh m_$(basename $lib .so) 2>&1 | grep -E "FETCH|WRITE" | tr '\n' ' ')
  echo "$lib: $t | $tr"
done
cp /tmp/keep.so redux_amd/libredux_hip.so
