#!/bin/bash
# usage: tools/ab/gen_grid_ab.sh lib.so ...  -> tools/measure_gen.py at 2048 .. 32768 blocks per build (where the 11- / 12-bit decoder's two forms cross)
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  for nb in 2048 4096 8192 16384 32768; do
    echo "$lib $nb: $(timeout -k 10 300 python tools/measure_gen.py $nb 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: v["decode_MBps"] for k, v in d.items() if isinstance(v, dict)})')"
  done
done
