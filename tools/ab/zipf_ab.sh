#!/bin/bash
# usage: tools/ab/zipf_ab.sh lib1.so lib2.so ...  -> Zipf workload: whole-step and kernel ms (10 + 10 steps), then FETCH_SIZE / WRITE_SIZE of the
# encode kernel, per build (GPU box; a variant is loaded through REDUX_LIB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export REDUX_LIB=$(realpath $lib)
  tag=$(basename $lib .so)
  echo "== $lib: $(timeout -k 10 200 python bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-decode --workload zipf 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*,\|kernel_ms[^}]*' | tr '\n' ' ')"
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/zab_$tag/f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --workload zipf > /dev/null 2>&1
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/zab_$tag/w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --workload zipf > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/zab_$tag k_encode | grep "SIZE"
done
