#!/bin/bash
# usage: tools/wr_ab.sh lib1.so ...  -> L2 (TCC) requests, writes and write-backs of the encode kernel per build (GPU box)
# (only counters known to this rocprofv3: an unknown name aborts it with a core dump)
cp redux_amd/libredux_hip.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  cp $lib redux_amd/libredux_hip.so
  tag=$(basename $lib .so)
  timeout -k 10 240 rocprofv3 --pmc TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum --output-format csv -d gpurun_out/wr_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > gpurun_out/wr_$tag.log 2>&1
  echo "== $lib"; python3 tools/pmc_summary.py gpurun_out/wr_$tag k_encode | grep "TCC"
done
cp /tmp/keep.so redux_amd/libredux_hip.so
