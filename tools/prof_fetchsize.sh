#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the decoders' access pattern (tools/ubench/fetchsize.hip): one PMC pass each.
# Usage: tools/prof_fetchsize.sh <tag>   -> gpurun_out/fetchsize_<tag>/summary.txt
TAG=${1:-x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/fetchsize_$TAG; mkdir -p $OUT tools/ubench/bin
hipcc --offload-arch=gfx950 -O2 -o tools/ubench/bin/fetchsize tools/ubench/fetchsize.hip || exit 1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- ./tools/ubench/bin/fetchsize > $OUT/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- ./tools/ubench/bin/fetchsize > $OUT/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rdreq -- ./tools/ubench/bin/fetchsize > $OUT/rdreq.log 2>&1
{ tail -1 $OUT/fetch.log; python3 tools/pmc_summary.py $OUT k_stream; for k in 'k_lanes<0>' 'k_lanes<1>' 'k_lanes<2>'; do python3 tools/pmc_summary.py $OUT "$k"; done; } > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
