#!/bin/bash
# A/B of the encode kernel's per-CU role book (GPU box only).  Build first:
#   B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value redux_amd/csrc/redux_hip.hip"
#   $B -o variants/claims.so; $B -DREDUX_CLAIMS=0 -o variants/noclaims.so
#   $B -DREDUX_STAMPS -DREDUX_KEEP8=0 -o variants/st_claims.so; $B -DREDUX_CLAIMS=0 -DREDUX_STAMPS -DREDUX_KEEP8=0 -o variants/st_noclaims.so
cd $GRAFT_REPO_ROOT
STEPS=20 WARMUP=20 tools/run_variants.sh variants/noclaims.so variants/claims.so variants/noclaims.so variants/claims.so
cp redux_amd/libredux_hip.so /tmp/keep2.so
trap 'cp /tmp/keep2.so redux_amd/libredux_hip.so' EXIT  # an interrupted run must not leave a variant build as the product library
for v in st_noclaims st_claims; do
  cp variants/$v.so redux_amd/libredux_hip.so
  echo "== $v alone"; timeout -k 10 120 python tools/enc_stamps.py
  echo "== $v seq";   timeout -k 10 120 python tools/enc_stamps.py seq
done
cp /tmp/keep2.so redux_amd/libredux_hip.so
