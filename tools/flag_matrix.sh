#!/bin/bash
# (CPU, ~5 min) Every A/B build flag DESIGN.md names still compiles for gfx950 (device code only, nothing is run).
cd "$(dirname "$0")/.." || exit 1
fail=0
for f in "-DREDUX_MASK_TABLE=0" "-DREDUX_PERM_ADDENDS=0" "-DREDUX_MASK_BUFFER=0" "-DREDUX_RING_PAIRS=0" "-DREDUX_ONE_WAIT=0" \
         "-DREDUX_CODER_TOUCH=1 -DREDUX_TOUCH_AT=64" "-DREDUX_ROWS=0" "-DREDUX_NO_LINE_QUEUE" "-DREDUX_STORE_X4 -DREDUX_ROWS=0" \
         "-DREDUX_CODER_BRANCHY" "-DREDUX_TOP_REG=0" "-DREDUX_KEEP8=0 -DREDUX_STAMPS" "-DREDUX_MASK_AHEAD=4" "-DREDUX_MASK_SDWA=0" \
         "-DREDUX_STATIC_LUT=0" "-DREDUX_AB" "-DREDUX_MODEL_DEPTH=2" "-DREDUX_PROBE_VMEM=1" "-DREDUX_PROBE_MODEL=2 -DREDUX_PROBE_CODER=2" \
         "-DREDUX_NO_STORE" "-DREDUX_DEC_STAMPS" "-DREDUX_DEC_GSTAMPS" "-DREDUX_DEC_DUP=3" "-DREDUX_DEC_CENSUS" "-DREDUX_MODEL_PRIO=0" "-DREDUX_CLAIMS=0"; do
  if hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value --cuda-device-only -c -o /tmp/flag_matrix.o redux_amd/csrc/redux_hip.hip $f 2>/tmp/flag_matrix.err; then
    echo "ok    $f"
  else
    echo "FAIL  $f"; grep error /tmp/flag_matrix.err | head -3; fail=1
  fi
done
exit $fail
