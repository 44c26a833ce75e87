#!/usr/bin/env python3
"""Round-end side measurements (GPU box): PCIe-inclusive host-pointer encode rate.
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402

BLOCK = 65536
n = 16384 * BLOCK  # 1 GiB
host = rx.gen_iid(n).cpu().numpy()
rx.compress_blocks(host[: 64 * BLOCK], BLOCK, (8, 30, 32))  # warm-up (context, code object)
t0 = time.perf_counter()
out, offs, st = rx.compress_blocks(host, BLOCK, (8, 30, 32))
dt = time.perf_counter() - t0
t0 = time.perf_counter()
dec, sizes, st2 = rx.decompress_blocks(out, offs, BLOCK, (8, 30, 32))
dt2 = time.perf_counter() - t0
assert (dec == host).all()
print(json.dumps({"host_pointer_encode_MBps": round(n / dt / 1e6, 1), "host_pointer_decode_MBps": round(n / dt2 / 1e6, 1),
                  "bytes": n, "note": "redux_encode_blocks / redux_decode_blocks: pageable host memory, hipMalloc + H2D + kernels + D2H + hipFree per call"}))
