#!/usr/bin/env python3
"""One shape through the host-pointer ABI (for profiling): python tools/host_abi_once.py <nblocks> <reps> [enc|dec|both]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from redux_amd import _lib  # noqa: E402

BLOCK = 65536
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
what = sys.argv[3] if len(sys.argv) > 3 else "both"
L = _lib.lib()
cp = _lib.Params(8, 30, 32)
n = nblocks * BLOCK
host = rx.gen_iid(n).cpu().numpy()
cap = L.redux_encode_bound(C.byref(cp), n, BLOCK)
out = np.zeros(cap, dtype=np.uint8)
out[:] = 1
offs = np.zeros(nblocks + 1, dtype=np.uint64)
status = np.zeros(nblocks, dtype=np.int32)
back = np.ones(n, dtype=np.uint8)
sizes = np.zeros(nblocks, dtype=np.uint32)
tr = (C.c_double * 512)()
for r in range(reps):
    t0 = time.perf_counter()
    assert L.redux_encode_blocks(C.byref(cp), host.ctypes.data, n, BLOCK, out.ctypes.data, cap, offs.ctypes.data, status.ctypes.data) == 0
    t1 = time.perf_counter()
    k = min(int(L.redux_host_trace(tr, 512)), 512)
    print("encode %.1f ms" % ((t1 - t0) * 1e3), [[round(tr[i + j] * 1e3, 1) for j in range(4)] for i in range(0, k, 4)])
    if what != "enc":
        t1 = time.perf_counter()
        assert L.redux_decode_blocks(C.byref(cp), out.ctypes.data, offs.ctypes.data, nblocks, BLOCK, back.ctypes.data, n, sizes.ctypes.data, status.ctypes.data) == 0
        t2 = time.perf_counter()
        k = min(int(L.redux_host_trace(tr, 512)), 512)
        print("decode %.1f ms" % ((t2 - t1) * 1e3), [[round(tr[i + j] * 1e3, 1) for j in range(4)] for i in range(0, k, 4)])
