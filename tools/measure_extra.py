#!/usr/bin/env python3
"""Side measurements of the host-pointer ABI (GPU box): PCIe-inclusive rates of redux_encode_blocks /
redux_decode_blocks -- what a Rust / C caller with plain host memory gets.  Never bench.py's `value`.
Buffers are allocated and touched BEFORE the timed calls (a caller that reuses its buffers): first-touch
page faults of a fresh 1 GiB numpy array would otherwise be most of the time.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from redux_amd import _lib  # noqa: E402

BLOCK = 65536
PARAMS = (8, 30, 32)
L = _lib.lib()
cp = _lib.Params(*PARAMS)


def run(nblocks, reps=3):
    n = nblocks * BLOCK
    host = rx.gen_iid(n).cpu().numpy()
    cap = L.redux_encode_bound(C.byref(cp), n, BLOCK)
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(nblocks + 1, dtype=np.uint64)
    status = np.zeros(nblocks, dtype=np.int32)
    back = np.zeros(n, dtype=np.uint8)
    sizes = np.zeros(nblocks, dtype=np.uint32)
    enc, dec = [], []
    allocs = []
    tr = (C.c_double * 256)()

    def trace():
        k = min(int(L.redux_host_trace(tr, 256)), 256)
        return [[round(tr[i + j] * 1e3, 1) for j in range(4)] for i in range(0, k, 4)]

    for r in range(reps + 1):  # the first call builds / grows the context: not timed
        t0 = time.perf_counter()
        st = L.redux_encode_blocks(C.byref(cp), host.ctypes.data, n, BLOCK, out.ctypes.data, cap, offs.ctypes.data, status.ctypes.data)
        t1 = time.perf_counter()
        assert st == 0, st
        tr_e = trace()
        st = L.redux_decode_blocks(C.byref(cp), out.ctypes.data, offs.ctypes.data, nblocks, BLOCK, back.ctypes.data, n,
                                   sizes.ctypes.data, status.ctypes.data)
        t2 = time.perf_counter()
        assert st == 0, st
        tr_d = trace()
        allocs.append(L.redux_host_allocations())
        if r:
            enc.append(t1 - t0)
            dec.append(t2 - t1)
    assert (back == host).all() and (sizes == BLOCK).all()
    assert allocs[-1] == allocs[0], allocs  # no hipMalloc / hipHostMalloc after the first call
    return {"bytes": n, "encode_GBps": round(n / min(enc) / 1e9, 2), "decode_GBps": round(n / min(dec) / 1e9, 2),
            "encode_ms": [round(x * 1e3, 1) for x in enc], "decode_ms": [round(x * 1e3, 1) for x in dec],
            "allocations_after_each_call": allocs,
            "timeline_ms [stage begins, enqueued, kernels done, drained] per chunk": {"encode": tr_e, "decode": tr_d}}


res = {"1GiB": run(16384), "4GiB": run(65536, reps=2)}
res["note"] = ("redux_encode_blocks / redux_decode_blocks on pageable host memory, 64 KiB blocks, (8,30,32): CPU threads -> pinned "
               "ring -> H2D, chunk kernels on their own streams, D2H; persistent context (allocation counter constant after the "
               "first call). Floor for 1 GiB: PCIe one way (~19 ms at 57 GB/s) + one kernel latency (12 ms encode, 28 ms decode).")
print(json.dumps(res))
