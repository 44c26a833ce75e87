#!/usr/bin/env python3
"""Side measurement (GPU box): k_decode_wave (one block per wave: blocks the lock-step decoder does not take, in launches of at
most 1024 blocks; ONE block of any length = redux_decompress, the literal redux::decompress).  HBM resident, HIP events,
decode only.  Prints one JSON line per shape."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from redux_amd import _lib  # noqa: E402

P = (8, 30, 32)
SHAPES = (("iid", 1 << 20, 1), ("zipf", 1 << 20, 1), ("iid", 1 << 20, 16), ("iid", 131072, 62), ("zipf", 262144, 256),
          ("zipf", 4 << 20, 1))  # (the last one: three quarters of it past the reciprocal table's window of 2^20 entries)
if len(sys.argv) > 1:  # tools/measure_wave.py zipf,4194304,1 ...
    SHAPES = tuple((a.split(",")[0], int(a.split(",")[1]), int(a.split(",")[2])) for a in sys.argv[1:])
# REDUX_MEASURE_PAD=<bytes>: allocate that much first, so that every buffer of the measurement lands elsewhere (placement check)
_pad = torch.empty(int(os.environ.get("REDUX_MEASURE_PAD", "0")) or 1, dtype=torch.uint8, device="cuda")
for kind, bs, nb in SHAPES:
    n = bs * nb
    d_in = (rx.gen_iid if kind == "iid" else rx.gen_zipf)(n)
    enc = rx.DeviceEncoder(P, bs, n)
    dec = rx.DeviceDecoder(P, bs, nb)
    out, offs, st, sm = enc.encode(d_in)
    torch.cuda.synchronize()
    assert sm.tolist() == [0, 0]
    total = int(offs[nb].item())
    dec.decode(out[:total], offs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d_out, sizes, dst, dsum = dec.decode(out[:total], offs)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    assert dsum.tolist() == [0, 0] and torch.equal(d_out[:n], d_in)
    cp = _lib.Params(*P)
    name = _lib.lib().redux_decode_kernel_name_n(C.byref(cp), None, bs, nb).decode().split(" (")[0]
    ms = sorted(ts)[1]
    print(json.dumps({"data": kind, "block_size": bs, "blocks": nb, "kernel": name, "decode_ms": round(ms, 3),
                      "MBps_per_block": round(bs / ms / 1e3, 3), "MBps": round(n / ms / 1e3, 2), "ns_per_symbol": round(ms * 1e6 / bs, 1)}), flush=True)
