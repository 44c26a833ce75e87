#!/usr/bin/env python3
"""Side measurement (GPU box): the lock-step kernels for 4- and 12-bit symbols (redux_gen.hpp) next to the one-lane-per-block
port (redux_any.hpp) at the neighbouring widths 5 and 11, which take the same number of tree levels +-1.  Device resident,
HIP events, whole encode pass (coder + scan + compaction) and decode.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402

BLOCK = 65536
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = nb * BLOCK
d_in = rx.gen_zipf(n)
res = {}
PARAMS = ((4, 28, 32), (5, 27, 32), (12, 20, 32), (11, 21, 32))
if len(sys.argv) > 2:  # tools/measure_gen.py <blocks> 9,23,32 10,22,32 ...
    PARAMS = tuple(tuple(int(x) for x in a.split(",")) for a in sys.argv[2:])
for params in PARAMS:
    enc = rx.DeviceEncoder(params, BLOCK, n)
    dec = rx.DeviceDecoder(params, BLOCK, nb)
    out, offs, st, sm = enc.encode(d_in)
    torch.cuda.synchronize()
    assert sm.tolist() == [0, 0]
    total = int(offs[-1].item())
    dec.decode(out[:total], offs)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    enc.encode(d_in)
    e[1].record()
    d_out, sizes, dst, dsum = dec.decode(out[:total], offs)
    e[2].record()
    torch.cuda.synchronize()
    assert dsum.tolist() == [0, 0]
    keep = BLOCK * 8 // params[0] * params[0] // 8
    got = d_out.view(nb, BLOCK)[:, :keep]
    assert bool((sizes == keep).all()) and torch.equal(got, d_in.view(nb, BLOCK)[:, :keep])
    res["%d,%d,%d" % params] = {"encode_MBps": round(n / e[0].elapsed_time(e[1]) / 1e3, 1), "decode_MBps": round(n / e[1].elapsed_time(e[2]) / 1e3, 1),
                                "ratio": round(total / n, 4)}
res["blocks"] = nb
res["note"] = "k_encode_gen / k_encode_gen_pair (redux_gen.hpp) and k_decode_cells (redux_decode_cells.hpp): lock-step kernels for every width"
print(json.dumps(res))
