#!/usr/bin/env python3
"""Side measurement (GPU box): the block coder at block sizes other than 64 KiB -- which kernels the launch picks and what they
reach.  Device resident, HIP events, whole encode pass (coder + scan + compaction) and decode, Zipf bytes, round trip checked.
usage: tools/measure_blocksize.py [total MiB] [s,f,c] [block KiB ...]   -> one JSON line per block size"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from redux_amd import _lib  # noqa: E402

total_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
params = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 30, 32)
sizes_kib = [int(x) for x in sys.argv[3:]] or [16, 64, 128, 256, 1024, 4096]
total_bytes = total_mib << 20
d_all = rx.gen_zipf(total_bytes)
L = _lib.lib()
cp = _lib.Params(*params)
for kib in sizes_kib:
    block = kib << 10
    nb = total_bytes // block
    n = nb * block
    d_in = d_all[:n]
    enc = rx.DeviceEncoder(params, block, n)
    dec = rx.DeviceDecoder(params, block, nb)
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
    keep = block * 8 // params[0] * params[0] // 8
    assert bool((sizes == keep).all()) and torch.equal(d_out.view(nb, block)[:, :keep], d_in.view(nb, block)[:, :keep])
    print(json.dumps({"params": list(params), "block_KiB": kib, "blocks": nb,
                      "encode_GBps": round(n / e[0].elapsed_time(e[1]) / 1e6, 2), "decode_GBps": round(n / e[1].elapsed_time(e[2]) / 1e6, 2),
                      "ratio": round(total / n, 4),
                      "encode_kernel": L.redux_encode_kernel_name(C.byref(cp), C.c_void_p(d_in.data_ptr()), n, block).decode().split(" (")[0],
                      "decode_kernel": L.redux_decode_kernel_name_n(C.byref(cp), None, block, nb).decode().split(" (")[0]}), flush=True)
    del enc, dec, out, offs, d_out
    torch.cuda.empty_cache()
