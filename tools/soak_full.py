#!/usr/bin/env python3
"""Soak (GPU box) at the headline shape: 65,536 blocks x 64 KiB (4 GiB) per pass, EVERY block's bytes
against the CPU oracle (16 threads, ~20 s per pass).  usage: tools/soak_full.py [passes=3] [seed=1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

BLOCK, NB = 65536, 65536
n = BLOCK * NB
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P = (8, 30, 32)
enc = rx.DeviceEncoder(P, BLOCK, n)
dec = rx.DeviceDecoder(P, BLOCK, NB)
for k in range(passes):
    t0 = time.time()
    if k % 3 == 0:
        d_in, name = rx.gen_iid(n, seed=seed + k), "iid"
    elif k % 3 == 1:
        d_in, name = rx.gen_zipf(n, seed=seed + k), "zipf(1.2)"
    else:  # every block its own regime: zipf bytes masked / shifted per block, runs in some
        d_in = rx.gen_zipf(n, seed=seed + k).view(NB, BLOCK)
        g = torch.Generator(device="cuda").manual_seed(seed + k)
        mask = torch.randint(1, 256, (NB, 1), device="cuda", dtype=torch.uint8, generator=g)
        shift = torch.randint(0, 256, (NB, 1), device="cuda", dtype=torch.uint8, generator=g)
        d_in = ((d_in & mask) + shift).contiguous().view(-1)
        name = "per-block regimes"
    out, offs, status, summary = enc.encode(d_in)
    torch.cuda.synchronize()
    assert summary.tolist() == [0, 0]
    host = d_in.cpu().numpy()
    o_buf, o_sizes, o_status, slot = ox.compress_blocks_raw(host, BLOCK, P, nthreads=16)
    assert (o_status == 0).all()
    offs_h = offs.cpu().numpy().astype(np.int64)
    sizes = np.diff(offs_h)
    assert (sizes == o_sizes.astype(np.int64)).all(), f"pass {k} ({name}): sizes differ at blocks {np.nonzero(sizes != o_sizes)[0][:8]}"
    out_h = out[: int(offs_h[-1])].cpu().numpy()
    bad = [b for b in range(NB) if not np.array_equal(out_h[offs_h[b]: offs_h[b + 1]], o_buf[b * slot: b * slot + int(o_sizes[b])])]
    assert not bad, f"pass {k} ({name}): bytes differ in blocks {bad[:8]}"
    d_out, d_sizes, d_status, d_sum = dec.decode(out[: int(offs_h[-1])], offs)
    torch.cuda.synchronize()
    assert d_sum.tolist() == [0, 0] and torch.equal(d_out, d_in)
    print(f"pass {k} ({name}): 65536 blocks, {int(offs_h[-1])} stream bytes, every block equal to the oracle's, decoded back; {time.time() - t0:.0f} s", flush=True)
print(f"full-size soak done: {passes} passes x 65536 blocks x 64 KiB")
