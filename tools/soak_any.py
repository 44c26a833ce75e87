#!/usr/bin/env python3
"""Soak (GPU box) of the general-parameter path: random valid Parameters triples (model/mod.rs:64) with
symbol_bits <= 16, random data, every block against the CPU oracle, then the device round trip.
usage: tools/soak_any.py [seconds=120] [seed=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
it = blocks = 0
seen = set()
while time.time() < t_end:
    sb = int(rng.integers(1, 17))
    fb = int(rng.integers(sb + 2, min(31, 62 - (sb + 4)) + 1))
    cb = int(rng.integers(fb + 2, min(64 - fb, 63) + 1))
    params = (sb, fb, cb)
    bs = int(rng.choice([64, 1000, 4096]))
    nb = int(rng.integers(1, 100))
    n = max(0, nb * bs - int(rng.integers(0, bs)))
    alpha = rng.uniform(0.0, 2.0)
    w = 1.0 / np.arange(1, 257) ** alpha
    host = rng.choice(256, n, p=w / w.sum()).astype(np.uint8) if n else np.zeros(0, dtype=np.uint8)
    out, offs, st = rx.compress_blocks(host, bs, params)
    nblk = len(offs) - 1
    for b in range(nblk):
        want, _ = ox.compress(host[b * bs:(b + 1) * bs].tobytes(), params, cap=bs * 9 + 4096)
        assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, f"iteration {it}: params {params} block_size {bs} n {n} block {b} differs"
    dec, sizes, status = rx.decompress_blocks(out[: int(offs[-1])], offs, bs, params)
    for b in range(nblk):
        lo = b * bs
        # a block of L bytes holds floor(8L / symbol_bits) symbols; the decoder writes them back as bits
        want_back, _ = ox.decompress(out[int(offs[b]): int(offs[b + 1])].tobytes(), params, cap=bs + 8)
        assert int(status[b]) == 0 and dec[lo: lo + int(sizes[b])].tobytes() == want_back, (it, params, b)
    seen.add(params)
    it += 1
    blocks += nblk
    if it % 20 == 0:
        print(f"{it} inputs, {blocks} blocks, {len(seen)} distinct triples", flush=True)
print(f"general-parameter soak done: {it} inputs, {blocks} blocks, {len(seen)} distinct triples, every block equal to the oracle's")
