#!/usr/bin/env python3
"""Soak (GPU box) of the lock-step kernels for the symbol widths other than 8 (redux_gen.hpp encoders, redux_decode_cells.hpp
decoder): random Parameters with symbol_bits 1 .. 12 and code_bits <= 32, block sizes up to 64 KiB (so the lock-step loops run
for thousands of turns, across freeze points), FULL waves, ragged tails; every stream against the CPU oracle on the box's
host threads, every decode against the oracle's decode -- intact streams and damaged ones (flipped bits, damaged dwords,
truncations, trailing bytes: status, length and bytes).
usage: tools/soak_cells.py [seconds=120] [seed=1]"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))


def oracle_decode_raw(stream, cap, params):
    a = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    st = ox.lib().ox_decompress(a.ctypes.data if len(a) else None, len(a), out.ctypes.data, cap, params[0], params[1], params[2],
                                ox.TREE, C.byref(bi), C.byref(bo))
    return (4 if st == 3 else st), out[: bo.value].tobytes()


t_end = time.time() + budget
it = blocks = damaged = 0
seen = set()
while time.time() < t_end:
    sb = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]))
    fb = int(rng.integers(sb + 2, 31))
    cb = int(rng.integers(fb + 2, 33)) if fb + 2 <= 32 else 32
    if cb > 32 or fb + cb > 64:
        continue
    params = (sb, fb, cb)
    bs = int(rng.choice([64, 1000, 4096, 16384, 65536, 12345, 33332]))
    nb = int(rng.integers(1, 3)) * 64 + int(rng.integers(0, 9))
    if bs >= 16384:
        nb = min(nb, 72)
    n = max(0, nb * bs - int(rng.integers(0, bs)))
    alpha = rng.uniform(0.0, 2.5)
    w = 1.0 / np.arange(1, 257) ** alpha
    host = rng.choice(256, n, p=w / w.sum()).astype(np.uint8) if n else np.zeros(0, dtype=np.uint8)
    if rng.random() < 0.3:  # long runs of one byte: long pending runs, hot tree paths
        host[: n // 2] = int(rng.integers(0, 256))
    out, offs, st = rx.compress_blocks(host, bs, params)
    nblk = len(offs) - 1
    cap = bs * 4 + 4096
    want = list(pool.map(lambda b: ox.compress(host[b * bs:(b + 1) * bs].tobytes(), params, cap=cap)[0], range(nblk)))
    for b in range(nblk):
        assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want[b], f"iteration {it}: params {params} block_size {bs} n {n} block {b}: stream differs"
    # decode: some streams damaged
    streams = list(want)
    for b in range(nblk):
        r = rng.random()
        s = bytearray(streams[b])
        if r < 0.08 and s:
            s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif r < 0.14:
            s = s[: int(rng.integers(0, len(s) + 1))]
        elif r < 0.18:
            s += rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()
        elif r < 0.22 and len(s) > 8:
            j = int(rng.integers(0, len(s) - 4))
            s[j: j + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        else:
            continue
        streams[b] = bytes(s)
        damaged += 1
    offs2 = np.zeros(nblk + 1, dtype=np.uint64)
    offs2[1:] = np.cumsum([len(x) for x in streams])
    dense = np.frombuffer(b"".join(streams), dtype=np.uint8)
    dec, sizes, status = rx.decompress_blocks(dense, offs2, bs, params, check=False)
    back = list(pool.map(lambda b: oracle_decode_raw(streams[b], bs, params), range(nblk)))
    for b in range(nblk):
        stw, outw = back[b]
        assert int(status[b]) == stw, (it, params, bs, b, int(status[b]), stw)
        assert int(sizes[b]) == len(outw) and dec[b * bs: b * bs + len(outw)].tobytes() == outw, (it, params, bs, b, int(sizes[b]), len(outw))
    seen.add(params)
    it += 1
    blocks += nblk
    if it % 10 == 0:
        print(f"{it} inputs, {blocks} blocks ({damaged} damaged), {len(seen)} distinct triples", flush=True)
print(f"cell-decoder soak done: {it} inputs, {blocks} blocks ({damaged} damaged), {len(seen)} distinct triples, every block equal to the oracle's")
