#!/usr/bin/env python3
"""Soak (GPU box) of the device decoder on DAMAGED streams in full waves: valid streams of 64+ blocks,
a third of them with a flipped bit, a damaged dword, a truncation or trailing garbage; status, decoded
length and decoded bytes of every block against the CPU oracle.  (The suite's fuzz test mixes short
garbage streams, so some lane finishes at once and the lock-step decoder never takes its unpredicated
commit; here it does until the first lane dies.)  usage: tools/soak_decode.py [seconds=120] [seed=1]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def oracle_decode(stream, cap, params):
    a = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    st = ox.lib().ox_decompress(a.ctypes.data if len(a) else None, len(a), out.ctypes.data, cap, params[0], params[1],
                                params[2], ox.TREE, C.byref(bi), C.byref(bo))
    return (4 if st == 3 else st), out[: bo.value].tobytes()  # the oracle's writer: IoError where the capacity ends


t_end = time.time() + budget
it = blocks = damaged = 0
seen = {}
while time.time() < t_end:
    params = [(8, 30, 32), (8, 22, 24), (8, 14, 16), (8, 16, 32)][rng.integers(0, 4)]
    bs = int(rng.choice([1024, 4096, 8192, 8192, 20000, 65536]))  # the long ones cross the freeze point of the narrow widths
    nb = int(rng.choice([64, 128, 64 + int(rng.integers(1, 64))]))
    alpha = rng.uniform(0.0, 2.5)
    w = 1.0 / np.arange(1, 257) ** alpha
    host = rng.choice(256, nb * bs, p=w / w.sum()).astype(np.uint8)
    out, offs, st = rx.compress_blocks(host, bs, params)
    streams = [bytearray(out[int(offs[b]): int(offs[b + 1])].tobytes()) for b in range(nb)]
    for b in range(nb):
        r = rng.random()
        s = streams[b]
        if r < 0.10 and len(s) > 8:                       # one flipped bit, anywhere (often deep inside)
            s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif r < 0.18 and len(s) > 8:                     # a damaged dword
            j = int(rng.integers(0, len(s) - 4))
            s[j: j + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        elif r < 0.25:                                    # truncated near the end (or anywhere)
            cut = int(rng.integers(max(0, len(s) - 16), len(s) + 1)) if rng.random() < 0.7 else int(rng.integers(0, len(s) + 1))
            del s[cut:]
        elif r < 0.30:                                    # trailing garbage
            s += rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()
        else:
            continue
        damaged += 1
    offs2 = np.zeros(nb + 1, dtype=np.uint64)
    offs2[1:] = np.cumsum([len(x) for x in streams])
    dense = np.frombuffer(b"".join(bytes(x) for x in streams), dtype=np.uint8)
    dec, sizes, status = rx.decompress_blocks(dense, offs2, bs, params, check=False)
    for b in range(nb):
        stt, want = oracle_decode(streams[b], bs, params)
        seen[stt] = seen.get(stt, 0) + 1
        ok = int(status[b]) == stt and int(sizes[b]) == len(want) and dec[b * bs: b * bs + len(want)].tobytes() == want
        if not ok:
            os.makedirs("gpurun_out", exist_ok=True)
            np.save("gpurun_out/soak_decode_fail_stream.npy", np.frombuffer(bytes(streams[b]), dtype=np.uint8))
            print(f"MISMATCH iteration {it}: params {params} block_size {bs} nb {nb} block {b}: status {int(status[b])} vs {stt}, "
                  f"size {int(sizes[b])} vs {len(want)}, stream {len(streams[b])} bytes (saved to gpurun_out/soak_decode_fail_stream.npy)", flush=True)
            sys.exit(1)
    it += 1
    blocks += nb
    if it % 10 == 0:
        print(f"{it} inputs, {blocks} blocks ({damaged} damaged), statuses {seen}", flush=True)
print(f"decode soak done: {it} inputs, {blocks} blocks ({damaged} damaged), statuses {seen}: status, length and bytes equal the oracle's")
