#!/usr/bin/env python3
"""Soak (GPU box): many randomly shaped inputs through the device encoder + decoder, EVERY block compared
with the CPU oracle (16 host threads).  Looks for rare-path bugs: the speculative coder half and its redo,
long pending runs, freeze crossings at small freq_bits, ragged tails, dead lanes.
usage: tools/soak_encode.py [seconds=120] [seed=1] [big]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WIDTHS = [(8, 30, 32), (8, 30, 32), (8, 22, 24), (8, 14, 16), (8, 16, 32), (8, 10, 32)]


def make(n):
    kind = rng.integers(0, 8)
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:  # zipf-like with a random exponent and a random permutation of the alphabet
        w = 1.0 / np.arange(1, 257) ** rng.uniform(0.5, 3.0)
        return rng.permutation(256).astype(np.uint8)[rng.choice(256, n, p=w / w.sum())]
    if kind == 2:  # long runs
        v = rng.integers(0, 256, n // 50 + 2, dtype=np.uint8)
        return np.repeat(v, rng.integers(1, 100, v.size))[:n].astype(np.uint8)
    if kind == 3:  # two symbols, one very rare: large code lengths next to tiny ones
        return np.where(rng.random(n) < rng.uniform(1e-4, 0.02), rng.integers(0, 256), rng.integers(0, 256)).astype(np.uint8)
    if kind == 4:  # ramps
        return (np.arange(n) * int(rng.integers(1, 7)) >> int(rng.integers(0, 6))).astype(np.uint8)
    if kind == 5:  # constant
        return np.full(n, rng.integers(0, 256), dtype=np.uint8)
    if kind == 6:  # alternating regimes inside a block
        a = rng.integers(0, 4, n, dtype=np.uint8)
        a[n // 3: 2 * n // 3] = rng.integers(0, 256, 2 * n // 3 - n // 3, dtype=np.uint8)
        return a
    return (rng.integers(0, 256, n) & rng.integers(0, 256)).astype(np.uint8)  # masked alphabet


t_end = time.time() + budget
it = blocks = 0
while time.time() < t_end:
    params = WIDTHS[rng.integers(0, len(WIDTHS))]
    # big: blocks beyond 64 KiB: the small-grid kernels window by window (linear slots below 64 blocks, group areas from there),
    # k_decode_wave up to 1024 blocks and k_decode_cells<8> beyond
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    bs = int(rng.choice([100000, 131072, 262144, 70001, 1500000, 65504 * 2, 65504 * 3 - 1])) if big else int(rng.choice([48, 1000, 4096, 16384, 65536]))
    nb = int(rng.integers(1, 140)) if big else int(rng.integers(1, 400)) if bs >= 16384 else int(rng.integers(1, 3000))
    if big and bs >= 1000000:
        nb = int(rng.integers(1, 8))
    elif big and bs < 140000 and rng.random() < 0.15:
        nb = int(rng.integers(1025, 1400))
    n = max(0, nb * bs - int(rng.integers(0, bs)))
    host = np.ascontiguousarray(make(n) if n else np.zeros(0, dtype=np.uint8))
    n = int(host.size)  # (a generator may return fewer bytes than asked)
    want, wst = ox.compress_blocks(host, bs, params, nthreads=16, slot=bs * 5 + 1024)  # (8,10,32) frozen: up to 12 bits/symbol
    assert (wst == 0).all()
    d_in = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    enc = rx.DeviceEncoder(params, bs, max(n, 1))
    out, offs, status, summary = enc.encode(d_in)
    torch.cuda.synchronize()
    assert summary.tolist() == [0, 0], (it, params, bs, n, summary.tolist())
    offs_h = offs.cpu().numpy()
    out_h = out[: int(offs_h[-1])].cpu().numpy()
    for b in range(len(want)):
        got = out_h[int(offs_h[b]): int(offs_h[b + 1])].tobytes()
        assert got == want[b], f"iteration {it}: params {params} block_size {bs} n {n} block {b} differs"
    dec = rx.DeviceDecoder(params, bs, len(want))
    d_out, d_sizes, d_status, d_sum = dec.decode(out[: int(offs_h[-1])], offs)
    torch.cuda.synchronize()
    if d_sum.tolist() != [0, 0]:
        bad = torch.nonzero(d_status).flatten().tolist()
        b = bad[0]
        os.makedirs("gpurun_out", exist_ok=True)
        np.save("gpurun_out/soak_fail_block.npy", host[b * bs: (b + 1) * bs])
        print(f"DECODE STATUS iteration {it}: params {params} block_size {bs} n {n} summary {d_sum.tolist()} "
              f"bad blocks {bad[:8]} (of {len(want)}) status {int(d_status[b])} size {int(d_sizes[b])} "
              f"stream bytes {len(want[b])}; block saved to gpurun_out/soak_fail_block.npy", flush=True)
        sys.exit(1)
    got = d_out.cpu().numpy()
    for b in range(len(want)):
        ln = min(bs, n - b * bs) if n else 0
        ok = int(d_sizes[b]) == ln and (got[b * bs: b * bs + ln] == host[b * bs: b * bs + ln]).all()
        if not ok:
            blk = host[b * bs: b * bs + ln]
            diff = np.nonzero(got[b * bs: b * bs + ln] != blk)[0]
            os.makedirs("gpurun_out", exist_ok=True)
            np.save("gpurun_out/soak_fail_block.npy", blk)
            print(f"DECODE MISMATCH iteration {it}: params {params} block_size {bs} n {n} block {b}/{len(want)} "
                  f"size got {int(d_sizes[b])} want {ln} status {int(d_status[b])} first diff at {diff[:5]} "
                  f"stream bytes {len(want[b])}; block saved to gpurun_out/soak_fail_block.npy", flush=True)
            sys.exit(1)
    it += 1
    blocks += len(want)
    if it % 20 == 0:
        print(f"{it} inputs, {blocks} blocks bit-exact", flush=True)
print(f"soak done: {it} inputs, {blocks} blocks, every block bit-exact against the oracle and round-tripped")
