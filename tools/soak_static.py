#!/usr/bin/env python3
"""Soak (GPU box) of the static-table coder: random tables x random data, every block against the
CPU oracle's static model, plus the device round trip.  usage: tools/soak_static.py [seconds=90] [seed=1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
it = blocks = 0
while time.time() < t_end:
    params = [(8, 30, 32), (8, 22, 24), (8, 14, 16), (8, 16, 32)][rng.integers(0, 4)]
    fmax = (1 << params[1]) - 1
    lim = int(min(fmax, 10 ** rng.uniform(2.5, 8.5)))             # target total, anywhere up to freq_max
    f = np.maximum(1, (rng.random(257) ** rng.uniform(1, 8) * (lim / 257 * 1.5)).astype(np.int64))
    while f.sum() > fmax:
        f = np.maximum(1, f // 2)
    cum = [0] + [int(x) for x in np.cumsum(f)]
    bs = int(rng.choice([48, 1000, 4096, 16384]))
    nb = int(rng.integers(1, 200))
    n = max(0, nb * bs - int(rng.integers(0, bs)))
    w = f[:256].astype(np.float64) ** rng.uniform(0.3, 1.5)
    host = rng.choice(256, n, p=w / w.sum()).astype(np.uint8) if n else np.zeros(0, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    coder = rx.DeviceStaticCoder(params, cum, bs, max(n, 1))
    out, offs, status, summary = coder.encode(d_in)
    torch.cuda.synchronize()
    assert summary.tolist() == [0, 0], (it, params, bs, n, cum[-1], summary.tolist())
    offs_h = offs.cpu().numpy()
    out_h = out[: int(offs_h[-1])].cpu().numpy()
    nblk = len(offs_h) - 1
    for b in range(nblk):
        want, _ = ox.compress_static(host[b * bs:(b + 1) * bs].tobytes(), cum, params)
        assert out_h[int(offs_h[b]): int(offs_h[b + 1])].tobytes() == want, \
            f"iteration {it}: params {params} total {cum[-1]} block_size {bs} n {n} block {b} differs"
    d_out, d_sizes, d_status, d_sum = coder.decode(out[: int(offs_h[-1])], offs)
    torch.cuda.synchronize()
    assert d_sum.tolist() == [0, 0], (it, params, bs, n, cum[-1], d_sum.tolist())
    got = d_out.cpu().numpy()
    for b in range(nblk):
        ln = min(bs, n - b * bs) if n else 0
        assert int(d_sizes[b]) == ln and (got[b * bs: b * bs + ln] == host[b * bs: b * bs + ln]).all(), (it, b)
    it += 1
    blocks += nblk
    if it % 20 == 0:
        print(f"{it} inputs, {blocks} blocks bit-exact", flush=True)
print(f"static soak done: {it} inputs, {blocks} blocks, every block bit-exact against the oracle and round-tripped")
