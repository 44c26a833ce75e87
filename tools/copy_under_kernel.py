#!/usr/bin/env python3
"""How fast pinned H2D / D2H copies run on a side stream WHILE the encode kernel occupies the chip
(GPU box).  The host-pointer pipeline (redux_host.hpp) overlaps exactly these."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402

BLOCK = 65536
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = nb * BLOCK
d_in = rx.gen_iid(n)
enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
enc.encode_slots(d_in)
torch.cuda.synchronize()
N = 256 << 20
hp = torch.empty(N, dtype=torch.uint8).pin_memory()
dv = torch.empty(N, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()


def timed_copy(dst, src):
    with torch.cuda.stream(side):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src, non_blocking=True)
        e1.record()
    return e0, e1


for name, dst, src in (("H2D", dv, hp), ("D2H", hp, dv)):
    e0, e1 = timed_copy(dst, src)
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    enc.encode_slots(d_in)
    k1.record()
    time.sleep(0.002)
    e0, e1 = timed_copy(dst, src)
    torch.cuda.synchronize()
    print(f"{name} 256 MiB pinned: alone {alone:.2f} ms = {N / alone / 1e6:.1f} GB/s; under the encode kernel ({nb} blocks) "
          f"{e0.elapsed_time(e1):.2f} ms = {N / e0.elapsed_time(e1) / 1e6:.1f} GB/s; the kernel took {k0.elapsed_time(k1):.2f} ms")
