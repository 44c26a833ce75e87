#!/usr/bin/env python3
"""How fast can this chip copy 4 GiB HBM -> HBM?  (the yardstick for k_compact: 2 x 4.3 GB in 1.71 ms = 5.0 TB/s)"""
import torch
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda:0").random_(0, 255)
b = torch.empty_like(a)
for name, fn in (("torch copy_ (u8)", lambda: b.copy_(a)),
                 ("torch copy_ (viewed as int64)", lambda: b.view(torch.int64).copy_(a.view(torch.int64)))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms per 4 GiB copy = {2 * n / ms / 1e9:.2f} TB/s read+write")
