#!/usr/bin/env python3
"""Side measurement (GPU box): the static-table coder on 4 GiB of Zipf(1.2) bytes in HBM (block size = argv[1], default 64 KiB), table = the
data's own histogram scaled to 2^16.  Prints one JSON line.  Not the headline: DESIGN.md section 3."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402

BLOCK = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = 4 << 30
nb = n // BLOCK
d_in = rx.gen_zipf(n)
hist = torch.bincount(d_in[: 1 << 24].to(torch.int64), minlength=256).cpu().numpy().astype(np.float64)
f = np.maximum(1, np.floor(hist / hist.sum() * 65000)).astype(np.int64)
cum = [0] + list(np.cumsum(np.append(f, 1)))  # EOF: frequency 1
coder = rx.DeviceStaticCoder((8, 30, 32), cum, BLOCK, n)
for _ in range(2):
    out, offs, st, summ = coder.encode(d_in)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
total = int(offs[nb].item())
coder.decode(out[:total], offs)  # first launch of the decode kernel: code object load
torch.cuda.synchronize()
e0.record()
out, offs, st, summ = coder.encode(d_in)
e1.record()
d_out, sizes, dst, dsum = coder.decode(out[:total], offs)
e2.record()
torch.cuda.synchronize()
assert summ.tolist() == [0, 0] and dsum.tolist() == [0, 0] and torch.equal(d_out, d_in)
print(json.dumps({"static_encode_MBps": round(n / e0.elapsed_time(e1) / 1e3, 1), "static_decode_MBps": round(n / e1.elapsed_time(e2) / 1e3, 1),
                  "bytes": n, "block_size": BLOCK, "compressed_over_input": round(total / n, 4), "table_total": int(cum[-1]),
                  "note": "k_encode_static + scan + compaction / k_decode_static (+ one offsets read-back), input and output in HBM"}))
