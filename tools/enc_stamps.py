#!/usr/bin/env python3
"""Diagnostic: per-wave work / in-barrier cycles of k_encode_pair (build with -DREDUX_STAMPS -DREDUX_KEEP8=0).
The kernel leaves {work, in-barrier, count, role} per wave in the spare slot of the workspace."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx
from redux_amd import _lib

BLOCK = 65536
nb = 65536
n = nb * BLOCK
d_in = rx.gen_iid(n)
enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
enc.encode_slots(d_in)
torch.cuda.synchronize()
seq = len(sys.argv) > 1 and sys.argv[1] == "seq"  # "seq": the stamped launch follows a compaction, as in bench.py
for _ in range(3):
    enc.encode_slots(d_in)
    if seq:
        enc.compact(n)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); enc.encode_slots(d_in); e1.record(); torch.cuda.synchronize()
L = _lib.lib()
cap = L.redux_encode_slot_bytes(C.byref(enc.cp), BLOCK)
stride = (cap + 32 + 127) // 128 * 128
if (stride // 128) % 2 == 0:
    stride += 128
al = lambda v, a: (v + a - 1) // a * a
rc_n = BLOCK + 1 + 32
off_slots = al(al(rc_n * 8, 256) + nb * 4, 256) + 256 + 8192  # + the mode word and the role book
off = enc.ws_off + off_slots + nb * stride
raw = enc.ws[off: off + 1024 * 2 * 32].cpu().numpy().view(np.uint64).reshape(1024, 2, 4)
print(f"kernel {e0.elapsed_time(e1):.2f} ms (stamped build)")
# placement census: per (xcc, se, sh, cu, simd), how many model and coder waves
from collections import Counter
cnt = {}
for b in range(1024):
    for w in range(2):
        tag = int(raw[b, w, 3]); hw = (tag >> 8) & 0xFFFF; xcc = (tag >> 24) & 0xF
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF, (hw >> 4) & 3)
        c = cnt.setdefault(key, [0, 0]); c[tag & 1] += 1
print("SIMDs used:", len(cnt), " (model, coder) waves per SIMD:", dict(Counter(tuple(v) for v in cnt.values())))
for w, name in ((0, "model wave"), (1, "coder wave")):
    work, wait, cnt = raw[:, w, 0].astype(float), raw[:, w, 1].astype(float), raw[:, w, 2].astype(float)
    print(f"{name}: barriers {cnt.mean():.0f}; per 8-symbol half: work {np.mean(work / cnt):.0f} cycles, in-barrier "
          f"{np.mean(wait / cnt):.0f} cycles (work/symbol {np.mean(work / cnt) / 8:.0f}, wait/symbol {np.mean(wait / cnt) / 8:.0f}); role tag {int(raw[0, w, 3]) & 1}")
