#!/usr/bin/env python3
"""Diagnostic: where a lock-step decode step's cycles go (library built with -DREDUX_DEC_PROBE; never the product build).
Prints, per step: the two LDS shadows, the waits behind them, and everything else; per group: the preamble."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx
from redux_amd import _lib

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B = 65536
n = nblocks * B
d_in = rx.gen_iid(n, 0x5EED0001, 0, device="cuda:0")
enc = rx.DeviceEncoder((8, 30, 32), B, n, device="cuda:0")
dec = rx.DeviceDecoder((8, 30, 32), B, nblocks, device="cuda:0")
enc.encode(d_in)
torch.cuda.synchronize()
out_bytes = int(enc.offsets[nblocks].item())
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
L = _lib.lib()
buf = (ctypes.c_uint64 * 8)()
L.redux_debug_dec_probe.argtypes = [ctypes.c_void_p]
L.redux_debug_dec_probe.restype = ctypes.c_int
assert L.redux_debug_dec_probe(buf) == 0
t = list(buf)
groups = t[7]
steps = 4 * groups
print(f"decode {ms:.2f} ms (probe build), {steps} steps in the fast loop")
for name, v, per in (("shadow B", t[0], steps), ("wait behind B (+stamp)", t[1], steps), ("C-wait .. next B-issue: D, E, A, B", t[2], steps),
                     ("shadow C", t[3], steps), ("wait behind C (+stamp)", t[4], steps), ("B-wait .. C-issue: levels 4, 3", t[5], steps),
                     ("group preamble", t[6], groups)):
    print(f"  {name:38s} {v / per:8.1f} cycles")
