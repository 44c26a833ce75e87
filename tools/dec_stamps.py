#!/usr/bin/env python3
"""Diagnostic: per-segment cycle stamps of k_decode_lock's lock-step step (build with -DREDUX_DEC_STAMPS).
Segments: 0 reader, 1 value, 2 round A (levels 7-5), 3 round B (4-2), 4 round C (1-0), 5 narrowing+renorm, 6 commit."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx
from redux_amd import _lib

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
static = len(sys.argv) > 2 and sys.argv[2] == "static"  # the static-table decoder (same body) on Zipf bytes
B = 65536
n = nblocks * B
if static:
    import numpy as np
    d_in = rx.gen_zipf(n)
    hist = torch.bincount(d_in[: 1 << 24].to(torch.int64), minlength=256).cpu().numpy().astype(np.float64)
    f = np.maximum(1, np.floor(hist / hist.sum() * 65000)).astype(np.int64)
    cum = [0] + list(np.cumsum(np.append(f, 1)))
    enc = rx.DeviceStaticCoder((8, 30, 32), cum, B, n)
    dec = enc
else:
    d_in = rx.gen_iid(n, 0x5EED0001, 0, device="cuda:0")
    enc = rx.DeviceEncoder((8, 30, 32), B, n, device="cuda:0")
    dec = rx.DeviceDecoder((8, 30, 32), B, nblocks, device="cuda:0")
enc.encode(d_in)
torch.cuda.synchronize()
out_bytes = int(enc.offsets[nblocks].item())
dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
L = _lib.lib()
buf = (ctypes.c_uint64 * 8)()
L.redux_debug_dec_stamps.argtypes = [ctypes.c_void_p]
L.redux_debug_dec_stamps.restype = ctypes.c_int
assert L.redux_debug_dec_stamps(buf) == 0
ts = list(buf)
steps = ts[7]
tot = sum(ts[:7])
print(f"decode {ms:.2f} ms, {steps} lock-step steps, {ms * 1e6 / 65537:.1f} ns/step")
names = ["reader", "value", "roundA", "roundB", "roundC", "narrow+renorm", "commit"]
for nme, t in zip(names, ts[:7]):
    print(f"  {nme:14s} {t / steps:8.2f} ticks/step  {100.0 * t / tot:5.1f} %")
print(f"  total          {tot / steps:8.2f} ticks/step  -> {ms * 1e6 / 65537 / (tot / steps):.3f} ns/tick")
