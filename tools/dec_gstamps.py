#!/usr/bin/env python3
"""Diagnostic: per-GROUP cycle stamps of k_decode_lock (a -DREDUX_DEC_GSTAMPS build as redux_amd/libredux_hip.so):
how long the once-per-four-steps preamble (retire the stream chunk, store the output, request the next chunk) takes
against the four steps themselves."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx
from redux_amd import _lib

nblocks = 65536
B = 65536
n = nblocks * B
d_in = rx.gen_iid(n)
enc = rx.DeviceEncoder((8, 30, 32), B, n)
enc.encode(d_in)
torch.cuda.synchronize()
out_bytes = int(enc.offsets[nblocks].item())
dec = rx.DeviceDecoder((8, 30, 32), B, nblocks)
dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
e1.record()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_uint64 * 8)()
assert L.redux_debug_dec_stamps(buf) == 0
groups = buf[7] / 4
print(f"decode {e0.elapsed_time(e1):.2f} ms; per group of four steps (100 MHz ticks x 24 = cycles at 2.4 GHz): preamble {buf[0] / groups:.1f} ticks, "
      f"four steps {buf[1] / groups:.1f} ticks  -> preamble = {100 * buf[0] / (buf[0] + buf[1]):.1f} % of the kernel")
