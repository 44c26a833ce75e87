#!/usr/bin/env python3
"""Diagnostic: which SIMD each k_decode_lock wave ran on (build with -DREDUX_DEC_CENSUS).
The kernel wants one wave per SIMD (4 groups of 40 KiB LDS per CU, 4 SIMDs)."""
import ctypes, sys, os
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx
from redux_amd import _lib

nblocks, B = 65536, 65536
n = nblocks * B
d_in = rx.gen_iid(n, 0x5EED0001, 0, device="cuda:0")
enc = rx.DeviceEncoder((8, 30, 32), B, n, device="cuda:0")
dec = rx.DeviceDecoder((8, 30, 32), B, nblocks, device="cuda:0")
L = _lib.lib()
L.redux_debug_dec_census.argtypes = [ctypes.c_void_p]
L.redux_debug_dec_census.restype = ctypes.c_int
buf = (ctypes.c_uint32 * 4096)()
for mode in ("decode after decode", "decode after encode+compact"):
    for _ in range(3):
        if mode.endswith("compact"):
            enc.encode(d_in)
        else:
            enc.encode(d_in) if _ == 0 else None
        out_bytes = int(enc.offsets[nblocks].item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.decode(enc.out[:out_bytes], enc.offsets[: nblocks + 1])
        e1.record()
        torch.cuda.synchronize()
    assert L.redux_debug_dec_census(buf) == 0
    per = Counter()
    for v in list(buf)[:1024]:
        assert v >> 31
        hw, xcc = v & 0xFFFF, (v >> 16) & 0xF
        per[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF, (hw >> 4) & 3)] += 1
    cus = Counter()
    for k, c in per.items():
        cus[k[:4]] += c
    print(f"{mode}: {e0.elapsed_time(e1):.2f} ms; SIMDs used {len(per)}; waves per SIMD {dict(Counter(per.values()))}; waves per CU {dict(Counter(cus.values()))}")
