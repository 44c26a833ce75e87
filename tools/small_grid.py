#!/usr/bin/env python3
"""Side measurement (GPU box): encode latency of small launches -- n blocks of 64 KiB resident in HBM, kernel(s) + scan + compaction,
median of 5 -- for the grids where most SIMDs idle.  Prints one JSON line per n; `tools/small_grid.py a.so b.so` compares
builds (the library is loaded once per process, so each build runs in a child process)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(HERE, "redux_amd", "libredux_hip.so")


def measure():
    import torch
    sys.path.insert(0, HERE)
    import redux_amd as rx
    from redux_amd import _lib
    import ctypes as C
    block = 65536
    for nb in [int(x) for x in os.environ.get("SMALL_GRID_BLOCKS", "1,16,62,64,256,1024,2048,4096").split(",")]:
        n = nb * block
        d_in = rx.gen_zipf(n)
        enc = rx.DeviceEncoder((8, 30, 32), block, n)
        dec = rx.DeviceDecoder((8, 30, 32), block, nb)
        for _ in range(2):
            out, offs, st, summ = enc.encode(d_in)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, offs, st, summ = enc.encode(d_in)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        verify = not os.environ.get("SMALL_GRID_NOVERIFY")  # (timing-only variants whose output is invalid)
        if verify:
            assert summ.tolist() == [0, 0]
            total = int(offs[nb].item())
            d_out, sizes, dst, dsum = dec.decode(out[:total], offs)
            torch.cuda.synchronize()
            assert torch.equal(d_out[:n], d_in)
        cp = _lib.Params(8, 30, 32)
        name = _lib.lib().redux_encode_kernel_name(C.byref(cp), C.c_void_p(d_in.data_ptr()), n, block).decode()
        print(json.dumps({"blocks": nb, "encode_ms": round(sorted(ts)[2], 3), "MBps": round(n / sorted(ts)[2] / 1e3, 1),
                          "kernel": name.split(" (")[0], "roundtrip": verify}), flush=True)


if __name__ == "__main__":
    libs = sys.argv[1:]
    if not libs:
        measure()
        sys.exit(0)
    for lib in libs:  # (each build in a child process that loads it through REDUX_LIB: the in-tree library is never overwritten)
        print("==", lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], check=False, timeout=300, env=dict(os.environ, REDUX_LIB=os.path.abspath(lib)))
