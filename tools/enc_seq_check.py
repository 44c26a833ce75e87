#!/usr/bin/env python3
"""Experiment: time of the coder kernel alone vs inside the encode+compact sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx
BLOCK = 65536; nb = 65536; n = nb * BLOCK
d_in = rx.gen_iid(n)
enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
def run(with_compact, k=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i in range(k):
        ev[i][0].record(); enc.encode_slots(d_in); ev[i][1].record()
        if with_compact:
            enc.compact(n)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return sum(t[4:]) / len(t[4:])
for _ in range(2):
    print(f"encode alone {run(False):.2f} ms   encode inside encode+compact {run(True):.2f} ms")

# filler experiment: a memory-bound torch copy of `gb` GB between coder kernels instead of the compaction
src = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0"); dstb = torch.empty_like(src)
def run_fill(reps, k=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i in range(k):
        ev[i][0].record(); enc.encode_slots(d_in); ev[i][1].record()
        for _ in range(reps):
            dstb.copy_(src)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return sum(t[4:]) / len(t[4:])
for reps in (0, 1, 2, 4, 0):
    print(f"filler copies of 2 GiB between coder kernels: {reps}  -> coder kernel {run_fill(reps):.2f} ms")

# idle-gap experiment: a one-thread spin kernel (the chip is nearly idle, like during k_scan_sizes) between coder kernels
def run_gap(cycles, k=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i in range(k):
        ev[i][0].record(); enc.encode_slots(d_in); ev[i][1].record()
        if cycles:
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return sum(t[4:]) / len(t[4:])
for cyc in (0, 200_000, 1_000_000, 4_000_000, 0):
    print(f"one-thread spin of {cyc} cycles between coder kernels -> coder kernel {run_gap(cyc):.2f} ms")

# which side of the compaction slows the next coder kernel: reading the slots, or writing the dense output?
def run_touch(kind, k=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    wsv = enc.ws[: (enc.ws.numel() // 8) * 8].view(torch.int64)
    for i in range(k):
        ev[i][0].record(); enc.encode_slots(d_in); ev[i][1].record()
        if kind == "read_ws":
            wsv.sum()
        elif kind == "write_out":
            enc.out.zero_()
        elif kind == "both":
            wsv.sum(); enc.out.zero_()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return sum(t[4:]) / len(t[4:])
for kind in ("none", "read_ws", "write_out", "both", "none"):
    print(f"between coder kernels: {kind:9s} -> coder kernel {run_touch(kind):.2f} ms")
