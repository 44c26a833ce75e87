#!/usr/bin/env python3
"""Experiment: overlap step i's scan+compaction with step i+1's coder kernel (two streams, double-buffered
encoders).  Prints ms/step serial vs pipelined."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import redux_amd as rx

BLOCK = 65536; nb = 65536; n = nb * BLOCK
P = (8, 30, 32)
d_in = rx.gen_iid(n, 0x5EED0001, 0, device="cuda:0")
E = [rx.DeviceEncoder(P, BLOCK, n), rx.DeviceEncoder(P, BLOCK, n)]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def serial(k):
    for i in range(k):
        e = E[i % 2]
        e.encode_slots(d_in); e.compact(n)

for _ in range(2):
    serial(4)
torch.cuda.synchronize()
t0 = time.perf_counter(); serial(K); torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / K * 1e3

sA = torch.cuda.Stream(priority=-1)   # coder kernel: high dispatch priority
sB = torch.cuda.Stream(priority=0)
def piped(k):
    ev_enc = [None] * k; ev_cmp = [None] * k
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    for i in range(k):
        e = E[i % 2]
        with torch.cuda.stream(sA):
            if i >= 2:
                sA.wait_event(ev_cmp[i - 2])
            e.encode_slots(d_in)
            ev_enc[i] = torch.cuda.Event(); ev_enc[i].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_enc[i])
            e.compact(n)
            ev_cmp[i] = torch.cuda.Event(); ev_cmp[i].record(sB)
    cur.wait_stream(sA); cur.wait_stream(sB)
piped(4); torch.cuda.synchronize()
t0 = time.perf_counter(); piped(K); torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / K * 1e3
for e in E:
    assert e.summary.tolist() == [0, 0]
assert torch.equal(E[0].offsets, E[1].offsets) and torch.equal(E[0].out[: int(E[0].offsets[nb])], E[1].out[: int(E[1].offsets[nb])])
print(f"serial {ts:.3f} ms/step  pipelined {tp:.3f} ms/step  ({n / tp / 1e6:.1f} GB/s)")
