"""Sharding the block path over the GPUs of one node (one process per GPU, torch.distributed).

Blocks are independent (fresh Codec + fresh model per block: the reference has no
inter-stream state, src/lib.rs:102-109), so the only communication is moving bytes to and
from the rank that owns the file:

  synthetic / already-distributed input   no collective at all: every rank codes its own
                                          contiguous range (bench.py)
  whole-file runs (BASELINE configs[3])   one exchange each way, sizes first, then EXACT byte counts as grouped
                                          point-to-point transfers (torch.distributed.batch_isend_irecv = grouped
                                          ncclSend / ncclRecv on RCCL): the root's xGMI links carry 1/world of the
                                          file each, in parallel; nothing is padded to the largest range and the
                                          root receives straight into the final buffers (no concatenation)

      encode   root -> r   the bytes of r's contiguous block range (views of the caller's tensor)
               r -> root   its per-block sizes (a small gather), then its compressed bytes, landing at their final
                           place in the dense output
      decode   root -> all the per-block stream sizes (a small broadcast), root -> r the streams of r's range
               r -> root   its decoded sizes (a small gather), then its decoded bytes

`encode_local` / `decode_local` do the actual coding on this rank's device; production passes
the DeviceEncoder/DeviceDecoder wrappers below, the CPU tests (gloo, world_size 2) inject the
oracle.  Backend "nccl" is RCCL on ROCm.  Under gloo with tensors on a GPU (bench.py
--rehearse-on-one-gpu: several ranks sharing one card) the transfers are staged through host
memory: gloo's point-to-point operations take CPU tensors only.

`marks` (optional list): (phase name, seconds) pairs are appended as the phases complete -- "scatter", "code",
"gather" -- each after a device synchronisation, for bench.py's whole-file line.
"""
import math
import time

import torch
import torch.distributed as dist


def shard_ranges(nblocks, world):
    """Contiguous block ranges, ceil(nblocks/world) per rank (the last ones may be empty)."""
    per = math.ceil(nblocks / world) if nblocks else 0
    return [(min(r * per, nblocks), min((r + 1) * per, nblocks)) for r in range(world)]


def block_count(nbytes, block_size):
    return 1 if nbytes == 0 else (nbytes + block_size - 1) // block_size


def _world(group):
    return dist.get_rank(group), dist.get_world_size(group)


def _peer(group, r):
    """Global rank of group rank r (send / recv address peers by global rank)."""
    return dist.get_global_rank(group, r) if group is not None else r


class _Comm:
    """Where the bytes of a transfer live: the coder's device, or host memory when the backend cannot move device
    tensors point to point (gloo)."""

    def __init__(self, device, group):
        self.device = torch.device(device)
        self.group = group
        self.staged = self.device.type != "cpu" and dist.get_backend(group) == "gloo"
        self.dev = torch.device("cpu") if self.staged else self.device

    def out(self, t):      # a tensor about to be sent
        return t.cpu() if self.staged else t

    def empty(self, n, dtype=torch.uint8):
        return torch.empty(n, dtype=dtype, device=self.dev)

    def back(self, t):     # a received tensor, to the coder's device
        return t.to(self.device) if self.staged else t

    def exchange(self, ops):
        """Grouped sends / receives of exact sizes; returns when they are complete on this rank."""
        if not ops:
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def mark(self, marks, name, t0):
        if marks is None:
            return t0
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        marks.append((name, t1 - t0))
        return t1


def _byte_ranges(n, block_size, ranges):
    return [(min(n, b0 * block_size), min(n, b1 * block_size)) if b1 > b0 else (0, 0) for b0, b1 in ranges]


def encode_file_sharded(data, block_size, encode_local, device, group=None, root=0, marks=None):
    """Root passes the file as a uint8 tensor on `device` (other ranks pass None).
    encode_local(uint8 tensor, block_size) -> (dense uint8 tensor, int64 offsets[nb+1]) on `device`; the returned tensors
    only have to stay valid until encode_local is called again (they are sent, or copied, before this function returns).
    Returns (dense streams, int64 offsets[nblocks+1]) on the root, (None, None) elsewhere."""
    rank, world = _world(group)
    C = _Comm(device, group)
    t0 = time.perf_counter()
    hdr = torch.zeros(1, dtype=torch.int64, device=C.dev) # (small metadata lives where the backend can move it)
    if rank == root:
        hdr[0] = data.numel()
    dist.broadcast(hdr, src=_peer(group, root), group=group)
    n = int(hdr.item())
    nblocks = block_count(n, block_size)
    ranges = shard_ranges(nblocks, world)
    if n == 0:
        # the empty input is ONE empty block (redux_block_count), and it belongs to the root
        # whichever rank that is: nobody else codes anything
        ranges = [(0, 1) if r == root else (0, 0) for r in range(world)]
    spans = _byte_ranges(n, block_size, ranges)
    per = max(b1 - b0 for b0, b1 in ranges)
    # 1. the root sends every rank the bytes of its range: exact counts, views of the caller's tensor
    lo, hi = spans[rank]
    if rank == root:
        src = C.out(data)
        C.exchange([dist.P2POp(dist.isend, src[spans[r][0]: spans[r][1]], _peer(group, r), group)
                    for r in range(world) if r != root and spans[r][1] > spans[r][0]])
        mine = data[lo:hi]
    else:
        buf = C.empty(hi - lo)
        C.exchange([dist.P2POp(dist.irecv, buf, _peer(group, root), group)] if hi > lo else [])
        mine = C.back(buf)
    t0 = C.mark(marks, "scatter", t0)
    # 2. code this rank's range
    b0, b1 = ranges[rank]
    sizes = torch.zeros(per, dtype=torch.int64, device=C.dev)
    if b1 > b0:
        out, offs = encode_local(mine, block_size)
        sizes[: b1 - b0] = C.out(offs[1:] - offs[:-1])
        total = int(offs[-1].item())
    else:
        out, total = torch.empty(0, dtype=torch.uint8, device=device), 0
    t0 = C.mark(marks, "code", t0)
    # 3. per-block sizes to the root (small), then the compressed bytes, each range straight to its final place
    if rank == root:
        all_sizes = [torch.empty_like(sizes) for _ in range(world)]
        dist.gather(sizes, all_sizes, dst=_peer(group, root), group=group)
        flat = torch.cat([all_sizes[r][: ranges[r][1] - ranges[r][0]] for r in range(world)])
        offsets = torch.zeros(flat.numel() + 1, dtype=torch.int64, device=C.dev)
        offsets[1:] = torch.cumsum(flat, 0)
        oh = offsets.tolist()
        offsets = C.back(offsets)
        place = [(oh[ranges[r][0]], oh[ranges[r][1]]) for r in range(world)]
        dense = C.empty(oh[-1])
        dense[place[root][0]: place[root][1]] = C.out(out[:total])
        C.exchange([dist.P2POp(dist.irecv, dense[place[r][0]: place[r][1]], _peer(group, r), group)
                    for r in range(world) if r != root and place[r][1] > place[r][0]])
        dense = C.back(dense)
        C.mark(marks, "gather", t0)
        return dense, offsets
    dist.gather(sizes, None, dst=_peer(group, root), group=group)
    C.exchange([dist.P2POp(dist.isend, C.out(out[:total]), _peer(group, root), group)] if total else [])
    C.mark(marks, "gather", t0)
    return None, None


def decode_file_sharded(dense, offsets, block_size, decode_local, device, group=None, root=0, marks=None):
    """Inverse of encode_file_sharded.  decode_local(dense uint8, int64 offsets[nb+1], block_size)
    -> (uint8[nb*block_size], int64 sizes[nb]), valid until decode_local is called again.  Returns the decoded file on
    the root."""
    rank, world = _world(group)
    C = _Comm(device, group)
    t0 = time.perf_counter()
    hdr = torch.zeros(1, dtype=torch.int64, device=C.dev)
    if rank == root:
        hdr[0] = offsets.numel() - 1
    dist.broadcast(hdr, src=_peer(group, root), group=group)
    nblocks = int(hdr.item())
    ranges = shard_ranges(nblocks, world)
    per = max(b1 - b0 for b0, b1 in ranges)
    # 1. stream sizes to everyone (small), then the root sends every rank exactly the streams of its range
    sizes = torch.zeros(nblocks, dtype=torch.int64, device=C.dev)
    if rank == root:
        sizes.copy_(C.out(offsets[1:] - offsets[:-1]))
    dist.broadcast(sizes, src=_peer(group, root), group=group)
    b0, b1 = ranges[rank]
    offs = torch.zeros(b1 - b0 + 1, dtype=torch.int64, device=C.dev)
    offs[1:] = torch.cumsum(sizes[b0:b1], 0)
    my_total = int(offs[-1].item())
    offs = C.back(offs)
    if rank == root:
        oh = offsets.tolist()
        src = C.out(dense)
        C.exchange([dist.P2POp(dist.isend, src[oh[ranges[r][0]]: oh[ranges[r][1]]], _peer(group, r), group)
                    for r in range(world) if r != root and oh[ranges[r][1]] > oh[ranges[r][0]]])
        mine = dense[oh[b0]: oh[b1]]
    else:
        buf = C.empty(my_total)
        C.exchange([dist.P2POp(dist.irecv, buf, _peer(group, root), group)] if my_total else [])
        mine = C.back(buf)
    t0 = C.mark(marks, "scatter", t0)
    # 2. decode this rank's range
    osz = torch.zeros(per, dtype=torch.int64, device=C.dev)
    dec = torch.empty(0, dtype=torch.uint8, device=device)
    if b1 > b0:
        dec, dsz = decode_local(mine, offs, block_size)
        osz[: b1 - b0] = C.out(dsz)
    t0 = C.mark(marks, "code", t0)

    def span(nb, sz):  # bytes of a range's output that carry data: whole block slots, the last one as far as it is filled
        return (nb - 1) * block_size + int(sz[nb - 1]) if nb else 0

    # 3. decoded sizes to the root (small), then the decoded bytes: block slots of block_size, the last one cut short
    if rank == root:
        szs = [torch.empty_like(osz) for _ in range(world)]
        dist.gather(osz, szs, dst=_peer(group, root), group=group)
        szh = [s.tolist() for s in szs]
        spans, pos = [], 0
        for r, (c0, c1) in enumerate(ranges):
            spans.append((pos, pos + span(c1 - c0, szh[r])))
            pos = spans[-1][1]
        got = C.empty(pos)
        got[spans[root][0]: spans[root][1]] = C.out(dec[: spans[root][1] - spans[root][0]])
        C.exchange([dist.P2POp(dist.irecv, got[spans[r][0]: spans[r][1]], _peer(group, r), group)
                    for r in range(world) if r != root and spans[r][1] > spans[r][0]])
        got = C.back(got)
        # a file's blocks are all full but the last one: the slots are then the file itself
        flat = [szh[r][i] for r, (c0, c1) in enumerate(ranges) for i in range(c1 - c0)]
        full = all(x == block_size for x in flat[:-1])
        if full:
            C.mark(marks, "gather", t0)
            return got
        parts = []
        for r, (c0, c1) in enumerate(ranges):
            for i in range(c1 - c0):
                parts.append(got[spans[r][0] + i * block_size: spans[r][0] + i * block_size + szh[r][i]])
        res = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=device)
        C.mark(marks, "gather", t0)
        return res
    dist.gather(osz, None, dst=_peer(group, root), group=group)
    n_mine = span(b1 - b0, osz.tolist())
    C.exchange([dist.P2POp(dist.isend, C.out(dec[:n_mine]), _peer(group, root), group)] if n_mine else [])
    C.mark(marks, "gather", t0)
    return None


# ---- production local coders (HBM-resident, the HIP path) ------------------------------------
# A rank codes many ranges of similar size: the coder objects (workspace, bound-sized output, offsets, status) are kept
# per (block size, device) and only replaced by a larger one; one summary read-back per call is the only synchronisation.
# LIFETIME: the tensors a call returns are VIEWS of the cached coder's buffers -- the next call of the same function
# overwrites them (or frees them, when a larger coder replaces the cached one).  encode_file_sharded / decode_file_sharded
# send or copy them before they return; a caller that keeps results across calls must clone them.
def hip_encode_local(params):
    from . import api
    cache = {}

    def f(t, block_size):
        key = (int(block_size), str(t.device))
        enc = cache.get(key)
        if enc is None or enc.max_in_len < t.numel():
            enc = cache[key] = api.DeviceEncoder(params, block_size, max(t.numel(), 1), device=str(t.device))
        out, offs, status, summary = enc.encode(t.contiguous())
        nb = offs.numel() - 1
        st = torch.cat([summary.to(torch.int64), offs[nb: nb + 1]]).tolist()  # (one read-back: summary + total size)
        if st[0] != 0:
            api._raise(st[0])
        return out[: st[2]], offs
    return f


def hip_decode_local(params):
    from . import api
    cache = {}

    def f(dense, offs, block_size):
        nb = offs.numel() - 1
        key = (int(block_size), str(dense.device))
        dec = cache.get(key)
        if dec is None or dec.max_blocks < nb:
            dec = cache[key] = api.DeviceDecoder(params, block_size, nb, device=str(dense.device))
        out, sizes, status, summary = dec.decode(dense.contiguous(), offs.contiguous())
        st = summary.tolist()
        if st[0] != 0:
            api._raise(st[0])
        return out, sizes.to(torch.int64)
    return f
