"""Worker side of tests/test_dist_cpu.py (importable in a spawned process on its own)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import cbind as ox  # noqa: E402
from redux_amd import dist as rd  # noqa: E402

PARAMS = (8, 30, 32)


def oracle_encode_local(t, block_size):
    streams, status = ox.compress_blocks(t.numpy().tobytes(), block_size, PARAMS)
    assert not status.any()
    offs = torch.zeros(len(streams) + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(torch.tensor([len(s) for s in streams]), 0)
    return torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8), offs


def oracle_decode_local(dense, offs, block_size):
    nb = offs.numel() - 1
    out = torch.zeros(nb * block_size, dtype=torch.uint8)
    sizes = torch.zeros(nb, dtype=torch.int64)
    raw = dense.numpy().tobytes()
    for b in range(nb):
        d, _ = ox.decompress(raw[int(offs[b]): int(offs[b + 1])], PARAMS, cap=block_size + 16)
        if d:
            out[b * block_size: b * block_size + len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8)
        sizes[b] = len(d)
    return out, sizes


def worker(rank, world, port, path, block_size, result_path, root=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = None
        if rank == root:
            raw = open(path, "rb").read() if path else b""
            data = torch.frombuffer(bytearray(raw), dtype=torch.uint8) if raw else torch.empty(0, dtype=torch.uint8)
        dense, offs = rd.encode_file_sharded(data, block_size, oracle_encode_local, "cpu", root=root)
        back = rd.decode_file_sharded(dense, offs, block_size, oracle_decode_local, "cpu", root=root)
        if rank == root:
            torch.save({"dense": dense, "offs": offs, "back": back}, result_path)
    finally:
        dist.destroy_process_group()
