"""The N>1 path on CPU: world_size-2 gloo processes run the same scatter / code / gather
driver as the GPU job, with the CPU oracle injected as the local coder, and the root's result
is checked against a single-process run."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN
from oracle import cbind as ox
from redux_amd import dist as rd

import _dist_worker

PARAMS = (8, 30, 32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("key,block_size,world,root", [
    ("canterbury/alice29.txt", 65536, 2, 0),   # 3 blocks over 2 ranks: ragged ranges
    ("canterbury/xargs.1", 65536, 2, 0),       # 1 block: rank 1 gets nothing
    ("calgary/paper1", 4096, 2, 0),            # 13 blocks
    (None, 65536, 2, 0),                       # empty input: one empty block
    (None, 65536, 2, 1),                       # ... owned by a root that is not rank 0
    ("calgary/paper1", 4096, 2, 1),
])
def test_sharded_encode_decode_matches_single_process(key, block_size, world, root, tmp_path):
    path = os.path.join(GOLDEN, "corpora", key) if key else None
    raw = open(path, "rb").read() if path else b""
    result = str(tmp_path / "root.pt")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dist_worker.worker, args=(r, world, port, path, block_size, result, root)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
            pytest.fail("rank did not finish")
        assert p.exitcode == 0
    got = torch.load(result)
    dense, offs, back = got["dense"].numpy().tobytes(), got["offs"].tolist(), got["back"].numpy().tobytes()
    want, _ = ox.compress_blocks(raw, block_size, PARAMS)
    assert [dense[offs[i]: offs[i + 1]] for i in range(len(offs) - 1)] == want
    assert back == raw


def test_shard_ranges():
    assert rd.shard_ranges(65536, 8) == [(i * 8192, (i + 1) * 8192) for i in range(8)]
    assert rd.shard_ranges(3, 2) == [(0, 2), (2, 3)]
    assert rd.shard_ranges(1, 2) == [(0, 1), (1, 1)]
    assert rd.shard_ranges(99, 8)[-1] == (91, 99) and rd.shard_ranges(62, 8)[-1] == (56, 62)
    for n in (0, 1, 7, 62, 99, 131072):
        for w in (1, 2, 4, 8):
            r = rd.shard_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
