"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.
Bit-exact: integer/byte work, no tolerance anywhere."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, corpus_files
from oracle import cbind as ox

pytestmark = pytest.mark.gpu

WIDTHS = [(8, 14, 16), (8, 22, 24), (8, 30, 32)]
BLOCK = 65536


@pytest.fixture(scope="module")
def rx():
    import redux_amd
    return redux_amd


def h64(b):
    return hashlib.blake2b(bytes(b), digest_size=8).hexdigest()


def split(out, offs):
    return [out[int(offs[i]): int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]


HAND_TRACED = [
    (b"", (8, 14, 16), "ff00"), (b"", (8, 30, 32), "ff00ff00"),
    (b"\x61", (8, 14, 16), "619d02"), (b"\x61", (8, 30, 32), "619d64970e"),
]


@pytest.mark.parametrize("data,params,hexs", HAND_TRACED)
def test_hand_traced_vectors(rx, data, params, hexs):
    o = io.BytesIO()
    counts = rx.compress(io.BytesIO(data), o, rx.AdaptiveTreeModel.new(rx.Parameters.new(*params)))
    assert o.getvalue().hex() == hexs and counts == (len(data), len(hexs) // 2)
    d = io.BytesIO()
    dc = rx.decompress(io.BytesIO(o.getvalue()), d, rx.AdaptiveTreeModel.new(rx.Parameters.new(*params)))
    assert d.getvalue() == data and dc == (len(hexs) // 2, len(data))


def test_doctest_roundtrip(rx):  # src/lib.rs:23-39
    data = bytes([0x72, 0x65, 0x64, 0x75, 0x78])
    model = rx.AdaptiveTreeModel.new(rx.Parameters.new(8, 14, 16))
    comp = io.BytesIO()
    rx.compress(io.BytesIO(data), comp, model)
    dec = io.BytesIO()
    rx.decompress(io.BytesIO(comp.getvalue()), dec, model)
    assert dec.getvalue() == data
    assert comp.getvalue() == ox.compress(data, (8, 14, 16))[0]


def test_kat_streams(rx):
    for k in json.load(open(os.path.join(GOLDEN, "kat_streams.json"))):
        data = bytes.fromhex(k["input_hex"])
        out, offs, st = rx.compress_blocks(data, max(len(data), 1), tuple(k["params"]))
        assert out.tobytes().hex() == k["stream_hex"], k["name"]


@pytest.mark.parametrize("key,path", corpus_files("artificial", "calgary", "canterbury"))
def test_corpus_blocks_bit_exact_and_roundtrip(rx, key, path):  # BASELINE.json config 3
    data = open(path, "rb").read()
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))[key]
    for w in WIDTHS:
        out, offs, st = rx.compress_blocks(data, BLOCK, w)
        streams = split(out, offs)
        g = gold["%d_%d_%d" % w]
        assert [len(s) for s in streams] == g["block_sizes"], (key, w)
        assert [h64(s) for s in streams] == g["block_hashes"], (key, w)
        dec, sizes, dst = rx.decompress_blocks(out, offs, BLOCK, w)
        got = b"".join(dec[b * BLOCK: b * BLOCK + int(sizes[b])].tobytes() for b in range(len(sizes)))
        assert got == data, (key, w)
    # direct comparison with the oracle (not only hashes) at the CLI's parameters
    want, stw = ox.compress_blocks(data, BLOCK, (8, 30, 32), nthreads=4)
    out, offs, st = rx.compress_blocks(data, BLOCK, (8, 30, 32))
    assert split(out, offs) == want


@pytest.mark.parametrize("key", ["large/bible.txt", "large/world192.txt", "misc/pi.txt"])
def test_large_files_blocks(rx, key):  # BASELINE.json config 4 inputs, single GPU
    data = open(os.path.join(GOLDEN, "corpora", key), "rb").read()
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))[key]["8_30_32"]
    out, offs, st = rx.compress_blocks(data, BLOCK, (8, 30, 32))
    streams = split(out, offs)
    assert [len(s) for s in streams] == gold["block_sizes"]
    assert [h64(s) for s in streams] == gold["block_hashes"]
    dec, sizes, _ = rx.decompress_blocks(out, offs, BLOCK, (8, 30, 32))
    assert b"".join(dec[b * BLOCK: b * BLOCK + int(sizes[b])].tobytes() for b in range(len(sizes))) == data


@pytest.mark.parametrize("key", ["canterbury/alice29.txt", "calgary/geo", "artificial/aaa.txt", "calgary/pic"])
def test_whole_stream_equals_reference_semantics(rx, key):  # config 1: one stream, any length (u32 tree path)
    data = open(os.path.join(GOLDEN, "corpora", key), "rb").read()
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))[key]
    for w in WIDTHS:
        o = io.BytesIO()
        counts = rx.compress(io.BytesIO(data), o, rx.Parameters(*w))
        g = gold["%d_%d_%d" % w]
        assert counts == (len(data), g["whole_size"]) and h64(o.getvalue()) == g["whole_hash"], (key, w)
        d = io.BytesIO()
        dc = rx.decompress(io.BytesIO(o.getvalue()), d, rx.Parameters(*w), max_output=len(data) + 64)
        assert d.getvalue() == data and dc == (g["whole_size"], len(data))


EDGE_LENGTHS = [0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 255, 256, 257, 1000, 4095, 4096, 4097]


@pytest.mark.parametrize("block_size", [16, 48, 1000, 4096, 65536])
def test_ragged_and_small_blocks(rx, block_size):
    rnd = np.random.default_rng(block_size)
    for n in EDGE_LENGTHS + [3 * block_size, 3 * block_size + 1, 5 * block_size - 1]:
        data = rnd.integers(0, 256, n, dtype=np.uint8).tobytes()
        for w in WIDTHS:
            out, offs, st = rx.compress_blocks(data, block_size, w)
            want, _ = ox.compress_blocks(data, block_size, w)
            assert split(out, offs) == want, (n, block_size, w)
            dec, sizes, _ = rx.decompress_blocks(out, offs, block_size, w)
            got = b"".join(dec[b * block_size: b * block_size + int(sizes[b])].tobytes() for b in range(len(sizes)))
            assert got == data


def test_many_ragged_lanes_in_one_wave(rx):
    # 200 blocks of 1000 bytes + a short last block: lanes of one wave finish at different steps
    rnd = np.random.default_rng(7)
    data = rnd.integers(0, 7, 200 * 1000 + 123, dtype=np.uint8).tobytes()
    out, offs, st = rx.compress_blocks(data, 1000, (8, 30, 32))
    want, _ = ox.compress_blocks(data, 1000, (8, 30, 32))
    assert split(out, offs) == want
    dec, sizes, _ = rx.decompress_blocks(out, offs, 1000, (8, 30, 32))
    assert b"".join(dec[b * 1000: b * 1000 + int(sizes[b])].tobytes() for b in range(len(sizes))) == data


STRESS = {
    "all_ff_64k": bytes([255]) * 65536,            # deepest update chain, u16 node reaches 65535
    "all_00_64k": bytes([0]) * 65536,
    "two_symbols": bytes([0, 255]) * 32768,
    "ramp": bytes(range(256)) * 256,
    "skewed": (bytes([7]) * 63 + bytes([200])) * 1024,
    "freeze_then_rare": (bytes([0]) * 16200 + bytes(range(1, 256)) * 194)[:65536],
}


@pytest.mark.parametrize("name", sorted(STRESS))
def test_stress_patterns(rx, name):  # pending-run / carry and freeze corner cases
    data = STRESS[name]
    for w in WIDTHS + [(8, 10, 16), (8, 16, 18), (8, 17, 19)]:
        out, offs, st = rx.compress_blocks(data, BLOCK, w)
        want, _ = ox.compress_blocks(data, BLOCK, w, slot=200000)
        assert split(out, offs) == want, (name, w)
        dec, sizes, _ = rx.decompress_blocks(out, offs, BLOCK, w)
        assert dec[: int(sizes[0])].tobytes() == data, (name, w)


def test_every_byte_value_in_both_lane_halves(rx):
    """The pair kernel's model wave takes its dot-product masks from a table row per byte value (rows s and s + 1; row 256
    is all zero, the node-256 term is added separately) and lanes l / l + 32 own the two halves of every tree dword:
    128 blocks = two full waves in which every lane sees every byte value, runs of 255 and of 0, at its own phase."""
    nb = 128
    i = np.arange(BLOCK, dtype=np.int64)
    blocks = []
    for b in range(nb):
        d = ((i * (2 * b + 1) + 37 * b) & 0xFF).astype(np.uint8)
        d[1000 + 16 * b: 1400 + 16 * b] = 255
        d[30000 + b: 30300 + b] = 0
        d[-(b + 1):] = 255                      # the block's last symbols, whose update is skipped
        blocks.append(d)
    data = np.concatenate(blocks).tobytes()
    for w in (WIDTHS[2], (8, 14, 16)):          # (8,14,16) freezes inside the block: the frozen chunks read the same table
        out, offs, st = rx.compress_blocks(data, BLOCK, w)
        want, _ = ox.compress_blocks(data, BLOCK, w, nthreads=8, slot=200000)
        got = split(out, offs)
        assert len(got) == nb
        for b in range(nb):
            assert got[b] == want[b], (w, b)
        dec, sizes, _ = rx.decompress_blocks(out, offs, BLOCK, w)
        assert all(int(x) == BLOCK for x in sizes) and dec.tobytes() == data


def test_random_param_sweep(rx):
    rnd = np.random.default_rng(99)
    for trial in range(12):
        fb = int(rnd.integers(10, 31))
        cb = int(rnd.integers(fb + 2, min(32, 64 - fb) + 1))
        n = int(rnd.integers(1, 30000))
        alpha = int(rnd.integers(2, 257))
        data = rnd.integers(0, alpha, n, dtype=np.uint8).tobytes()
        bs = int(rnd.choice([512, 4096, 65536]))
        out, offs, st = rx.compress_blocks(data, bs, (8, fb, cb))
        want, _ = ox.compress_blocks(data, bs, (8, fb, cb), slot=4 * bs + 4096)
        assert split(out, offs) == want, (fb, cb, n, bs)
        dec, sizes, _ = rx.decompress_blocks(out, offs, bs, (8, fb, cb))
        assert b"".join(dec[b * bs: b * bs + int(sizes[b])].tobytes() for b in range(len(sizes))) == data


def test_error_paths(rx):
    data = open(os.path.join(GOLDEN, "corpora", "canterbury", "xargs.1"), "rb").read()
    out, offs, st = rx.compress_blocks(data, BLOCK, (8, 30, 32))
    # truncated stream -> Eof (bitio/mod.rs:107 via codec.rs:50)
    cut = np.array([0, int(offs[1]) // 2], dtype=np.uint64)
    dec, sizes, status = rx.decompress_blocks(out[: int(cut[1])], cut, BLOCK, (8, 30, 32), check=False)
    assert status[0] == 1
    with pytest.raises(rx.Eof):
        rx.decompress_blocks(out[: int(cut[1])], cut, BLOCK, (8, 30, 32))
    # stream shorter than code_bits
    dec, sizes, status = rx.decompress_blocks(out[:2], np.array([0, 2], dtype=np.uint64), BLOCK, (8, 30, 32), check=False)
    assert status[0] == 1 and sizes[0] == 0
    # decoded data larger than the block capacity
    dec, sizes, status = rx.decompress_blocks(out, offs, 1024, (8, 30, 32), check=False)
    assert status[0] == 4
    # parameters
    with pytest.raises(rx.InvalidInput):
        rx.compress_blocks(data, BLOCK, (8, 9, 16))
    with pytest.raises(rx.Unsupported):
        rx.compress_blocks(data, BLOCK, (17, 19, 21))


def _oracle_decode_raw(stream, cap, params):
    """ox_decompress without raising: (status, bytes written before the status was decided)."""
    import ctypes as C
    a = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    st = ox.lib().ox_decompress(a.ctypes.data if len(a) else None, len(a), out.ctypes.data, cap, params[0], params[1],
                                params[2], ox.TREE, C.byref(bi), C.byref(bo))
    return st, out[: bo.value].tobytes()


@pytest.mark.parametrize("params", [(8, 30, 32), (8, 14, 16)])
def test_decode_fuzz_matches_oracle(rx, params):
    """Garbage, corrupted, truncated and over-long streams: status, decoded length and decoded
    bytes of every block equal the CPU restatement's (Eof where read_bits fails, bitio/mod.rs:107;
    a symbol is only emitted once its renormalisation has its bits, codec.rs:140-158).  Also the
    decoder's memory-safety net: streams of every length and alignment, lanes finishing at
    different steps, a last stream that ends exactly at the end of the buffer."""
    rnd = np.random.default_rng(20260311)
    cap = 2048
    streams = []
    for n in list(range(0, 24)) + [int(x) for x in rnd.integers(24, 700, 60)]:
        streams.append(rnd.integers(0, 256, n, dtype=np.uint8).tobytes())           # garbage
    streams += [b"\x00" * n for n in (1, 4, 5, 64, 300)] + [b"\xff" * n for n in (1, 4, 7, 64, 300)]
    for i in range(70):                                                                # damaged valid streams
        kind = i % 5
        src = rnd.integers(0, (4, 256, 16, 256, 2)[kind], int(rnd.integers(0, 1500)), dtype=np.uint8).tobytes()
        good, _ = ox.compress(src, params)
        b = bytearray(good)
        if kind == 0 and b:
            b[int(rnd.integers(0, len(b)))] ^= 1 << int(rnd.integers(0, 8))         # one flipped bit
        elif kind == 1:
            b = b[: int(rnd.integers(0, len(b) + 1))]                                 # truncated
        elif kind == 2:
            b += rnd.integers(0, 256, int(rnd.integers(1, 9)), dtype=np.uint8).tobytes()  # trailing bytes
        elif kind == 3 and len(b) > 8:
            j = int(rnd.integers(0, len(b) - 4))
            b[j: j + 4] = rnd.integers(0, 256, 4, dtype=np.uint8).tobytes()         # a damaged dword
        streams.append(bytes(b))                                                      # kind 4: intact
    offs = np.zeros(len(streams) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in streams])
    dense = np.frombuffer(b"".join(streams), dtype=np.uint8)
    dec, sizes, status = rx.decompress_blocks(dense, offs, cap, params, check=False)
    seen = set()
    for b, stream in enumerate(streams):
        st, want = _oracle_decode_raw(stream, cap, params)
        st = 4 if st == 3 else st  # the oracle's writer fails with IoError where the block capacity ends
        seen.add(st)
        assert int(status[b]) == st, (b, len(stream), int(status[b]), st)
        assert int(sizes[b]) == len(want), (b, len(stream), int(sizes[b]), len(want))
        assert dec[b * cap: b * cap + len(want)].tobytes() == want, (b, len(stream))
    assert {0, 1, 4} <= seen  # the mix really exercises Ok, Eof and capacity overflow


ANY_PARAMS = [(4, 10, 16), (12, 14, 16), (8, 24, 40), (1, 3, 5), (16, 18, 20), (2, 30, 34), (8, 30, 33),
              (12, 20, 44), (7, 9, 55), (3, 31, 33), (8, 10, 12)]


@pytest.mark.parametrize("params", ANY_PARAMS)
def test_general_parameters_match_oracle(rx, params):
    """every triple Parameters::new accepts (symbol_bits <= 16) on the device --
    4-/12-/16-bit symbols, code_bits > 32, models that freeze inside a block ((8, 10, 12): freq_max
    1023).  Streams bit-exact with the oracle, decode equal to the oracle's decode (for symbol
    widths that do not divide the input the reference drops the trailing bits, lib.rs:113-120)."""
    rnd = np.random.default_rng(sum(params))
    for bs, n, hi in ((1000, 7013, 256), (4096, 3 * 4096, 4), (333, 2000, 256)):
        data = rnd.integers(0, hi, n, dtype=np.uint8).tobytes()
        out, offs, st = rx.compress_blocks(data, bs, params)
        want, _ = ox.compress_blocks(data, bs, params)
        got = split(out, offs)
        assert len(got) == len(want)
        for b, (g, w) in enumerate(zip(got, want)):
            assert g == w, (params, bs, b, len(g), len(w))
        dec, sizes, status = rx.decompress_blocks(out, offs, bs, params)
        for b, w in enumerate(want):
            stw, outw = _oracle_decode_raw(w, bs, params)
            assert stw == 0 and int(status[b]) == 0
            assert dec[b * bs: b * bs + int(sizes[b])].tobytes() == outw, (params, bs, b)
    # whole-stream drop-ins (one block of any length) and the reader/writer byte counts of lib.rs:108,119
    data = rnd.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    o = io.BytesIO()
    cin, cout = rx.compress(io.BytesIO(data), o, rx.AdaptiveTreeModel(rx.Parameters(*params)))
    w, (wi, wo) = ox.compress(data, params)
    assert o.getvalue() == w and (cin, cout) == (wi, wo)
    # damaged streams: status and partial output as the oracle
    bad = [w[: len(w) // 2], w[:3], b"", bytes(rnd.integers(0, 256, 300, dtype=np.uint8)), w + b"\x00\x01"]
    offs2 = np.zeros(len(bad) + 1, dtype=np.uint64)
    offs2[1:] = np.cumsum([len(x) for x in bad])
    dec, sizes, status = rx.decompress_blocks(np.frombuffer(b"".join(bad), dtype=np.uint8), offs2, 8192, params, check=False)
    for b, stream in enumerate(bad):
        stw, outw = _oracle_decode_raw(stream, 8192, params)
        stw = 4 if stw == 3 else stw
        assert int(status[b]) == stw, (params, b, int(status[b]), stw)
        assert dec[b * 8192: b * 8192 + int(sizes[b])].tobytes() == outw, (params, b)


def test_rcp_f64_error_bound_exhaustive(rx):
    """dec_value (codec.rs:131) multiplies by the raw v_rcp_f64 of `range`, an integer in
    [1, 2^32]; its proof needs |rcp(x)*x - 1| < 2^-24 for every such x.  All 2^32 of them."""
    import ctypes as C
    from redux_amd import _lib
    err = C.c_double()
    assert _lib.lib().redux_debug_rcp_check(1, 1 << 32, C.byref(err)) == 0
    assert 0.0 < err.value < 2.0 ** -24, err.value
    print("max |rcp(x)*x-1| over [1, 2^32]:", err.value)



def test_generators_match_host_definition(rx):
    import torch
    n = 1 << 16

    def splitmix(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        for first in (0, 8 * 12345, 13):
            got = rx.gen_iid(n, 0x5EED0001, first).cpu().numpy()
            j = np.arange(first, first + n, dtype=np.uint64)
            words = splitmix(np.uint64(0x5EED0001) + (j >> np.uint64(3)))
            want = ((words >> ((j & np.uint64(7)) * np.uint64(8))) & np.uint64(0xFF)).astype(np.uint8)
            assert (got == want).all(), first
        got = rx.gen_zipf(n, 0x5EED0005, 777).cpu().numpy()
        j = np.arange(777, 777 + n, dtype=np.uint64)
        u = (splitmix(np.uint64(0x5EED0005) + j) >> np.uint64(32)).astype(np.uint32)
        want = np.searchsorted(rx.zipf_thresholds(), u, side="left").astype(np.uint8)
        assert (got == want).all()
    torch.cuda.synchronize()


@pytest.mark.parametrize("kind,nblocks", [("iid", 4096), ("zipf", 2048)])
def test_device_resident_pipeline(rx, kind, nblocks):
    """Config 2/5 shape at reduced block count, HBM-resident end to end: generate, encode,
    decode, compare on device; a sample of blocks is compared byte-for-byte with the oracle."""
    import torch
    n = nblocks * BLOCK
    d_in = rx.gen_iid(n) if kind == "iid" else rx.gen_zipf(n)
    enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
    out, offs, status, summary = enc.encode(d_in)
    torch.cuda.synchronize()
    assert summary.tolist() == [0, 0]
    offs_h = offs.cpu().numpy().astype(np.uint64)
    total = int(offs_h[-1])
    sizes = np.diff(offs_h.astype(np.int64))
    assert (sizes > 0).all()
    host_in = d_in.cpu().numpy()
    for b in [0, 1, 63, 64, 65, nblocks // 2, nblocks - 1]:
        want, _ = ox.compress(host_in[b * BLOCK:(b + 1) * BLOCK].tobytes(), (8, 30, 32))
        got = out[int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy().tobytes()
        assert got == want, b
    dec = rx.DeviceDecoder((8, 30, 32), BLOCK, nblocks)
    d_out, d_sizes, d_status, d_sum = dec.decode(out[:total], offs)
    torch.cuda.synchronize()
    assert d_sum.tolist() == [0, 0]
    assert bool((d_sizes == BLOCK).all())
    assert torch.equal(d_out, d_in)


@pytest.mark.parametrize("nblocks", [1, 64 * 37 + 5, 64 * 1024 + 64 * 300 + 1])
def test_encoder_role_book_is_returned(rx, nblocks):
    """k_encode_pair books its (model, coder) roles per CU in the workspace and returns the
    booking when the workgroup ends: after any completed encode -- fewer workgroups than the
    chip holds, a partial last group, more than one resident round -- the book reads zero, and
    a second encode on the same workspace produces the same bytes."""
    import ctypes as C
    import torch
    from redux_amd import _lib
    bs = 4096  # small blocks keep the largest case at 350 MB
    n = nblocks * bs - 7
    d_in = rx.gen_zipf(n)
    enc = rx.DeviceEncoder((8, 30, 32), bs, n)
    off, nbytes = C.c_uint64(), C.c_uint64()
    assert _lib.lib().redux_debug_role_book(C.byref(enc.cp), n, bs, C.byref(off), C.byref(nbytes)) == 0
    assert nbytes.value == 8192
    firsts = []
    for _ in range(2):
        out, offs, status, summary = enc.encode(d_in)
        torch.cuda.synchronize()
        assert summary.tolist() == [0, 0]
        book = enc.ws[enc.ws_off + off.value: enc.ws_off + off.value + nbytes.value]
        assert int(book.count_nonzero()) == 0
        firsts.append(out[: int(offs[nblocks])].clone())
    assert torch.equal(firsts[0], firsts[1])
    host = d_in.cpu().numpy()
    offs_h = offs.cpu().numpy()
    for b in sorted({0, nblocks // 2, nblocks - 1}):
        want, _ = ox.compress(host[b * bs:(b + 1) * bs].tobytes(), (8, 30, 32))
        assert firsts[1][int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy().tobytes() == want, b


def _static_tables():
    from test_oracle_codec import static_tables
    return static_tables()


@pytest.mark.parametrize("name", ["flat", "skewed", "wide", "narrow16", "full16", "mid17"])
def test_static_model_matches_oracle(rx, name):
    """the coder core under a fixed frequency table.  Every block's stream equals
    the oracle's compress with the same static model (the codec is the reference's, the model is
    this build's: device == restatement is the bar), and the device decoder inverts it."""
    import torch
    params, cum = _static_tables()[name]
    rng = np.random.default_rng(11)
    for bs, n in ((4096, 4096 * 70 + 123), (65536, 65536 * 3 + 1), (1000, 0), (48, 48 * 129)):
        host = rng.integers(0, 256, n, dtype=np.uint8)
        if n > 5000:
            host[1000:3000] = 255           # the last data symbol, next to EOF in the table
            host[3000:5000] = 0
        d_in = torch.from_numpy(host).cuda()
        coder = rx.DeviceStaticCoder(params, cum, bs, max(n, 1))
        out, offs, status, summary = coder.encode(d_in)
        torch.cuda.synchronize()
        assert summary.tolist() == [0, 0]
        nb = offs.numel() - 1
        offs_h = offs.cpu().numpy()
        out_h = out[: int(offs_h[-1])].cpu().numpy()
        check = range(nb) if nb <= 80 else sorted({0, 1, 63, 64, nb // 2, nb - 2, nb - 1})
        for b in check:
            want, _ = ox.compress_static(host[b * bs:(b + 1) * bs].tobytes(), cum, params)
            assert out_h[int(offs_h[b]): int(offs_h[b + 1])].tobytes() == want, (name, bs, b)
        d_out, d_sizes, d_status, d_sum = coder.decode(out[: int(offs_h[-1])], offs)
        torch.cuda.synchronize()
        assert d_sum.tolist() == [0, 0]
        got = d_out.cpu().numpy()
        sizes = d_sizes.cpu().numpy()
        for b in range(nb):
            lo = b * bs
            ln = min(bs, n - lo) if n else 0
            assert sizes[b] == ln and (got[lo: lo + ln] == host[lo: lo + ln]).all(), (name, bs, b)


def test_static_model_errors(rx):
    import torch
    flat = list(range(258))
    with pytest.raises(rx.InvalidInput):
        rx.DeviceStaticCoder((8, 30, 32), flat[:100] + [flat[99]] + flat[101:], 4096, 4096)
    with pytest.raises(rx.InvalidInput):
        rx.DeviceStaticCoder((8, 14, 16), [i * 100 for i in range(258)], 4096, 4096)  # total > freq_max
    with pytest.raises(rx.Unsupported):
        rx.DeviceStaticCoder((12, 20, 32), flat, 4096, 4096)
    # a truncated stream decodes to Err(Eof) for that block only, as with the adaptive model
    coder = rx.DeviceStaticCoder((8, 30, 32), flat, 4096, 8192)
    d_in = torch.arange(8192, dtype=torch.int32).to(torch.uint8).cuda()
    out, offs, _, _ = coder.encode(d_in)
    torch.cuda.synchronize()
    offs2 = offs.clone()
    cut = offs.clone()
    cut[1] = offs[1] - 40  # block 0 loses its last 40 bytes (block 1 still starts at offs[1])
    streams = torch.cat([out[: int(cut[1])], out[int(offs[1]): int(offs[2])]])
    offs2[1] = cut[1]
    offs2[2] = cut[1] + (offs[2] - offs[1])
    d_out, d_sizes, d_status, d_sum = coder.decode(streams, offs2)
    torch.cuda.synchronize()
    assert d_status.tolist()[0] == 1 and d_status.tolist()[1] == 0  # REDUX_EOF, REDUX_OK
    assert d_sum.tolist() == [1, 1]
    assert torch.equal(d_out[4096:8192], d_in[4096:8192])


def test_decode_full_wave_interval_collapse_narrow_codes(rx):
    """Found by tools/soak_encode.py: with 16-bit codes the interval can collapse to low == high
    (here at symbol 12443 of the block: range 1, all 16 bits shift out).  For codes narrower than
    32 bits that shows as k == code_bits, which the lock-step decoder's unpredicated commit must
    hand to the careful one -- and it only takes the unpredicated commit when all 64 lanes of a
    wave are live, so the wave has to be full."""
    import torch
    blk = np.load(os.path.join(GOLDEN, "soak_block_8_14_16.npy"))
    P = (8, 14, 16)
    want, _ = ox.compress(blk.tobytes(), P, cap=200000)
    for nrep in (64, 67):
        host = np.tile(blk, nrep)
        d_in = torch.from_numpy(host).cuda()
        enc = rx.DeviceEncoder(P, 65536, host.size)
        out, offs, _, summ = enc.encode(d_in)
        torch.cuda.synchronize()
        assert summ.tolist() == [0, 0]
        assert out[: int(offs[1])].cpu().numpy().tobytes() == want
        dec = rx.DeviceDecoder(P, 65536, nrep)
        d_out, d_sizes, d_status, d_sum = dec.decode(out[: int(offs[nrep])], offs)
        torch.cuda.synchronize()
        assert d_sum.tolist() == [0, 0]
        assert torch.equal(d_out, d_in)


def test_eof_is_decided_before_the_capacity_error(rx):
    """Found by tools/soak_decode.py: a damaged stream that fills the block AND runs dry inside the
    renormalisation of the next symbol.  decompress_symbol (codec.rs:123-161) fails with Eof before
    decompress_stream gets to write the symbol (codec.rs:171), so the status is Eof, not the
    capacity error -- alone in a wave (careful commit only) and in a full wave."""
    stream = np.load(os.path.join(GOLDEN, "soak_stream_8_16_32_eof_at_capacity.npy")).tobytes()
    P, cap = (8, 16, 32), 8192
    st, want = _oracle_decode_raw(stream, cap, P)
    assert st == 1 and len(want) == cap
    for nrep in (1, 64):
        offs = np.arange(nrep + 1, dtype=np.uint64) * len(stream)
        dec, sizes, status = rx.decompress_blocks(np.frombuffer(stream * nrep, dtype=np.uint8), offs, cap, P, check=False)
        assert status.tolist() == [1] * nrep and sizes.tolist() == [cap] * nrep
        assert dec[:cap].tobytes() == want and dec[(nrep - 1) * cap:].tobytes() == want


def test_device_entry_points_capture_into_a_hip_graph(rx):
    """The _dev entry points only enqueue work on the caller's stream (no allocation, no
    synchronisation), so an encode + decode pass can be captured once and replayed as a hipGraph
    on new input bytes in the same buffers."""
    import torch
    nblocks = 300
    n = nblocks * BLOCK - 1234
    d_in = rx.gen_zipf(n, seed=1)
    enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
    dec = rx.DeviceDecoder((8, 30, 32), BLOCK, nblocks)
    enc.encode(d_in)  # warm-up outside the capture (code objects, torch's allocator)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            enc.encode_slots(d_in)
            enc.compact(n)
    torch.cuda.current_stream().wait_stream(side)
    for seed in (2, 3):
        rx.gen_zipf(n, seed=seed, out=d_in)      # new bytes, same buffer
        enc.summary.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert enc.summary.tolist() == [0, 0]
        offs = enc.offsets[: nblocks + 1]
        total = int(offs[nblocks])
        host = d_in.cpu().numpy()
        for b in (0, 63, 64, nblocks - 1):
            want, _ = ox.compress(host[b * BLOCK:(b + 1) * BLOCK].tobytes(), (8, 30, 32))
            assert enc.out[int(offs[b]): int(offs[b + 1])].cpu().numpy().tobytes() == want, (seed, b)
        d_out, d_sizes, d_status, d_sum = dec.decode(enc.out[:total], offs)
        torch.cuda.synchronize()
        assert d_sum.tolist() == [0, 0] and torch.equal(d_out[:n], d_in)


@pytest.mark.parametrize("nblocks", [4096, 8192 + 4096, 5000])
def test_scan_offsets_and_status_summary(rx, nblocks):
    """sizes -> offsets + status summary, on the coalesced kernel (whole chunks of 4096 blocks) and
    on the general one: offsets are the exclusive scan of the block sizes, and statuses planted
    between the two phases come back as (first failing status, number of failing blocks)."""
    import torch
    bs = 64
    n = nblocks * bs
    d_in = rx.gen_zipf(n, seed=nblocks)
    enc = rx.DeviceEncoder((8, 30, 32), bs, n)
    enc.encode_slots(d_in)
    enc.compact(n)
    torch.cuda.synchronize()
    offs = enc.offsets[: nblocks + 1].cpu().numpy()
    assert enc.summary.tolist() == [0, 0] and offs[0] == 0
    host = d_in.cpu().numpy()
    sizes = np.diff(offs)
    for b in (0, 1, 1023, 1024, 4095, nblocks - 1):
        want, _ = ox.compress(host[b * bs:(b + 1) * bs].tobytes(), (8, 30, 32))
        assert sizes[b] == len(want)
        assert enc.out[int(offs[b]): int(offs[b + 1])].cpu().numpy().tobytes() == want
    enc.encode_slots(d_in)
    bad = sorted({nblocks - 1, 4095, 2000, 77})
    enc.status[bad[1:]] = 3
    enc.status[bad[0]] = 1   # the first failing block decides summary[0]
    enc.compact(n)
    torch.cuda.synchronize()
    assert enc.summary.tolist() == [1, len(bad)]
    assert (enc.offsets[: nblocks + 1].cpu().numpy() == offs).all()


def test_dense_output_too_small_is_reported_not_overrun(rx):
    import ctypes as C
    import torch
    from redux_amd import _lib
    n = 8 * BLOCK
    d_in = rx.gen_iid(n)
    enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
    guard = torch.full((1024,), 0xAB, dtype=torch.uint8, device="cuda:0")
    small = torch.cat([torch.zeros(3 * BLOCK, dtype=torch.uint8, device="cuda:0"), guard])
    st = _lib.lib().redux_encode_blocks_dev(C.byref(enc.cp), C.c_void_p(d_in.data_ptr()), n, BLOCK,
                                            C.c_void_p(small.data_ptr()), 3 * BLOCK, C.c_void_p(enc.offsets.data_ptr()),
                                            C.c_void_p(enc.status.data_ptr()), C.c_void_p(enc.summary.data_ptr()),
                                            enc._ws_ptr(), enc.ws_bytes, None)
    torch.cuda.synchronize()
    assert st == 0
    assert enc.summary[0].item() == 4 and enc.summary[1].item() >= 5
    assert bool((small[3 * BLOCK:] == 0xAB).all())


@pytest.mark.parametrize("params,bs", [((8, 30, 32), 4096), ((8, 30, 32), 1000), ((8, 14, 16), 4090), ((8, 30, 32), 37)])
def test_decoder_stays_inside_its_blocks(rx, params, bs):
    """Guard bands: streams that decode to more symbols than a block holds, garbage and intact ones,
    in full waves (the staged 16-byte / 4-byte stores of the lock-step decoder) and with block
    sizes that are not store-aligned.  Nothing is written behind the last block, and a block that
    overflows or fails leaves every OTHER block's bytes exactly as an all-intact decode gives them."""
    import torch
    rnd = np.random.default_rng(bs)
    nb = 130
    host = rnd.integers(0, 4, nb * bs, dtype=np.uint8)           # compressible: short streams
    long_src = rnd.integers(0, 4, 3 * bs, dtype=np.uint8)        # one stream of 3 blocks' worth of symbols
    out, offs, _ = rx.compress_blocks(host, bs, params)
    streams = [out[int(offs[b]): int(offs[b + 1])].tobytes() for b in range(nb)]
    too_long, _ = ox.compress(long_src.tobytes(), params)
    bad = {5: too_long, 64: too_long, 70: rnd.integers(0, 256, 600, dtype=np.uint8).tobytes(), 129: too_long, 100: b""}
    mixed = [bad.get(b, streams[b]) for b in range(nb)]
    offs2 = np.zeros(nb + 1, dtype=np.int64)
    offs2[1:] = np.cumsum([len(x) for x in mixed])
    d_streams = torch.from_numpy(np.frombuffer(b"".join(mixed), dtype=np.uint8).copy()).cuda()
    d_offs = torch.from_numpy(offs2).cuda()
    dec = rx.DeviceDecoder(params, bs, nb)
    big = torch.full((nb * bs + 4096,), 0xAB, dtype=torch.uint8, device="cuda:0")
    dec.out = big[: nb * bs]
    d_out, d_sizes, d_status, d_sum = dec.decode(d_streams, d_offs)
    torch.cuda.synchronize()
    assert bool((big[nb * bs:] == 0xAB).all())                    # nothing behind the last block
    got = d_out.cpu().numpy()
    st = d_status.cpu().numpy()
    assert st[5] == 4 and st[64] == 4 and st[129] == 4 and st[100] != 0
    for b in range(nb):
        if b not in bad:
            assert st[b] == 0 and int(d_sizes[b]) == bs
            assert (got[b * bs:(b + 1) * bs] == host[b * bs:(b + 1) * bs]).all(), b
    for b in (5, 64, 129):                                         # a full block of the long stream's symbols
        assert int(d_sizes[b]) == bs and (got[b * bs:(b + 1) * bs] == long_src[:bs]).all(), b


def test_cpp_host_mirror_end_to_end(rx, tmp_path):
    """The C++ mirror of the reference API (redux_amd/host/redux.hpp) through the C ABI on the GPU:
    doc-test, corpus round trip at three widths, Eof on a truncated stream."""
    import subprocess
    from test_abi_cpu import build_host_mirror_test
    exe = build_host_mirror_test(tmp_path)
    out = subprocess.run([exe, os.path.join(GOLDEN, "corpora", "canterbury", "alice29.txt")], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout + out.stderr


def test_sharded_driver_on_gpu_world1(rx):
    """BASELINE.json configs[3] driver (scatter / code / gather over RCCL) with the HIP local
    coders, world_size 1 (the only size a one-GPU box can run), on resources/large/bible.txt."""
    import torch
    import torch.distributed as dist
    from redux_amd import dist as rd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        raw = open(os.path.join(GOLDEN, "corpora", "large", "bible.txt"), "rb").read()
        data = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        dense, offs = rd.encode_file_sharded(data, BLOCK, rd.hip_encode_local((8, 30, 32)), "cuda:0")
        gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))["large/bible.txt"]["8_30_32"]
        o = offs.cpu().numpy()
        d = dense.cpu().numpy()
        assert list(np.diff(o)) == gold["block_sizes"]
        assert [h64(d[int(o[i]): int(o[i + 1])]) for i in range(len(o) - 1)] == gold["block_hashes"]
        back = rd.decode_file_sharded(dense, offs, BLOCK, rd.hip_decode_local((8, 30, 32)), "cuda:0")
        assert back.cpu().numpy().tobytes() == raw
        # the local coders hand out VIEWS of their cached buffers, valid until their next call (redux_amd/dist.py): a result that
        # is to outlive the next call is cloned, and calling again with the first input gives the first result again
        f = rd.hip_encode_local((8, 30, 32))
        a, b = data[: 5 * BLOCK + 123], data[7 * BLOCK: 9 * BLOCK]
        o1, f1 = f(a, BLOCK)
        o1, f1 = o1.clone(), f1.clone()
        f(b, BLOCK)
        o3, f3 = f(a, BLOCK)
        assert torch.equal(o1, o3) and torch.equal(f1, f3[: f1.numel()])
    finally:
        dist.destroy_process_group()


def _adversarial_cases():
    import random
    adv = os.path.join(GOLDEN, "adversarial")
    rnd = random.Random(77)
    out = []
    for f in sorted(os.listdir(adv)):
        if f.endswith(".bin"):
            base = open(os.path.join(adv, f), "rb").read()
            params = tuple(int(x) for x in f[:-4].split("_")[-3:])
            tail = bytes(rnd.randrange(256) for _ in range(300))
            out.append((f, params, base))
            out.append((f + "+tail", params, base + tail))
    return out


@pytest.mark.parametrize("name,params,data", _adversarial_cases())
def test_adversarial_rare_paths(rx, name, params, data):
    """Forces the data-dependent rare paths: pending runs of hundreds of bits (flushed at EOF
    and, with a random tail, inside the unrolled loop) and low == high after narrowing
    (k = code_bits).  Every lane of a wave gets the input so the wave-level ballots fire."""
    blocks = data + bytes((-len(data)) % 16)           # 16-byte multiple so that all blocks are equal
    bs = len(blocks)
    many = blocks * 5 + data                            # 5 identical full blocks + the exact input as a ragged last one
    for w in {params, (8, 30, 32)}:
        out, offs, st = rx.compress_blocks(many, bs, w)
        want, _ = ox.compress_blocks(many, bs, w, slot=4 * bs + 4096)
        assert split(out, offs) == want, (name, w)
        dec, sizes, _ = rx.decompress_blocks(out, offs, bs, w)
        got = b"".join(dec[b * bs: b * bs + int(sizes[b])].tobytes() for b in range(len(sizes)))
        assert got == many, (name, w)


def test_decompress_counts_ignore_trailing_bytes(rx):
    """decompress returns (bytes the reader fetched, bytes written) (src/lib.rs:119): bytes after
    the end of the stream are never read, exactly like the oracle (and tests/corpora.rs:40-41)."""
    data = open(os.path.join(GOLDEN, "corpora", "canterbury", "xargs.1"), "rb").read()
    for w in WIDTHS:
        s, _ = ox.compress(data, w)
        want, wc = ox.decompress(s + b"\x00\x00junk-after-the-stream", w)
        out = io.BytesIO()
        got = rx.decompress(io.BytesIO(s + b"\x00\x00junk-after-the-stream"), out, rx.Parameters(*w), max_output=len(data) + 64)
        assert out.getvalue() == data == want and got == wc == (len(s), len(data))


def test_container_and_cli_end_to_end(rx, tmp_path):
    """the block container round-trips, every payload equals the oracle's
    per-block stream, and the CLI keeps the reference's behaviour (src/main.rs): raw mode is
    byte-identical to redux::compress, the stderr summary and exit codes match."""
    import subprocess
    import sys
    from redux_amd import container
    src = os.path.join(GOLDEN, "corpora", "canterbury", "alice29.txt")
    data = open(src, "rb").read()
    blob = container.compress_bytes(data, BLOCK, (8, 30, 32))
    P, bs, total, offs, payload = container.unpack(blob)
    want, _ = ox.compress_blocks(data, BLOCK, (8, 30, 32))
    assert split(payload, offs) == want and total == len(data) and bs == BLOCK
    assert container.decompress_bytes(blob) == data
    assert container.decompress_bytes(container.compress_bytes(b"", BLOCK)) == b""

    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    run = lambda *a: subprocess.run([sys.executable, "-m", "redux_amd.cli", *a], capture_output=True, env=env, timeout=120)
    last = lambda r: r.stderr.decode().strip().splitlines()[-1]  # the GPU box's libdrm may print a warning first
    raw, blk, back1, back2 = (str(tmp_path / n) for n in ("raw.rdx", "blk.rdx", "b1", "b2"))
    r = run("-c", "-i", src, "-o", raw)                       # reference-compatible single stream
    whole, _ = ox.compress(data, (8, 30, 32))
    assert r.returncode == 0 and open(raw, "rb").read() == whole
    assert last(r) == "Compressed %d bytes into %d bytes, ratio: %.3f" % (len(data), len(whole), len(data) / len(whole))
    r = run("-d", "-i", raw, "-o", back1)
    assert r.returncode == 0 and open(back1, "rb").read() == data
    assert last(r) == "Decompressed %d bytes from %d bytes, ratio: %.3f" % (len(data), len(whole), len(data) / len(whole))
    r = run("-c", "-i", src, "-o", blk, "--block-size", "65536")
    assert r.returncode == 0 and open(blk, "rb").read() == blob
    r = run("-d", "-i", blk, "-o", back2)
    assert r.returncode == 0 and open(back2, "rb").read() == data
    r = run("-d", "-i", src, "-o", back2)                       # not a stream: Eof or garbage-but-no-crash
    assert r.returncode in (0, 3)
    open(raw, "wb").write(whole[: len(whole) // 2])
    r = run("-d", "-i", raw, "-o", back1)
    assert r.returncode == 3 and last(r) == "Decompression error: Unexpected end of file"
    r = subprocess.run([sys.executable, "-m", "redux_amd.cli", "-c"], input=b"redux", capture_output=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout == ox.compress(b"redux", (8, 30, 32))[0]   # stdin -> stdout


@pytest.mark.parametrize("kind", ["iid", "zipf"])
def test_full_size_config_roundtrip(rx, kind):
    """BASELINE.json configs[1] / configs[4] shape at FULL size on one GPU (65,536 blocks of
    64 KiB = 4 GiB): size-independent properties -- no block reports an error, decode(encode(x))
    == x compared on device, offsets are monotone and consistent with the sizes -- plus blocks
    sampled across the whole range compared byte for byte with the oracle."""
    import torch
    nblocks = 65536
    n = nblocks * BLOCK
    d_in = rx.gen_iid(n) if kind == "iid" else rx.gen_zipf(n)
    enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
    out, offs, status, summary = enc.encode(d_in)
    torch.cuda.synchronize()
    assert summary.tolist() == [0, 0]
    sizes = offs[1:] - offs[:-1]
    assert int(offs[0]) == 0 and bool((sizes > 0).all()) and int(sizes.max()) <= 74752
    total = int(offs[-1])
    ratio = total / n
    assert (1.0 < ratio < 1.01) if kind == "iid" else (0.6 < ratio < 0.72), ratio
    offs_h = offs.cpu().numpy()
    for b in [0, 1, 63, 64, 4095, 4096, 32767, 32768, 65471, 65535]:
        blk = d_in[b * BLOCK:(b + 1) * BLOCK].cpu().numpy().tobytes()
        want, _ = ox.compress(blk, (8, 30, 32))
        assert out[int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy().tobytes() == want, b
    dec = rx.DeviceDecoder((8, 30, 32), BLOCK, nblocks)
    d_out, d_sizes, d_status, d_sum = dec.decode(out[:total], offs)
    torch.cuda.synchronize()
    assert d_sum.tolist() == [0, 0] and bool((d_sizes == BLOCK).all())
    assert torch.equal(d_out, d_in)
    del dec, enc, d_in, d_out, out
    torch.cuda.empty_cache()


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _host_zipf(rx, first, n):
    with np.errstate(over="ignore"):
        j = np.arange(first, first + n, dtype=np.uint64)
        u = (_splitmix(np.uint64(0x5EED0005) + j) >> np.uint64(32)).astype(np.uint32)
    return np.searchsorted(rx.zipf_thresholds(), u, side="left").astype(np.uint8)


def _host_iid(first, n):
    with np.errstate(over="ignore"):
        j = np.arange(first, first + n, dtype=np.uint64)
        words = _splitmix(np.uint64(0x5EED0001) + (j >> np.uint64(3)))
        return ((words >> ((j & np.uint64(7)) * np.uint64(8))) & np.uint64(0xFF)).astype(np.uint8)


def test_config4_every_rank_shard_at_its_stated_offset(rx):
    """BASELINE.json configs[4] at its STATED size: the 8 GiB Zipf(1.2) stream is 8 rank shards of
    16,384 blocks, rank r owning stream bytes [r GiB, (r+1) GiB).  One GPU codes the eight shards
    one after the other exactly as rank r would (`first_byte = r << 30`, so shards 4..7 lie beyond
    4 GiB): the generator is checked against its host definition AT that offset, every shard is
    encoded, decoded back on the device and compared, and blocks sampled across the shard are
    compared byte for byte with the oracle."""
    import torch
    nblocks = 16384
    n = nblocks * BLOCK
    enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
    dec = rx.DeviceDecoder((8, 30, 32), BLOCK, nblocks)
    d_in = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    totals = []
    for r in range(8):
        first = r << 30
        rx.gen_zipf(n, 0x5EED0005, first, out=d_in)
        for b in (0, 9999, nblocks - 1):   # the shard really is bytes [first, first + n) of the one 8 GiB stream
            got = d_in[b * BLOCK: b * BLOCK + 4096].cpu().numpy()
            assert (got == _host_zipf(rx, first + b * BLOCK, 4096)).all(), (r, b)
        out, offs, status, summary = enc.encode(d_in)
        torch.cuda.synchronize()
        assert summary.tolist() == [0, 0], r
        offs_h = offs.cpu().numpy()
        total = int(offs_h[-1])
        totals.append(total)
        assert 0.6 < total / n < 0.72, (r, total / n)
        for b in (0, 63, 64, 8191, nblocks - 1):
            blk = d_in[b * BLOCK:(b + 1) * BLOCK].cpu().numpy().tobytes()
            want, _ = ox.compress(blk, (8, 30, 32))
            assert out[int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy().tobytes() == want, (r, b)
        d_out, d_sizes, d_status, d_sum = dec.decode(out[:total], offs)
        torch.cuda.synchronize()
        assert d_sum.tolist() == [0, 0] and bool((d_sizes == BLOCK).all()), r
        assert torch.equal(d_out, d_in), r
    assert len(set(totals)) == 8  # eight different shards, not one shard eight times
    del enc, dec, d_in
    torch.cuda.empty_cache()


@pytest.mark.parametrize("r", [1, 3, 7])
def test_config2_rank_offsets_beyond_4gib(rx, r):
    """bench.py gives rank r the iid stream bytes [r * 4 GiB, (r+1) * 4 GiB): generator and coder at
    those 64-bit offsets (a window of 256 blocks at the start and at the very end of the shard)."""
    import torch
    nb = 256
    for first in (r * (4 << 30), (r + 1) * (4 << 30) - nb * BLOCK):
        d_in = rx.gen_iid(nb * BLOCK, 0x5EED0001, first)
        host = d_in.cpu().numpy()
        assert (host[:8192] == _host_iid(first, 8192)).all() and (host[-8192:] == _host_iid(first + nb * BLOCK - 8192, 8192)).all()
        enc = rx.DeviceEncoder((8, 30, 32), BLOCK, nb * BLOCK)
        out, offs, status, summary = enc.encode(d_in)
        torch.cuda.synchronize()
        assert summary.tolist() == [0, 0]
        offs_h = offs.cpu().numpy()
        for b in (0, 100, nb - 1):
            want, _ = ox.compress(host[b * BLOCK:(b + 1) * BLOCK].tobytes(), (8, 30, 32))
            assert out[int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy().tobytes() == want, (first, b)


def test_decompress_without_max_output_grows_its_buffer(rx):
    """redux::decompress writes to an unbounded io::Write (src/lib.rs:113).  3 MiB of one byte
    value compresses to a few hundred bytes; the default capacity guess (64 x input, at least
    1 MiB) is too small for it and must grow instead of failing."""
    from redux_amd import _lib
    data = b"\0" * (3 << 20)
    model = rx.AdaptiveTreeModel.new(rx.Parameters.new(8, 30, 32))
    c = io.BytesIO()
    bi, bo = rx.compress(io.BytesIO(data), c, model)
    assert bi == len(data) and bo < 4096
    want, _ = ox.compress(data, (8, 30, 32))
    assert c.getvalue() == want
    assert _lib.lib().redux_host_release() == 0   # (what the encoder and earlier tests left in the contexts does not count below)
    d = io.BytesIO()
    assert rx.decompress(io.BytesIO(c.getvalue()), d, model) == (bo, len(data))
    assert d.getvalue() == data
    with pytest.raises(rx.OutputTooSmall):   # an explicit limit is still a limit
        rx.decompress(io.BytesIO(c.getvalue()), io.BytesIO(), model, max_output=1 << 20)
    # the decoder's state is O(1) in the reference (lib.rs:113-120): a generous capacity must not cost memory.  The
    # reciprocal table is a window of 2^20 entries (the 3 MiB above ran two thirds of their steps past it, on computed
    # reciprocals), and what the contexts hold afterwards is the chunk pipeline's own, not gigabytes
    assert _lib.lib().redux_host_resident_bytes() < (256 << 20)
    d = io.BytesIO()
    assert rx.decompress(io.BytesIO(c.getvalue()), d, model, max_output=1 << 30) == (bo, len(data)) and d.getvalue() == data
    assert _lib.lib().redux_host_resident_bytes() < (1 << 30) + (256 << 20)


def test_long_blocks_decode_past_the_reciprocal_window(rx):
    """Blocks of more than 2^20 symbols: the decoders that take them (k_decode_wave in small launches, k_decode otherwise)
    read their reciprocals from a table of 2^20 entries and compute the later ones (rc_lookup): the same values, so the
    same bytes.  1.5 MiB blocks of text and noise, streams from the encoder checked against the oracle first."""
    rng = np.random.default_rng(31)
    text = open(os.path.join(GOLDEN, "corpora", "large", "bible.txt"), "rb").read()
    bs = (3 << 19) + 40
    data = text[: bs + 1000] + bytes((rng.integers(0, 256, 2 * bs - 1000 - 77, dtype=np.uint8) >> 1).tolist())
    for w in ((8, 30, 32), (8, 22, 24)):
        out, offs, st = rx.compress_blocks(data, bs, w)
        assert not st.any() and len(offs) == 4
        for b in range(3):
            assert out[int(offs[b]): int(offs[b + 1])].tobytes() == ox.compress(data[b * bs:(b + 1) * bs], w)[0], (w, b)
        dec, sizes, dst = rx.decompress_blocks(out, offs, bs, w)
        assert not dst.any() and b"".join(dec[b * bs: b * bs + int(sizes[b])].tobytes() for b in range(3)) == data


def test_kernel_names_follow_the_dispatch(rx):
    import ctypes as C
    from redux_amd import _lib
    L = _lib.lib()

    def enc(params, ptr, n, bs):
        p = _lib.Params(*params)
        return L.redux_encode_kernel_name(C.byref(p), C.c_void_p(ptr), n, bs).decode()

    def dec(params, bs, ptr=0):
        p = _lib.Params(*params)
        return L.redux_decode_kernel_name(C.byref(p), C.c_void_p(ptr), bs).decode()

    assert enc((8, 30, 32), 0, 1 << 30, 65536).startswith("k_encode_pair<false, true>")
    assert enc((8, 14, 16), 0, 1 << 30, 65536).startswith("k_encode_pair<false, false>")
    assert enc((8, 30, 32), 4, 1 << 30, 65536).startswith("k_encode<true, false>")     # unaligned input
    assert enc((8, 30, 32), 0, 1 << 35, 1 << 20).startswith("k_encode<false, true>")    # u32 tree: more than 24,576 blocks above 64 KiB
    assert enc((8, 30, 32), 0, 1 << 32, 1 << 20).startswith("k_coop_model")             # ... up to there: coded in windows by the small-grid kernels
    assert enc((8, 30, 32), 0, 1 << 20, 1 << 20).startswith("k_coop_model")             # one block of any length: redux_compress
    assert enc((8, 30, 32), 0, 62 << 16, 65536).startswith("k_coop_model")              # a small launch
    assert enc((12, 14, 16), 0, 1 << 20, 65536).startswith("k_encode_gen_pair<12>")
    assert enc((4, 10, 16), 0, 1 << 20, 65536).startswith("k_encode_gen<4>")
    assert enc((12, 20, 44), 0, 1 << 20, 65536).startswith("k_encode_any")              # code_bits > 32
    assert enc((5, 10, 16), 0, 1 << 20, 65536).startswith("k_encode_gen<5>")           # every width 1 .. 12 has a lock-step kernel
    assert enc((11, 21, 32), 0, 1 << 20, 65536).startswith("k_encode_gen_pair<11>")
    assert enc((13, 21, 32), 0, 1 << 20, 65536).startswith("k_encode_any")
    assert dec((5, 27, 32), 65536).startswith("k_decode_cells<5>")
    assert dec((11, 21, 32), 65536).startswith("k_decode_cells<11>")
    assert dec((4, 28, 32), 65536).startswith("k_decode_cells<4>")
    assert dec((4, 28, 32), 1 << 20).startswith("k_decode_cells<4>")                     # count passes 2^17: the same kernel's fix-up instance
    assert dec((2, 30, 32), 65536).startswith("k_decode_cells<2>") and enc((2, 30, 32), 0, 1 << 20, 65536).startswith("k_encode_gen<2>")
    assert dec((13, 21, 32), 65536).startswith("k_decode_any")
    assert dec((8, 30, 32), 65536).startswith("k_decode_lock<true>")
    assert dec((8, 22, 24), 65536).startswith("k_decode_lock<false>")
    assert dec((8, 30, 32), 1001, 2).startswith("k_decode_lock<true>")
    assert dec((8, 30, 32), 1 << 20).startswith("k_decode_cells<8>")                     # blocks above 64 KiB on a full grid: u32 cells
    assert dec((8, 30, 32), 1 << 23).startswith("k_decode<false, true>")                  # ... above 4 MiB: per-lane control flow
    cp8 = _lib.Params(8, 30, 32)
    name_n = lambda bs, nb: L.redux_decode_kernel_name_n(C.byref(cp8), None, bs, nb).decode()
    assert name_n(1 << 17, 1024).startswith("k_decode_wave") and name_n(1 << 17, 1025).startswith("k_decode_cells<8>")
    assert name_n(1 << 20, 768).startswith("k_decode_wave") and name_n(1 << 20, 769).startswith("k_decode_cells<8>")  # (blocks of 1 MiB and more: from 769)
    assert enc((8, 3, 32), 0, 1, 1) == ""


def test_bench_self_launch_two_ranks_on_one_gpu(rx):
    """`python bench.py --gpus 2` with no RANK in the environment: the parent launches two fresh rank
    processes (here sharing cuda:0 over gloo) and passes rank 0's line through."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(GOLDEN))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--blocks", "2048", "--steps", "2",
                        "--warmup", "1", "--rehearse-on-one-gpu", "--no-decode"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "rehearsal" in line
    # BASELINE configs[3] the same way: bible.txt scattered over two ranks (sizes first, then exact byte counts; gloo stages the
    # transfers through host memory), coded with the HIP kernels, gathered, decoded back, streams against tests/golden/blocks.json
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "file", "--steps", "2", "--warmup", "1",
                        "--rehearse-on-one-gpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["blocks"] == 62 and line["config"]["blocks_per_rank"] == 31
    assert line["roundtrip_equal"] and line["golden"] and all(line["phases"][k] >= 0 for k in ("scatter_ms", "encode_ms", "gather_ms", "decode_ms"))


def test_host_abi_pipeline_many_chunks(rx):
    """The host-pointer calls cut the work into chunks of whole blocks that run on several streams
    (redux_amd/csrc/redux_host.hpp).  Inputs that span several chunks, with a ragged tail, at three
    block sizes: every stream equals the oracle's, offsets are dense, the decode returns the bytes,
    and a repeated call allocates nothing."""
    import ctypes as C
    from redux_amd import _lib
    rng = np.random.default_rng(7)
    text = open(os.path.join(GOLDEN, "corpora", "large", "bible.txt"), "rb").read()
    big = np.concatenate([np.frombuffer(text, dtype=np.uint8)] * 40 + [rng.integers(0, 256, 9_000_001, dtype=np.uint8)])
    for bs, n in ((65536, len(big)), (4096, 70_000_123), (1 << 20, len(big))):   # ~171 MB -> 3 chunks; 4 KiB blocks -> 2 chunks
        data = big[:n]
        out, offs, st = rx.compress_blocks(data, bs, (8, 30, 32))
        nb = len(offs) - 1
        assert nb == (n + bs - 1) // bs and int(offs[0]) == 0 and not st.any()
        sample = sorted(set([0, 1, nb // 3, nb // 2, nb - 2, nb - 1] + list(rng.integers(0, nb, 12))))
        for b in sample:
            want, _ = ox.compress(data[b * bs:(b + 1) * bs].tobytes(), (8, 30, 32))
            assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, (bs, b)
        dec, sizes, dst = rx.decompress_blocks(out, offs, bs, (8, 30, 32))
        assert not dst.any() and int(sizes.sum()) == n
        if n % bs == 0:
            assert (dec == data).all()
        else:
            assert (dec[: (nb - 1) * bs] == data[: (nb - 1) * bs]).all() and (dec[(nb - 1) * bs:][: n - (nb - 1) * bs] == data[(nb - 1) * bs:]).all()
    a0 = _lib.lib().redux_host_allocations()
    out2, offs2, _ = rx.compress_blocks(big, 1 << 20, (8, 30, 32))
    assert _lib.lib().redux_host_allocations() == a0 and (offs2 == offs).all() and (out2 == out).all()


def test_host_abi_from_two_threads_and_release(rx):
    import threading
    from redux_amd import _lib
    text = open(os.path.join(GOLDEN, "corpora", "calgary", "book1"), "rb").read()
    want, _ = ox.compress_blocks(text, BLOCK, (8, 30, 32))
    res = {}

    def work(i):
        out, offs, st = rx.compress_blocks(text, BLOCK, (8, 30, 32))
        dec, sizes, _ = rx.decompress_blocks(out, offs, BLOCK, (8, 30, 32))
        res[i] = (split(out, offs) == want, b"".join(dec[b * BLOCK: b * BLOCK + int(sizes[b])].tobytes() for b in range(len(sizes))) == text)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert res == {i: (True, True) for i in range(4)}
    assert _lib.lib().redux_host_release() == 0       # the context is rebuilt on the next call
    out, offs, st = rx.compress_blocks(text, BLOCK, (8, 30, 32))
    assert split(out, offs) == want


@pytest.mark.parametrize("params", [(4, 10, 16), (4, 22, 24), (4, 28, 32), (12, 14, 16), (12, 18, 30), (12, 20, 32),
                                    (1, 3, 5), (2, 10, 16), (3, 12, 14), (5, 27, 32), (6, 9, 12), (7, 24, 30), (9, 11, 13), (10, 22, 32),
                                    (11, 21, 32), (1, 25, 30), (2, 30, 32), (3, 29, 32)])
def test_lockstep_kernels_for_4_and_12_bit_symbols(rx, params):
    """The widths src/model/tests.rs:95-251 exercises besides 8 (4 and 12) and the ones between, on the lock-step kernels of
    redux_gen.hpp (encode) and redux_decode_cells.hpp (decode):
    64 KiB blocks (131,072 resp. 43,690 symbols: the model freezes inside the block for the narrow frequency
    widths, count passes 2^17 for 4-bit symbols), ragged tail, 70 blocks = two waves, streams bit-exact with the
    oracle, decode equal to the oracle's decode (12-bit symbols: the trailing 8 bits of a 64 KiB block are dropped).  The last
    three triples have 174,762 to 524,288 symbols per block and models that do not freeze: the count passes 2^17 and the
    kernels' fix-up instances run."""
    import ctypes as C
    from redux_amd import _lib
    p = _lib.Params(*params)
    assert _lib.lib().redux_encode_kernel_name(C.byref(p), None, 1 << 20, BLOCK).decode().startswith(("k_encode_gen<%d>" % params[0], "k_encode_gen_pair<%d>" % params[0]))
    assert _lib.lib().redux_decode_kernel_name(C.byref(p), None, BLOCK).decode().startswith(f"k_decode_cells<{params[0]}>")
    rng = np.random.default_rng(params[0] * 100 + params[1])
    text = open(os.path.join(GOLDEN, "corpora", "large", "world192.txt"), "rb").read()
    data = (text + rng.integers(0, 256, 70 * BLOCK - len(text) + 4321, dtype=np.uint8).tobytes())
    data = data[: 69 * BLOCK + 4321]
    out, offs, st = rx.compress_blocks(data, BLOCK, params)
    nb = len(offs) - 1
    assert nb == 70 and not st.any()
    dec, sizes, dst = rx.decompress_blocks(out, offs, BLOCK, params)
    assert not dst.any()
    for b in (0, 1, 36, 37, 63, 64, 68, 69):
        blk = data[b * BLOCK:(b + 1) * BLOCK]
        want, _ = ox.compress(blk, params)
        assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, (params, b)
        stw, outw = _oracle_decode_raw(want, BLOCK, params)
        assert stw == 0 and dec[b * BLOCK: b * BLOCK + int(sizes[b])].tobytes() == outw, (params, b)
        keep = len(blk) * 8 // params[0] * params[0] // 8   # whole symbols, whole bytes
        assert outw == blk[:keep]


# ---- many independent inputs in one call (BASELINE.json configs[2]; reference harness tests/corpora.rs:32-85) ----
def _all_corpus_files():
    return corpus_files("artificial", "calgary", "canterbury", "large", "misc")


@pytest.mark.parametrize("w", WIDTHS)
def test_all_corpus_files_in_one_call_match_the_golden_blocks(rx, w):
    """Every file of every corpus through redux_encode_blocks_v: ONE launch, each file cut into 64 KiB blocks on its own
    (ragged tails, no padding); every block against tests/golden/blocks.json; one redux_decode_blocks_v back."""
    files = _all_corpus_files()
    datas = [open(path, "rb").read() for _, path in files]
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))
    out, offs, st, first = rx.compress_blocks_v(datas, BLOCK, w)
    assert (st == 0).all() and int(first[-1]) == len(offs) - 1
    streams = split(out, offs)
    for i, (key, _) in enumerate(files):
        g = gold[key]["%d_%d_%d" % w]
        mine = streams[int(first[i]): int(first[i + 1])]
        assert [len(s) for s in mine] == g["block_sizes"], (key, w)
        assert [h64(s) for s in mine] == g["block_hashes"], (key, w)
    dec, sizes, dst = rx.decompress_blocks_v(out, offs, [len(d) for d in datas], BLOCK, w)
    assert (dst == 0).all()
    for d, got in zip(datas, dec):
        assert got.tobytes() == d


def test_batch_call_equals_per_input_calls_on_ragged_inputs(rx):
    """Inputs of awkward lengths (0, 1, one block exactly, one block + 1, ...) at a small block size: the batch call must
    give, block for block, what redux_encode_blocks gives for each input alone, and decode back."""
    rng = np.random.default_rng(0x5EED0B)
    bs = 4096
    lens = [0, 1, 15, 16, 17, bs - 1, bs, bs + 1, 3 * bs, 3 * bs + 5, 100, 0, 7 * bs + 4095, 64 * bs, 64 * bs + 1]
    datas = [bytes((rng.integers(0, 256, n, dtype=np.uint8) >> rng.integers(0, 7)).tolist()) for n in lens]
    for w in [(8, 30, 32), (8, 14, 16)]:
        out, offs, st, first = rx.compress_blocks_v(datas, bs, w)
        streams = split(out, offs)
        for i, d in enumerate(datas):
            o1, of1, _ = rx.compress_blocks(d, bs, w)
            assert streams[int(first[i]): int(first[i + 1])] == split(o1, of1), (i, len(d), w)
        dec, sizes, dst = rx.decompress_blocks_v(out, offs, lens, bs, w)
        assert [g.tobytes() for g in dec] == datas


def test_block_table_v_orders_whole_blocks_first(rx):
    tab = rx.block_table_v([0, 1000, 5000], [10, 2 * 4096 + 3, 4096], 4096)
    live = tab[tab["index"] != rx.BLOCK_IDLE]
    assert [int(e["index"]) for e in live] == [1, 2, 4, 0, 3]         # whole blocks in block order, then tails, longest first
    assert [int(e["length"]) for e in live] == [4096, 4096, 4096, 10, 3]
    assert [int(e["offset"]) for e in live] == [1000, 1000 + 4096, 5000, 0, 1000 + 8192]
    # the tails do not share a wave with the whole blocks: idle entries fill the first wave
    assert len(tab) == 66 and (tab["index"][3:64] == 0xFFFFFFFF).all() and int(tab["index"][64]) == 0


def test_batch_decode_reports_damage_per_block_and_stays_inside_each_input(rx):
    """A damaged stream in the middle of a batch: its block reports the error, the other inputs decode, and nothing is
    written outside an input's own out_len bytes (the tail block of an input has less room than block_size)."""
    rng = np.random.default_rng(7)
    bs = 4096
    datas = [bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist()) for n in (5000, 300, 9000)]
    out, offs, st, first = rx.compress_blocks_v(datas, bs, (8, 30, 32))
    bad = out.copy()
    b = int(first[1])                       # the only block of input 1: claim more symbols than it has room for
    other, oo, _ = rx.compress_blocks(bytes(rng.integers(0, 256, 2000, dtype=np.uint8).tolist()), bs, (8, 30, 32))
    streams = split(bad, offs)
    streams[b] = other.tobytes()            # a valid stream of 2000 bytes where 300 are expected
    offs2 = np.zeros(len(offs), dtype=np.uint64)
    offs2[1:] = np.cumsum([len(s) for s in streams])
    dec, sizes, dst = rx.decompress_blocks_v(np.frombuffer(b"".join(streams), dtype=np.uint8), offs2, [len(d) for d in datas], bs,
                                             (8, 30, 32), check=False)
    assert int(dst[b]) == 4 and int(sizes[b]) == 300          # OutputTooSmall, 300 bytes written
    assert dec[0].tobytes() == datas[0] and dec[2].tobytes() == datas[2]


# ---- host pipeline: slot reuse, abort paths, several contexts (redux_amd/csrc/redux_host.hpp) ----
def _mixed_bytes(n, seed):
    rng = np.random.default_rng(seed)
    text = np.frombuffer(open(os.path.join(GOLDEN, "corpora", "large", "bible.txt"), "rb").read(), dtype=np.uint8)
    parts = [text, rng.integers(0, 256, 3_000_001, dtype=np.uint8), (rng.integers(0, 256, 2_000_000, dtype=np.uint8) >> 5)]
    return np.resize(np.concatenate(parts), n)


def test_host_pipeline_slot_reuse_with_small_chunks(rx):
    """Chunk limits shrunk by the test hook: ~100 MB becomes 24 chunks on 8 slots, so every slot, stream, pinned piece and
    the slot's pinned offset mirror are reused three times.  Streams against the single-call result; decode back."""
    from redux_amd import api
    data = _mixed_bytes(100_000_037, 11)
    bs = 65536
    ref_out, ref_offs, _ = rx.compress_blocks(data, bs, (8, 30, 32))
    api.host_set_chunk_bytes(2 << 20, 2 << 20)
    try:
        cb, nc = api.host_chunk_plan((len(data) + bs - 1) // bs, bs, 1)
        assert nc >= 20  # (2 MiB rounds up to a whole wave of 64 blocks: 4 MiB chunks, every slot used three times)
        out, offs, st = rx.compress_blocks(data, bs, (8, 30, 32))
        assert (offs == ref_offs).all() and (out == ref_out).all() and not st.any()
        dec, sizes, dst = rx.decompress_blocks(out, offs, bs, (8, 30, 32))
        assert not dst.any() and (dec[: len(data)] == data).all() and int(sizes.sum()) == len(data)
    finally:
        api.host_set_chunk_bytes(0, 0)


def test_host_pipeline_aborts_cleanly_mid_pipeline(rx):
    """Errors that only show in a LATE chunk, with many chunks in flight: an output buffer that runs out, and offsets that
    stop being monotonic.  The call must come back with the right status and the library must keep working."""
    import ctypes as C
    from redux_amd import _lib, api
    L = _lib.lib()
    data = _mixed_bytes(60_000_000, 12)
    bs = 65536
    out, offs, _ = rx.compress_blocks(data, bs, (8, 30, 32))
    nb = len(offs) - 1
    cp = _lib.Params(8, 30, 32)
    api.host_set_chunk_bytes(2 << 20, 2 << 20)
    try:
        small = np.empty(int(offs[nb // 2]) + 5, dtype=np.uint8)          # room for half of the streams only
        o2 = np.zeros(nb + 1, dtype=np.uint64)
        s2 = np.zeros(nb, dtype=np.int32)
        st = L.redux_encode_blocks(C.byref(cp), data.ctypes.data, len(data), bs, small.ctypes.data, small.size, o2.ctypes.data, s2.ctypes.data)
        assert st == _lib.OUTPUT_TOO_SMALL
        bad = offs.copy()
        bad[nb - 3] = bad[nb - 5]                                         # offsets go backwards in the last chunk
        dec = np.empty(nb * bs, dtype=np.uint8)
        sz = np.zeros(nb, dtype=np.uint32)
        st = L.redux_decode_blocks(C.byref(cp), out.ctypes.data, bad.ctypes.data, nb, bs, dec.ctypes.data, dec.size, sz.ctypes.data, s2.ctypes.data)
        assert st == _lib.INVALID_INPUT
        # and the pipeline is intact afterwards
        out3, offs3, _ = rx.compress_blocks(data, bs, (8, 30, 32))
        assert (offs3 == offs).all() and (out3 == out).all()
    finally:
        api.host_set_chunk_bytes(0, 0)


def test_host_calls_on_two_contexts_of_one_device(rx):
    """redux_host_set_devices([0, 0]): two contexts on the one GPU this box has, chunks dealt round-robin, each context with its
    own issuing and drain threads; the dense output is assembled in chunk order across both.  Same bytes as one context."""
    from redux_amd import api
    data = _mixed_bytes(90_000_011, 13)
    bs = 65536
    ref_out, ref_offs, _ = rx.compress_blocks(data, bs, (8, 30, 32))
    api.host_set_chunk_bytes(4 << 20, 4 << 20)
    try:
        api.host_set_devices([0, 0])
        out, offs, st = rx.compress_blocks(data, bs, (8, 30, 32))
        assert (offs == ref_offs).all() and (out == ref_out).all() and not st.any()
        dec, sizes, dst = rx.decompress_blocks(out, offs, bs, (8, 30, 32))
        assert not dst.any() and (dec[: len(data)] == data).all()
        # three contexts, a call of fewer chunks than contexts, and the batch call (context 0 of the fleet)
        api.host_set_devices([0, 0, 0])
        small = data[: 3 * bs + 17]
        o1, of1, _ = rx.compress_blocks(small, bs, (8, 30, 32))
        assert o1.tobytes() == ref_out[: int(ref_offs[3])].tobytes() + ox.compress(small[3 * bs:].tobytes(), (8, 30, 32))[0]
        o2, of2, _, first = rx.compress_blocks_v([small.tobytes(), b"abc"], bs, (8, 30, 32))
        assert o2[: int(of2[4])].tobytes() == o1.tobytes()
        # the batch calls deal their GROUPS of inputs over the fleet (group k on context k mod 3); with groups of 1 MiB every
        # corpus is several groups, the dense output keeps block order through the ledger of group sizes
        files = [open(p, "rb").read() for _, p in corpus_files("calgary", "canterbury")]
        ref = rx.compress_blocks_v(files, bs, (8, 30, 32))
        api.host_set_chunk_bytes(1 << 20, 1 << 20)
        got = rx.compress_blocks_v(files, bs, (8, 30, 32))
        assert (got[1] == ref[1]).all() and (got[0] == ref[0]).all() and (got[3] == ref[3]).all() and not got[2].any()
        back, bsz, bst = rx.decompress_blocks_v(got[0], got[1], [len(f) for f in files], bs, (8, 30, 32))
        assert not bst.any() and [b.tobytes() for b in back] == files
        api.host_set_chunk_bytes(4 << 20, 4 << 20)
        with pytest.raises(rx.InvalidInput):
            api.host_set_devices([0, 99])
    finally:
        api.host_set_devices([])
        api.host_set_chunk_bytes(0, 0)
    out4, offs4, _ = rx.compress_blocks(data[: 10 * bs], bs, (8, 30, 32))
    assert (offs4 == ref_offs[:11]).all()


def test_4_bit_symbols_in_blocks_past_2_17_symbols(rx):
    """256 KiB blocks of 4-bit symbols: 524,288 symbols, the count passes 2^17 a quarter of the way in (round 3: the per-level
    k_decode_gen<4>; now k_decode_cells<4>'s fix-up instance and k_encode_gen<4>'s fix-up turns).  A whole wave + a ragged one."""
    rng = np.random.default_rng(44)
    bs = 262144
    data = (rng.integers(0, 256, 66 * bs - 777, dtype=np.uint8) >> rng.integers(0, 5)).astype(np.uint8)
    P = (4, 28, 32)
    out, offs, st = rx.compress_blocks(data, bs, P)
    assert not st.any() and len(offs) == 67
    for b in (0, 31, 63, 64, 65):
        assert out[int(offs[b]): int(offs[b + 1])].tobytes() == ox.compress(data[b * bs:(b + 1) * bs].tobytes(), P)[0], b
    dec, sizes, dst = rx.decompress_blocks(out, offs, bs, P)
    assert not dst.any() and int(sizes.sum()) == len(data)
    assert (dec[: 65 * bs] == data[: 65 * bs]).all() and (dec[65 * bs:][: bs - 777] == data[65 * bs:]).all()


def test_gen_kernels_hand_large_blocks_to_the_one_lane_kernels(rx):
    """Blocks beyond what the lock-step kernels for 4- and 12-bit symbols take (4-bit blocks of 12 - 256 MiB used to come
    back Unsupported once 64 slots no longer fitted a 32-bit lane offset).  The limits are now part of the kernel choice -- 4 MiB
    for 4-bit symbols, 63,487 symbols = 95,230 bytes for 12-bit ones (u16 tree nodes holding lowbit + increments) -- and the one-lane kernels code what
    is above them.  Checked here just above the 12-bit limit (the 4-bit one takes a minute on one lane)."""
    from redux_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(5)
    data = (rng.integers(0, 256, 95_232 + 2000, dtype=np.uint8) & 0x3F).tobytes()
    cp = _lib.Params(12, 20, 32)
    assert b"k_encode_any" in _lib.lib().redux_encode_kernel_name(C.byref(cp), None, len(data), len(data))
    assert b"k_encode_gen" in _lib.lib().redux_encode_kernel_name(C.byref(cp), None, 95_230, 95_230)
    cp4 = _lib.Params(4, 22, 24)
    assert b"k_encode_any" in _lib.lib().redux_encode_kernel_name(C.byref(cp4), None, (4 << 20) + 1, (4 << 20) + 1)
    assert b"k_encode_gen" in _lib.lib().redux_encode_kernel_name(C.byref(cp4), None, 4 << 20, 4 << 20)
    out, offs, st = rx.compress_blocks(data, len(data), (12, 20, 32))
    want, _ = ox.compress(data, (12, 20, 32))
    assert out.tobytes() == want
    dec, sizes, dst = rx.decompress_blocks(out, offs, len(data), (12, 20, 32))
    keep = len(data) * 8 // 12 * 12 // 8
    assert int(sizes[0]) == keep and dec[:keep].tobytes() == data[:keep]
    # the case ADVICE r2 named: one block of 4-bit symbols past the lock-step limit (about 50 s on one lane)
    data = (rng.integers(0, 256, (4 << 20) + 4096, dtype=np.uint8) & 0x3F).tobytes()
    out, offs, st = rx.compress_blocks(data, len(data), (4, 22, 24))
    want, _ = ox.compress(data, (4, 22, 24))
    assert out.tobytes() == want
    dec, sizes, dst = rx.decompress_blocks(out, offs, len(data), (4, 22, 24))
    assert int(sizes[0]) == len(data) and dec.tobytes() == data


def test_12_bit_decoder_on_a_large_grid_uses_workspace_trees(rx):
    """A grid of more than one wave of 12-bit symbols: k_decode_cells<12, 64, true> (bottom cells in the workspace, 64 blocks per wave).
    Small blocks keep it quick: streams against the oracle on a sample, everything decoded back."""
    rng = np.random.default_rng(21)
    bs, nb = 96, 16400 + 37
    data = (rng.integers(0, 256, bs * nb - 5, dtype=np.uint8) & rng.integers(1, 256, bs * nb - 5, dtype=np.uint8))
    P = (12, 20, 32)
    out, offs, st = rx.compress_blocks(data, bs, P)
    assert not st.any()
    for b in (0, 1, 8191, 16383, 16384, nb - 1):
        want, _ = ox.compress(data[b * bs:(b + 1) * bs].tobytes(), P)
        assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, b
    dec, sizes, dst = rx.decompress_blocks(out, offs, bs, P)
    assert not dst.any()
    keep = bs * 8 // 12 * 12 // 8
    assert (sizes[:-1] == keep).all()
    got = dec.reshape(nb, bs)[:, :keep]
    assert (got[:-1] == data[: (nb - 1) * bs].reshape(nb - 1, bs)[:, :keep]).all()


def test_small_grid_kernels_on_a_hand_made_table(rx):
    """The small-grid encoder (redux_coop.hpp: a block's model by 64 lanes, chain and bit writer as two waves) on what
    redux_block_table_v never produces: idle entries in FRONT of a wave's blocks, and blocks from 1 to 65,536 bytes in the
    same wave (so the lock-step part is short and thousands of symbols run on the per-lane path).  Every stream equals the
    oracle's; the kernel names say which launches take this path."""
    import ctypes as C
    import torch
    from redux_amd import _lib
    L = _lib.lib()
    BS = 65536
    for w in ((8, 30, 32), (8, 20, 24)):
        cp = _lib.Params(*w)
        assert b"k_coop_model" in L.redux_encode_kernel_name(C.byref(cp), None, 62 * BS, BS)
        assert b"k_encode_pair" in L.redux_encode_kernel_name(C.byref(cp), None, 4096 * BS, BS)
        rng = np.random.default_rng(sum(w))
        lens = [BS, 1, 40000, BS - 1, 63, 64, 65, 1024, 17, BS, 30000, 2, 33, 5000, 0, 16]
        datas = [bytes((rng.integers(0, 256, n, dtype=np.uint8) >> rng.integers(0, 6)).tolist()) for n in lens]
        offs_in, pos = [], 0
        for d in datas:
            offs_in.append(pos)
            pos += (len(d) + 15) & ~15
        total_in = pos + 16
        packed = np.zeros(total_in, dtype=np.uint8)
        for o, d in zip(offs_in, datas):
            packed[o: o + len(d)] = np.frombuffer(d, dtype=np.uint8)
        nb = len(lens)
        # wave 0: three idle entries, then blocks 0..7 in a shuffled order; wave 1: one idle entry, blocks 8..15, idle entries to the end
        order0, order1 = [5, 0, 7, 2, 1, 6, 3, 4], [15, 8, 14, 9, 13, 10, 12, 11]
        tab = np.zeros(128, dtype=rx.BLOCK_DTYPE)
        tab["index"] = rx.BLOCK_IDLE
        for slot, b in list(zip(range(3, 11), order0)) + list(zip(range(65, 73), order1)):
            tab[slot] = (offs_in[b], lens[b], b)
        ne = len(tab)
        d_in = torch.from_numpy(packed).cuda()
        d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
        wsb = L.redux_encode_workspace_bytes(C.byref(cp), ne * BS, BS)
        cap = nb * L.redux_encode_slot_bytes(C.byref(cp), BS)
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device="cuda")
        wsp = ws.data_ptr() + (-ws.data_ptr()) % 256
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
        d_st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        d_sum = torch.zeros(2, dtype=torch.int32, device="cuda")
        strm = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        r = L.redux_encode_blocks_v_dev(C.byref(cp), C.c_void_p(d_in.data_ptr()), total_in, C.c_void_p(d_tab.data_ptr()), ne, nb, BS, 1,
                                        C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_offs.data_ptr()), C.c_void_p(d_st.data_ptr()),
                                        C.c_void_p(d_sum.data_ptr()), C.c_void_p(wsp), wsb, strm)
        torch.cuda.synchronize()
        assert r == 0 and d_sum.tolist() == [0, 0]
        offs = d_offs.cpu().numpy().astype(np.uint64)
        out = d_out.cpu().numpy()
        for b, d in enumerate(datas):
            want, _ = ox.compress(d, w)
            assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, (w, b, len(d))


def test_small_grid_kernels_on_large_blocks(rx):
    """Blocks past 64 KiB on the small-launch kernels: u32 tree nodes in k_coop_model, the quotient fix-up in the chain wave
    (the count passes 2^17 inside a 300,000-byte block), pairs in rows as wide as the launch, a ragged last block; and ONE
    block of 1.5 MB: what redux_compress runs.  Streams equal the oracle's, the decode returns the bytes."""
    import ctypes as C
    from redux_amd import _lib
    rng = np.random.default_rng(77)
    text = open(os.path.join(GOLDEN, "corpora", "large", "world192.txt"), "rb").read()
    for w, bs, n in (((8, 30, 32), 300_000, 1_333_333), ((8, 22, 24), 300_000, 700_001), ((8, 30, 32), 1_500_000, 1_500_000)):
        cp = _lib.Params(*w)
        assert b"k_coop_model" in _lib.lib().redux_encode_kernel_name(C.byref(cp), None, n, bs)
        data = text[: n // 2] + bytes((rng.integers(0, 256, n - n // 2, dtype=np.uint8) >> 2).tolist())
        out, offs, st = rx.compress_blocks(data, bs, w)
        assert not st.any()
        for b in range(len(offs) - 1):
            want, _ = ox.compress(data[b * bs:(b + 1) * bs], w)
            assert out[int(offs[b]): int(offs[b + 1])].tobytes() == want, (w, bs, b)
        dec, sizes, dst = rx.decompress_blocks(out, offs, bs, w)
        assert not dst.any() and b"".join(dec[b * bs: b * bs + int(sizes[b])].tobytes() for b in range(len(sizes))) == data


def _coop_window(nentries, block_size):
    """geometry() of redux_hip.hip: the window (symbols) the small-launch kernels code a launch of large blocks in."""
    w = max(4096, min(65504, (2816 << 20) // 2 // (8 * nentries)))  # (two buffers of pairs)
    if w >= block_size + 1:
        return block_size + 1, 1
    nwin = (block_size + 1 + w - 1) // w
    return ((block_size + 1 + nwin - 1) // nwin + 31) & ~31, nwin


@pytest.mark.parametrize("params,bs,nfull", [((8, 30, 32), 2_500_000, 1), ((8, 20, 24), 2_500_000, 1), ((8, 14, 16), 2_500_000, 0),
                                             ((8, 30, 32), 600_000, 58)])
def test_small_grid_kernels_code_long_blocks_in_windows(rx, params, bs, nfull):
    """Blocks past 64 KiB on the small-launch kernels are coded window by window (EncArgs::win0: the pairs area and the
    reciprocal table hold one window, the coder state and the symbol counts of a block travel in the workspace).  Block
    lengths on and around the window edges -- a block that ends exactly at an edge has its EOF symbol alone in the next
    window --, blocks that end windows before the others (their lanes must leave their finished slots alone: linear slots
    with fewer than 64 entries, row-major group areas with more), a model that freezes inside a later window (20 frequency
    bits) or in the first one (14).  Every stream equals the oracle's; the workspace stays far below the 8 bytes of pairs per
    input byte that whole blocks would take."""
    import ctypes as C
    from redux_amd import _lib
    rng = np.random.default_rng(bs + sum(params))
    L = _lib.lib()
    cp = _lib.Params(*params)

    def lengths(win):
        return [win, win - 1, win + 1, 2 * win, 2 * win + 5, bs, 0, 1, 777, win - 16, win + 17, bs - 1] + [bs] * nfull

    # the table's entries decide the window, the window decides the lengths: settle on a fixed point
    win, nwin = _coop_window(12 + nfull, bs)
    for _ in range(4):
        lens = [n for n in lengths(win) if n <= bs]
        offs = np.cumsum([0] + lens[:-1])
        ne = len(rx.block_table_v(offs, lens, bs))
        win2, nwin = _coop_window(ne, bs)
        if win2 == win:
            break
        win = win2
    assert win2 == win and nwin >= 2, (win, win2, nwin)
    assert b"k_coop_model" in L.redux_encode_kernel_name(C.byref(cp), None, ne * bs, bs)
    # (the slots, and one window of pairs per entry + change: whole blocks would take 8 bytes of pairs per input byte of every entry)
    assert L.redux_encode_workspace_bytes(C.byref(cp), ne * bs, bs) < (ne + 65) * (L.redux_encode_slot_bytes(C.byref(cp), bs) + 256) + 2 * ne * 66000 * 8 + (8 << 20)
    text = open(os.path.join(GOLDEN, "corpora", "large", "world192.txt"), "rb").read()
    datas = []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            datas.append((text * (n // len(text) + 1))[:n])
        else:
            datas.append(bytes((rng.integers(0, 256, n, dtype=np.uint8) >> rng.integers(0, 6)).tolist()))
    out, o, st, first = rx.compress_blocks_v(datas, bs, params)
    assert not st.any() and len(o) - 1 == len(lens)
    for b, d in enumerate(datas):
        if b < 14 or b % 16 == 0:  # (the full blocks behind the first ones: every sixteenth against the oracle, all of them decoded)
            want, _ = ox.compress(d, params)
            assert out[int(o[b]): int(o[b + 1])].tobytes() == want, (params, b, len(d))
    dec, sizes, dst = rx.decompress_blocks_v(out, o, [len(d) for d in datas], bs, params)
    assert not dst.any() and all(x.tobytes() == d for x, d in zip(dec, datas))


def _damaged_large_blocks(rx, params, cap, filler, kernel):
    """Intact, bit-flipped, truncated and over-long streams of large blocks, and garbage, + `filler` short intact streams in
    front of them (what decides which decoder the launch gets): status, decoded length and decoded bytes of every block
    against the CPU restatement's, at a capacity some of them overflow."""
    import ctypes as C
    from redux_amd import _lib
    rnd = np.random.default_rng(sum(params) + 5 + filler)
    fill_src = [rnd.integers(0, 256, int(rnd.integers(0, 40)), dtype=np.uint8).tobytes() for _ in range(16)]
    fill = [(src, ox.compress(src, params)[0]) for src in fill_src]
    streams = [fill[i % 16][1] for i in range(filler)]
    streams += [rnd.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 1, 3, 4, 5, 17, 300, 4001)]  # garbage
    for i in range(20):
        kind = i % 5
        n = int(rnd.integers(cap + 1, cap + 50_000)) if kind == 4 else int(rnd.integers(70_000, cap - 1000))  # kind 4: more symbols than the capacity
        src = (rnd.integers(0, 256, n, dtype=np.uint8) >> int(rnd.integers(0, 7))).tobytes()
        good, _ = ox.compress(src, params)
        b = bytearray(good)
        if kind == 0:
            b[int(rnd.integers(0, len(b)))] ^= 1 << int(rnd.integers(0, 8))         # one flipped bit
        elif kind == 1:
            b = b[: int(rnd.integers(len(b) // 2, len(b) + 1))]                       # truncated
        elif kind == 2:
            b += rnd.integers(0, 256, int(rnd.integers(1, 9)), dtype=np.uint8).tobytes()  # trailing bytes
        streams.append(bytes(b))                                                      # kinds 3, 4: intact (4: may overflow the capacity)
    cp = _lib.Params(*params)
    assert kernel in _lib.lib().redux_decode_kernel_name_n(C.byref(cp), None, cap, len(streams))
    offs = np.zeros(len(streams) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in streams])
    dense = np.frombuffer(b"".join(streams), dtype=np.uint8)
    dec, sizes, status = rx.decompress_blocks(dense, offs, cap, params, check=False)
    seen = set()
    for b, stream in enumerate(streams):
        if b < filler:
            want = fill[b % 16][0]
            assert int(status[b]) == 0 and int(sizes[b]) == len(want) and dec[b * cap: b * cap + len(want)].tobytes() == want, b
            continue
        st, want = _oracle_decode_raw(stream, cap, params)
        st = 4 if st == 3 else st  # the oracle's writer fails with IoError where the block capacity ends
        seen.add(st)
        assert int(status[b]) == st, (b, len(stream), int(status[b]), st)
        assert int(sizes[b]) == len(want), (b, len(stream), int(sizes[b]), len(want))
        assert dec[b * cap: b * cap + len(want)].tobytes() == want, (b, len(stream))
    assert {0, 1, 4} <= seen


@pytest.mark.parametrize("params", [(8, 30, 32), (8, 22, 24), (8, 14, 16)])
def test_wave_decoder_on_damaged_large_blocks(rx, params):
    """k_decode_wave (one block per wave, the model as a cumulative table across the lanes: redux_decode_wave.hpp) takes the
    blocks the lock-step decoder cannot -- capacities past 64 KiB -- in small launches."""
    _damaged_large_blocks(rx, params, 150_000, 0, b"k_decode_wave")


def test_many_blocks_of_a_mebibyte_take_the_cell_decoder(rx):
    """More than 768 blocks of 1 MiB or more: the wave decoder slows down there (547 ns per symbol at 1024 waves over such
    streams against 401 over 128 KiB blocks), so the launch is the cell decoder's from 769 blocks on: the same checks, 800 short
    streams in front of the damaged ones."""
    import ctypes as C
    from redux_amd import _lib
    cp = _lib.Params(8, 30, 32)
    assert b"k_decode_wave" in _lib.lib().redux_decode_kernel_name_n(C.byref(cp), None, 1 << 20, 768)
    _damaged_large_blocks(rx, (8, 30, 32), 1 << 20, 800, b"k_decode_cells<8>")


@pytest.mark.parametrize("params,cap", [((8, 30, 32), 150_000), ((8, 30, 32), 100_000), ((8, 22, 24), 150_000), ((8, 14, 16), 100_000), ((8, 16, 18), 150_000)])
def test_cell_decoder_on_damaged_large_blocks(rx, params, cap):
    """... and in launches of more than 1024 blocks they run on the cell decoder with u32 nodes (k_decode_cells<8>,
    redux_decode_cells.hpp): lock-step, 64 blocks per wave -- its fix-up instance where the count can pass 2^17 (capacity
    150,000 under a model that does not freeze before), the plain one otherwise (capacity 100,000; a model of 14 or 16 frequency
    bits, which freezes inside the block)."""
    _damaged_large_blocks(rx, params, cap, 1100, b"k_decode_cells<8>")


def test_hostile_block_tables_are_rejected_on_the_device(rx):
    """The `_v_dev` calls take their block table from device memory: caller data.  Five hostile tables each way -- an index past
    nblocks, a length above block_size, an offset + length past the buffer (once by overflow of the sum), a duplicated index, an
    offset that breaks the REDUX_V_ALIGNED16 promise -- come back with InvalidInput (2) in the summary, the blocks of the VALID
    entries coded exactly as without the hostile one, and guard bands around every buffer untouched (the reference's surface never
    writes out of bounds: bitio/mod.rs:148-198 returns Err)."""
    import ctypes as C
    import torch
    from redux_amd import _lib
    L = _lib.lib()
    BS, W = 4096, (8, 30, 32)
    cp = _lib.Params(*W)
    rng = np.random.default_rng(99)
    nb = 70
    data = (rng.integers(0, 256, nb * BS, dtype=np.uint8) >> 3)
    good = np.zeros(128, dtype=rx.BLOCK_DTYPE)
    good["index"] = rx.BLOCK_IDLE
    for b in range(nb):
        good[b] = (b * BS, BS, b)
    G = 4096  # guard bytes on either side of every buffer
    strm = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def guarded(nbytes, dtype=torch.uint8):
        raw = torch.full((nbytes + 2 * G,), 0xAB, dtype=torch.uint8, device="cuda")
        return raw, raw[G: G + nbytes]

    def guards_ok(raw, nbytes):
        return bool((raw[:G] == 0xAB).all()) and bool((raw[G + nbytes:] == 0xAB).all())

    def encode(tab, flags=1):
        ne = len(tab)
        wsb = L.redux_encode_workspace_bytes(C.byref(cp), ne * BS, BS)
        cap = nb * L.redux_encode_slot_bytes(C.byref(cp), BS)
        r_in, d_in = guarded(nb * BS)
        d_in.copy_(torch.from_numpy(data))
        r_ws, ws = guarded(wsb)
        r_out, d_out = guarded(cap)
        r_off, d_off = guarded((nb + 1) * 8)
        r_st, d_st = guarded(nb * 4)
        r_sum, d_sum = guarded(8)
        d_sum.zero_()
        d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
        assert ws.data_ptr() % 256 == 0
        r = L.redux_encode_blocks_v_dev(C.byref(cp), C.c_void_p(d_in.data_ptr()), nb * BS, C.c_void_p(d_tab.data_ptr()), ne, nb, BS, flags,
                                        C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_off.data_ptr()), C.c_void_p(d_st.data_ptr()),
                                        C.c_void_p(d_sum.data_ptr()), C.c_void_p(ws.data_ptr()), wsb, strm)
        torch.cuda.synchronize()
        assert r == 0
        for raw, n in ((r_in, nb * BS), (r_ws, wsb), (r_out, cap), (r_off, (nb + 1) * 8), (r_st, nb * 4), (r_sum, 8)):
            assert guards_ok(raw, n)
        assert bool((torch.from_numpy(tab.view(np.uint8).copy()).cuda() == d_tab).all())  # the caller's table is not written
        offs = d_off.view(torch.int64).cpu().numpy()
        return d_sum.view(torch.int32).tolist(), d_st.view(torch.int32).cpu().numpy(), offs, d_out.cpu().numpy()

    summary, st, offs, out = encode(good)
    assert summary == [0, 0] and not st.any()
    want = [out[int(offs[b]): int(offs[b + 1])].tobytes() for b in range(nb)]
    assert want[3] == ox.compress(data[3 * BS: 4 * BS].tobytes(), W)[0]

    def hostile(kind):
        t = good.copy()
        hit = 5  # the entry that is damaged (block 5), or the slot a bad entry is added in
        if kind == "index":
            t[100] = (0, BS, nb + 1000)
        elif kind == "length":
            t[hit]["length"] = BS + 1
        elif kind == "offset":
            t[hit]["offset"] = nb * BS - 100
        elif kind == "overflow":
            t[hit]["offset"] = (1 << 64) - 16
        elif kind == "duplicate":
            t[100] = (7 * BS, BS, 7)
        elif kind == "misaligned":
            t[hit]["offset"] = 5 * BS + 8
        return t

    for kind in ("index", "length", "offset", "overflow", "duplicate", "misaligned"):
        summary, st, offs, out = encode(hostile(kind))
        assert summary[0] == 2 and summary[1] >= 1, (kind, summary)
        lost = {5} if kind in ("length", "offset", "overflow", "misaligned") else set()
        for b in range(nb):
            got = out[int(offs[b]): int(offs[b + 1])].tobytes()
            if b in lost:
                assert st[b] == 2 and got == b""
            else:
                assert st[b] == 0 and got == want[b], (kind, b)

    # ---- decode: entry.offset / entry.length say where a block's output goes
    streams = np.frombuffer(b"".join(want), dtype=np.uint8)
    soffs = np.zeros(nb + 1, dtype=np.int64)
    soffs[1:] = np.cumsum([len(x) for x in want])

    def decode(tab, flags=1):
        ne = len(tab)
        wsb = L.redux_decode_workspace_bytes(C.byref(cp), ne, BS)
        r_in, d_in = guarded(len(streams))
        d_in.copy_(torch.from_numpy(streams.copy()))
        r_ws, ws = guarded(wsb)
        r_out, d_out = guarded(nb * BS)
        r_sz, d_sz = guarded(nb * 4)
        r_st, d_st = guarded(nb * 4)
        r_sum, d_sum = guarded(8)
        d_sum.zero_()
        d_offs = torch.from_numpy(soffs).cuda()
        d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
        r = L.redux_decode_blocks_v_dev(C.byref(cp), C.c_void_p(d_in.data_ptr()), C.c_void_p(d_offs.data_ptr()), C.c_void_p(d_tab.data_ptr()), ne, nb,
                                        BS, flags, C.c_void_p(d_out.data_ptr()), nb * BS, C.c_void_p(d_sz.data_ptr()), C.c_void_p(d_st.data_ptr()),
                                        C.c_void_p(d_sum.data_ptr()), C.c_void_p(ws.data_ptr()), wsb, strm)
        torch.cuda.synchronize()
        assert r == 0
        for raw, n in ((r_in, len(streams)), (r_ws, wsb), (r_out, nb * BS), (r_sz, nb * 4), (r_st, nb * 4), (r_sum, 8)):
            assert guards_ok(raw, n)
        return d_sum.view(torch.int32).tolist(), d_st.view(torch.int32).cpu().numpy(), d_sz.view(torch.int32).cpu().numpy(), d_out.cpu().numpy()

    summary, st, sz, dec = decode(good)
    assert summary == [0, 0] and (dec == data).all()
    for kind in ("index", "length", "offset", "overflow", "duplicate", "misaligned"):
        summary, st, sz, dec = decode(hostile(kind))
        assert summary[0] == 2 and summary[1] >= 1, (kind, summary)
        lost = {5} if kind in ("length", "offset", "overflow", "misaligned") else set()
        for b in range(nb):
            if b in lost:
                assert st[b] == 2 and sz[b] == 0
            else:
                assert st[b] == 0 and sz[b] == BS and (dec[b * BS:(b + 1) * BS] == data[b * BS:(b + 1) * BS]).all(), (kind, b)
