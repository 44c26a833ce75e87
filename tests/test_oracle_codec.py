"""Codec-level pins of the oracle: hand-traced streams, the src/lib.rs:23-39
doc-test, the tests/corpora.rs round-trip contract (identity + returned counts == lengths)
for every corpus file x {Linear, Tree} x freq bits {14,22,30}, linear-stream == tree-stream,
the committed golden tables, and C oracle == independent Python restatement."""
import hashlib
import json
import os
import random

import pytest

from conftest import GOLDEN, corpus_files
from oracle import cbind as ox
from oracle import redux_ref as rr

WIDTHS = [(8, 14, 16), (8, 22, 24), (8, 30, 32)]

HAND_TRACED = [  # traced on paper from codec.rs / adaptive_tree.rs, not produced by any code
    (b"", (8, 14, 16), "ff00", (0, 2)),
    (b"", (8, 30, 32), "ff00ff00", (0, 4)),
    (b"\x61", (8, 14, 16), "619d02", (1, 3)),
    (b"\x61", (8, 30, 32), "619d64970e", (1, 5)),
]


@pytest.mark.parametrize("data,params,hexs,counts", HAND_TRACED)
def test_hand_traced_vectors(data, params, hexs, counts):
    s, c = ox.compress(data, params, ox.TREE)
    assert s.hex() == hexs and c == counts
    s2, c2 = rr.compress(data, rr.AdaptiveTreeModel(rr.Parameters(*params)))
    assert s2.hex() == hexs and c2 == counts
    s3, _ = ox.compress(data, params, ox.LINEAR)
    assert s3.hex() == hexs
    d, dc = ox.decompress(s, params, ox.TREE)
    assert d == data and dc == (counts[1], counts[0])


def test_doctest_roundtrip():  # src/lib.rs:23-39
    data = bytes([0x72, 0x65, 0x64, 0x75, 0x78])
    s, c = ox.compress(data, (8, 14, 16), ox.TREE)
    assert c == (5, len(s))
    d, dc = ox.decompress(s, (8, 14, 16), ox.TREE)
    assert d == data and dc == (len(s), 5)


@pytest.mark.parametrize("key,path", corpus_files("artificial", "calgary", "canterbury", "large", "misc"))
def test_corpus_roundtrip_tree_and_golden(key, path):  # tests/corpora.rs:32-85 (Tree rows)
    data = open(path, "rb").read()
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))[key]
    for w in WIDTHS:
        s, c = ox.compress(data, w, ox.TREE)
        assert c == (len(data), len(s))  # corpora.rs:40-41
        d, dc = ox.decompress(s, w, ox.TREE, cap=len(data) + 16)
        assert d == data and dc == (len(s), len(data))  # corpora.rs:59-61
        g = gold["%d_%d_%d" % w]
        assert len(s) == g["whole_size"]
        assert hashlib.blake2b(s, digest_size=8).hexdigest() == g["whole_hash"]


@pytest.mark.parametrize("key,path", corpus_files("artificial", "calgary", "canterbury"))
def test_corpus_linear_equals_tree(key, path):  # tests/corpora.rs Linear rows + SURVEY 8c cross-check (1)
    data = open(path, "rb").read()[:200000]  # the O(N) linear model is slow; a 200 kB prefix per file
    for w in WIDTHS:
        st, ct = ox.compress(data, w, ox.TREE)
        sl, cl = ox.compress(data, w, ox.LINEAR)
        assert st == sl and ct == cl
        d, _ = ox.decompress(sl, w, ox.LINEAR, cap=len(data) + 16)
        assert d == data


def test_golden_block_tables():
    gold = json.load(open(os.path.join(GOLDEN, "blocks.json")))
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for key in ["canterbury/alice29.txt", "calgary/geo", "artificial/aaa.txt", "calgary/pic"]:
        data = open(os.path.join(GOLDEN, "corpora", key), "rb").read()
        assert hashlib.md5(data).hexdigest() == manifest[key]["md5"]
        for w in WIDTHS:
            streams, status = ox.compress_blocks(data, 65536, w, ox.TREE, nthreads=4)
            assert not status.any()
            g = gold[key]["%d_%d_%d" % w]
            assert [len(s) for s in streams] == g["block_sizes"]
            assert [hashlib.blake2b(s, digest_size=8).hexdigest() for s in streams] == g["block_hashes"]
            # each block is exactly redux::compress(block)
            b0, _ = ox.compress(data[:65536], w, ox.TREE)
            assert streams[0] == b0


def test_kat_streams_file():
    for k in json.load(open(os.path.join(GOLDEN, "kat_streams.json"))):
        data = bytes.fromhex(k["input_hex"])
        s, c = ox.compress(data, tuple(k["params"]), ox.TREE)
        assert s.hex() == k["stream_hex"] and list(c) == k["counts"]


def test_c_equals_python_restatement():
    rnd = random.Random(11)
    alice = open(os.path.join(GOLDEN, "corpora", "canterbury", "alice29.txt"), "rb").read()
    cases = [alice[:3000], bytes(rnd.randrange(256) for _ in range(2500)), b"a" * 3000, bytes(range(256)) * 4,
             bytes(rnd.choice(b"ab") for _ in range(3000))]
    for data in cases:
        for w in WIDTHS + [(8, 10, 16)]:  # (8,10,16) freezes after 766 symbols
            s, c = ox.compress(data, w, ox.TREE)
            s2, c2 = rr.compress(data, rr.AdaptiveTreeModel(rr.Parameters(*w)))
            assert s == s2 and c == c2
            d2, dc2 = rr.decompress(s, rr.AdaptiveTreeModel(rr.Parameters(*w)))
            assert d2 == data and dc2 == (len(s), len(data))


def test_truncated_and_overlong_streams():
    data = open(os.path.join(GOLDEN, "corpora", "canterbury", "xargs.1"), "rb").read()
    s, _ = ox.compress(data, (8, 30, 32))
    with pytest.raises(ox.OracleError) as e:  # bitio/mod.rs:107 via codec.rs:50
        ox.decompress(s[: len(s) // 2], (8, 30, 32))
    assert e.value.status == ox.EOF
    with pytest.raises(ox.OracleError) as e:  # bounded sink => IoError
        ox.decompress(s, (8, 30, 32), cap=100)
    assert e.value.status == ox.IO_ERROR
    d, dc = ox.decompress(s + b"\x00\x00junk", (8, 30, 32))  # trailing bytes are never read
    assert d == data and dc[0] == len(s)


ADV = os.path.join(GOLDEN, "adversarial")


def adversarial_cases():
    rnd = random.Random(77)
    out = []
    for f in sorted(os.listdir(ADV)):
        if not f.endswith(".bin"):
            continue
        base = open(os.path.join(ADV, f), "rb").read()
        params = tuple(int(x) for x in f[:-4].split("_")[-3:])
        tail = bytes(rnd.randrange(256) for _ in range(300))
        out.append((f, params, base))
        out.append((f + "+tail", params, base + tail))
    return out


@pytest.mark.parametrize("name,params,data", adversarial_cases())
def test_adversarial_inputs_c_equals_python(name, params, data):
    """Inputs built to force long pending runs (hundreds of E3 bits) and low == high after
    narrowing (tests/golden/make_adversarial.py): both restatements agree and round-trip."""
    d = data[:4000]
    s, c = ox.compress(d, params, ox.TREE)
    s2, c2 = rr.compress(d, rr.AdaptiveTreeModel(rr.Parameters(*params)))
    assert s == s2 and c == c2
    back, _ = ox.decompress(s, params, ox.TREE, cap=len(d) + 16)
    assert back == d
    sl, _ = ox.compress(d, params, ox.LINEAR)
    assert sl == s


GENERAL_PARAMS = [(4, 10, 16), (12, 14, 16), (8, 24, 40), (1, 3, 5), (16, 18, 20), (2, 30, 34), (8, 30, 33),
                  (12, 20, 44), (7, 9, 55), (3, 31, 33)]


@pytest.mark.parametrize("params", GENERAL_PARAMS)
def test_general_parameters_c_equals_python(params):
    """The two restatements agree outside the CLI's (8, 30, 32) as well: other symbol widths,
    code_bits > 32, linear == tree, and what decompress returns when symbol_bits does not divide
    the input (the trailing bits are dropped, lib.rs:113-120).  These are the streams the GPU's
    general-parameter path (redux_any.hpp) is compared with."""
    rnd = random.Random(sum(params))
    for n, hi in ((0, 256), (1, 256), (700, 256), (1500, 3)):
        data = bytes(rnd.randrange(hi) for _ in range(n))
        s, c = ox.compress(data, params, ox.TREE)
        s2, c2 = rr.compress(data, rr.AdaptiveTreeModel(rr.Parameters(*params)))
        assert s == s2 and c == c2, (params, n)
        sl, cl = ox.compress(data, params, ox.LINEAR)
        assert sl == s and cl == c
        d, dc = ox.decompress(s, params, ox.TREE, cap=len(data) + 8)
        d2, dc2 = rr.decompress(s, rr.AdaptiveTreeModel(rr.Parameters(*params)))
        assert d == d2 and dc == dc2
        whole = (len(data) * 8 // params[0]) * params[0] // 8  # bytes made only of whole symbols
        assert d == data[: len(d)] and len(d) >= whole - 1 and dc[0] == len(s)


# ---- static-table model (not in the reference: C restatement vs Python) --------
def static_tables():
    """name -> (params, cum[258])"""
    import random as _r
    t = {}
    t["flat"] = ((8, 30, 32), list(range(258)))                        # every symbol frequency 1
    rng = _r.Random(7)
    f = [1 + int(4000 * rng.random() ** 6) for _ in range(257)]         # skewed, total ~ 2^16
    t["skewed"] = ((8, 30, 32), [0] + [sum(f[: i + 1]) for i in range(257)])
    g = [1 + (997 * i) % 8191 for i in range(257)]                      # total ~ 2^20: the fix-up division
    t["wide"] = ((8, 30, 32), [0] + [sum(g[: i + 1]) for i in range(257)])
    h = [1 + (i % 7) * 9 for i in range(257)]                           # total < 2^14 - 1 for 16-bit codes
    t["narrow16"] = ((8, 14, 16), [0] + [sum(h[: i + 1]) for i in range(257)])
    # total exactly 2^16: the largest table the device decodes by direct lookup (one byte per code value)
    k = [1 + (i * i) % 499 for i in range(257)]
    k[0] += 65536 - sum(k)
    t["full16"] = ((8, 30, 32), [0] + [sum(k[: i + 1]) for i in range(257)])
    # 2^16 < total < 2^17: too large for the lookup, small enough for the division without fix-up (the Fenwick-form descent)
    m = [1 + (i * 37) % 700 for i in range(257)]
    t["mid17"] = ((8, 30, 32), [0] + [sum(m[: i + 1]) for i in range(257)])
    assert t["full16"][1][257] == 65536 and min(k) >= 1 and 65536 < t["mid17"][1][257] < (1 << 17)
    return t


@pytest.mark.parametrize("name", sorted(static_tables()))
def test_static_model_c_equals_python_and_roundtrips(name):
    params, cum = static_tables()[name]
    rng = random.Random(hash(name) & 0xFFFF)
    P = rr.Parameters(*params)
    for data in (b"", b"a", bytes(rng.randrange(256) for _ in range(700)), b"abracadabra" * 40, bytes([255]) * 300):
        c_stream, c_counts = ox.compress_static(data, cum, params)
        p_stream, p_counts = rr.compress(data, rr.StaticModel(P, cum))
        assert c_stream == p_stream and tuple(c_counts) == tuple(p_counts), name
        back, _ = ox.decompress_static(c_stream, cum, params)
        assert back == data
        assert rr.decompress(c_stream, rr.StaticModel(P, cum))[0] == data


def test_static_model_rejects_bad_tables():
    P = rr.Parameters(8, 14, 16)
    flat = list(range(258))
    for bad in (flat[:-1], [1] + flat[1:], flat[:100] + [flat[99]] + flat[101:], [i * 100 for i in range(258)]):
        with pytest.raises(rr.InvalidInput):
            rr.StaticModel(P, bad)
    with pytest.raises(ox.OracleError):
        ox.compress_static(b"x", [i * 100 for i in range(258)], (8, 14, 16))  # total > freq_max
