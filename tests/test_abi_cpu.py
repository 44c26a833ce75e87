"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/redux_hip.h declares, and its host-only entry points (validation, geometry) behave
like the reference's Parameters::new (src/model/mod.rs:64).  No compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import cbind as ox
from redux_amd import _lib, api


def declared_functions():
    text = open(os.path.join(ROOT, "include", "redux_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(redux_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = declared_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), f"libredux_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"ctypes signature missing for {n}"
    assert b"gfx950" in L.redux_version()


def test_params_check_matches_reference_rule():
    L = _lib.lib()
    for s in range(0, 14):
        for f in range(0, 40, 1):
            for c in (f, f + 1, f + 2, f + 3, 32, 33, 64 - f, 65 - f):
                if c < 0:
                    continue
                want = ox.lib().ox_params_new(s, f, c, C.byref(ox.Params()))
                got = L.redux_params_check(s, f, c)
                assert (got == 0) == (want == 0), (s, f, c)
                assert got in (_lib.OK, _lib.INVALID_INPUT)


def test_device_supports():
    L = _lib.lib()
    assert L.redux_device_supports(C.byref(_lib.Params(8, 30, 32))) == _lib.OK
    assert L.redux_device_supports(C.byref(_lib.Params(8, 14, 16))) == _lib.OK
    # general parameters run on the one-lane-per-block path (redux_any.hpp) ...
    assert L.redux_device_supports(C.byref(_lib.Params(4, 10, 16))) == _lib.OK
    assert L.redux_device_supports(C.byref(_lib.Params(12, 14, 16))) == _lib.OK
    assert L.redux_device_supports(C.byref(_lib.Params(8, 24, 40))) == _lib.OK
    assert L.redux_device_supports(C.byref(_lib.Params(16, 18, 46))) == _lib.OK
    # ... up to 16-bit symbols (a tree of 2^symbol_bits + 2 entries per block)
    assert L.redux_device_supports(C.byref(_lib.Params(17, 19, 21))) == _lib.UNSUPPORTED
    assert L.redux_device_supports(C.byref(_lib.Params(8, 9, 16))) == _lib.INVALID_INPUT


def test_geometry():
    L = _lib.lib()
    assert L.redux_block_count(0, 65536) == 1
    assert L.redux_block_count(1, 65536) == 1
    assert L.redux_block_count(65536, 65536) == 1
    assert L.redux_block_count(65537, 65536) == 2
    assert L.redux_block_count(148481, 65536) == 3
    p = _lib.Params(8, 30, 32)
    slot = L.redux_encode_slot_bytes(C.byref(p), 65536)
    assert slot >= 65536 * 9 // 8 + 8
    assert L.redux_encode_bound(C.byref(p), 148481, 65536) == 3 * slot
    assert L.redux_encode_workspace_bytes(C.byref(p), 148481, 65536) > 3 * slot
    # a model that freezes inside the block can cost freq_bits+1 bits per symbol
    p14 = _lib.Params(8, 14, 16)
    assert L.redux_encode_slot_bytes(C.byref(p14), 65536) >= 65536 * 15 // 8


def test_small_launches_get_a_pairs_area_and_their_own_kernels():
    """The kernel choice is host logic (redux_hip.hip: geometry, pick_encode_kernel): launches of at most 2048 slots of
    u16-node blocks take the small-grid path (redux_coop.hpp) and their workspace holds 8 bytes per input symbol for the
    (low, high) pairs; larger launches, larger blocks and the other widths do not."""
    L = _lib.lib()
    BS = 65536
    for w in ((8, 30, 32), (8, 14, 16), (8, 20, 24)):
        p = _lib.Params(*w)
        small, big = L.redux_encode_workspace_bytes(C.byref(p), 62 * BS, BS), L.redux_encode_workspace_bytes(C.byref(p), 4096 * BS, BS)
        assert small >= 64 * BS * 8 and big < 4096 * BS * 3      # pairs for whole groups of 64 slots / slots only
        assert b"k_coop_model" in L.redux_encode_kernel_name(C.byref(p), None, 62 * BS, BS)
        assert b"k_coop_model" in L.redux_encode_kernel_name(C.byref(p), None, 2048 * BS, BS)
        assert b"k_encode_pair" in L.redux_encode_kernel_name(C.byref(p), None, 2049 * BS, BS)
    p = _lib.Params(8, 30, 32)
    assert b"k_coop" not in L.redux_encode_kernel_name(C.byref(p), None, 10 * 512, 512)          # blocks below 1 KiB
    assert b"k_coop" in L.redux_encode_kernel_name(C.byref(p), None, 1 << 20, 1 << 20)           # ONE block of any length (redux_compress): u32 nodes
    assert L.redux_encode_workspace_bytes(C.byref(p), 1 << 20, 1 << 20) < (1 << 20) * 90         # ... its pairs in rows of one lane, not 64
    # blocks above 64 KiB are coded in windows of at most 65,504 symbols: the pairs area holds one window of every block (two of them, at most
    # 2816 MiB in all), whatever the block length -- 2048 blocks of 1 MiB, and ONE stream of 1 GiB (whole blocks would ask for 16 and
    # 8 GiB of pairs, + 8 GiB of reciprocals) -- and such launches take the small-grid kernels up to 24,576 blocks
    assert b"k_coop" in L.redux_encode_kernel_name(C.byref(p), None, 2048 << 20, 1 << 20)
    slot = L.redux_encode_slot_bytes(C.byref(p), 1 << 20)
    assert L.redux_encode_workspace_bytes(C.byref(p), 2048 << 20, 1 << 20) < 2049 * (slot + 256) + (2100 << 20)
    assert b"k_coop" in L.redux_encode_kernel_name(C.byref(p), None, 1 << 30, 1 << 30)
    assert L.redux_encode_workspace_bytes(C.byref(p), 1 << 30, 1 << 30) < L.redux_encode_slot_bytes(C.byref(p), 1 << 30) + (8 << 20)
    assert b"k_coop" in L.redux_encode_kernel_name(C.byref(p), None, 24576 << 17, 1 << 17)
    assert b"k_coop" not in L.redux_encode_kernel_name(C.byref(p), None, 24577 << 17, 1 << 17)
    assert b"k_coop" not in L.redux_encode_kernel_name(C.byref(_lib.Params(12, 20, 32)), None, 62 * BS, BS)


def test_worst_case_slot_bound_holds_on_cpu_oracle():
    # adversarial input for a frozen model: saturate one symbol, then send only rare ones
    data = bytes([0]) * 16200 + bytes(range(1, 256)) * 193
    data = data[:65536]
    s, _ = ox.compress(data, (8, 14, 16))
    p14 = _lib.Params(8, 14, 16)
    assert len(s) <= _lib.lib().redux_encode_slot_bytes(C.byref(p14), 65536)
    assert len(s) > 65536  # it really expands


def test_parameters_mirror():
    P = api.Parameters(8, 30, 32)
    q = ox.params_new(8, 30, 32)
    for k in ("symbol_bits", "symbol_eof", "symbol_count", "freq_bits", "freq_max", "code_bits", "code_min",
              "code_one_fourth", "code_half", "code_three_fourths", "code_max"):
        assert getattr(P, k) == getattr(q, k)
    with pytest.raises(api.InvalidInput):
        api.Parameters(8, 9, 16)
    assert api.AdaptiveTreeModel.new(P).parameters() is P


def test_zipf_table_properties():
    th = api.zipf_thresholds()
    assert th.dtype == np.uint32 and th.shape == (256,)
    assert (np.diff(th.astype(np.int64)) > 0).all() and th[-1] == 2**32 - 1
    w = np.arange(1, 257, dtype=np.float64) ** -1.2
    cdf = np.cumsum(w) / w.sum()
    assert np.abs(th.astype(np.float64) / 2**32 - cdf).max() < 1e-9


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "redux_amd", "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        api.compress_blocks(b"abc", 65536)


def build_host_mirror_test(tmpdir):
    import subprocess
    exe = os.path.join(str(tmpdir), "host_mirror_test")
    libdir = os.path.join(ROOT, "redux_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L" + libdir, "-lredux_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_mirror_compiles_and_validates(tmp_path):
    """redux_amd/host/redux.hpp (the C++ mirror of the reference's API) builds against the C ABI
    with plain g++ and its host-only checks pass (no GPU call)."""
    import subprocess
    _lib.lib()
    exe = build_host_mirror_test(tmp_path)
    out = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "host-side checks ok" in out.stdout, out.stderr


def test_static_table_check_is_host_side():
    """redux_static_table_check validates the table on the host (no GPU call): strictly increasing,
    cum[0] = 0, total <= freq_max; only the widths of the fast coder core are supported."""
    import ctypes as C
    from redux_amd import _lib
    L = _lib.lib()
    tab = lambda xs: (C.c_uint32 * 258)(*xs)
    flat = list(range(258))
    ok = _lib.Params(8, 30, 32)
    assert L.redux_static_table_check(C.byref(ok), tab(flat)) == _lib.OK
    assert L.redux_static_table_check(C.byref(ok), tab([1] + flat[1:])) == _lib.INVALID_INPUT
    assert L.redux_static_table_check(C.byref(ok), tab(flat[:50] + [flat[49]] + flat[51:])) == _lib.INVALID_INPUT
    small = _lib.Params(8, 14, 16)
    assert L.redux_static_table_check(C.byref(small), tab([i * 100 for i in range(258)])) == _lib.INVALID_INPUT
    assert L.redux_static_table_check(C.byref(_lib.Params(12, 20, 32)), tab(flat)) == _lib.UNSUPPORTED
    assert L.redux_static_encode_workspace_bytes(C.byref(ok), 1 << 20, 65536) > 0
    assert L.redux_static_encode_bound(C.byref(ok), 1 << 20, 65536) >= (1 << 20) * 4


def test_block_table_v_is_host_logic_and_orders_whole_blocks_first():
    # redux_block_table_v / redux_block_count_v (the `_v` calls' geometry): no device call
    tab = api.block_table_v([0, 1000, 5000], [10, 2 * 4096 + 3, 4096], 4096)
    live = tab[tab["index"] != api.BLOCK_IDLE] if "api" in globals() else tab[tab["index"] != 0xFFFFFFFF]
    assert [int(e["index"]) for e in live] == [1, 2, 4, 0, 3]         # whole blocks in block order, then tails, longest first
    assert [int(e["length"]) for e in live] == [4096, 4096, 4096, 10, 3]
    assert [int(e["offset"]) for e in live] == [1000, 1000 + 4096, 5000, 0, 1000 + 8192]
    # the tails do not share a wave with the whole blocks: idle entries fill the first wave
    assert len(tab) == 66 and (tab["index"][3:64] == 0xFFFFFFFF).all() and int(tab["index"][64]) == 0
    L = _lib.lib()
    lens = np.array([0, 1, 4096, 4097, 0], dtype=np.uint64)
    assert L.redux_block_count_v(lens.ctypes.data, 5, 4096) == 1 + 1 + 1 + 2 + 1   # an empty input is one empty block
    assert L.redux_block_count_v(lens.ctypes.data, 5, 0) == 0
    tab = api.block_table_v([0, 0, 0, 0, 0], lens, 4096)
    assert sorted(int(e["index"]) for e in tab if e["index"] != api.BLOCK_IDLE) == list(range(6))
    assert api.BLOCK_DTYPE.itemsize == 16


def test_host_chunk_plan_deals_whole_waves_round_robin():
    # redux_host_chunk_plan: pure host arithmetic (what redux_encode_blocks / redux_decode_blocks do with a fleet of contexts)
    for nblocks, bs, nctx, dec in [(65536, 65536, 1, False), (65536, 65536, 8, False), (65536, 65536, 8, True), (1000, 4096, 2, False),
                                   (1, 65536, 8, False), (16384, 65536, 3, True), (200000, 1024, 4, False)]:
        cb, nc = api.host_chunk_plan(nblocks, bs, nctx, dec)
        assert cb >= 1 and (cb % 64 == 0 or cb == nblocks)          # whole waves of 64 blocks (or the whole small call)
        assert (nc - 1) * cb < nblocks <= nc * cb                    # the chunks cover the blocks exactly
        limit = (256 if dec else 128) << 20
        assert cb * bs <= limit + 64 * bs                            # a chunk's payload stays under the limit (rounded up to a wave)
        if nblocks * bs >= nctx * 8 * (16 << 20):                    # big calls: every context gets chunks, about eight each or more
            per_ctx = [len(range(d, nc, nctx)) for d in range(nctx)]
            assert min(per_ctx) >= 1 and max(per_ctx) - min(per_ctx) <= 1
    # the test hook shrinks chunks so that a small input exercises slot reuse
    api.host_set_chunk_bytes(1 << 20, 1 << 20)
    try:
        cb, nc = api.host_chunk_plan(65536, 4096, 2, False)
        assert cb * 4096 == 1 << 20 and nc == 256
    finally:
        api.host_set_chunk_bytes(0, 0)
    cb0, nc0 = api.host_chunk_plan(65536, 4096, 2, False)
    assert cb0 * 4096 >= 16 << 20
    L = _lib.lib()
    assert L.redux_host_chunk_plan(0, 4096, 1, 0, C.byref(C.c_uint64()), C.byref(C.c_uint64())) == _lib.INVALID_INPUT
    assert L.redux_host_chunk_plan(10, 4096, 17, 0, C.byref(C.c_uint64()), C.byref(C.c_uint64())) == _lib.INVALID_INPUT


def test_decode_wrappers_reject_offsets_outside_the_streams():
    """The host calls copy streams[offsets[b] .. offsets[b + 1]) out of caller memory: the mirrors refuse an offsets table that
    runs past the bytes they were given, or backwards, before anything is read (no GPU involved)."""
    streams = np.zeros(100, dtype=np.uint8)
    with pytest.raises(api.InvalidInput):
        api.decompress_blocks(streams, [0, 50, 101], 4096)
    with pytest.raises(api.InvalidInput):
        api.decompress_blocks(streams, [0, 60, 50], 4096)
    with pytest.raises(api.InvalidInput):
        api.decompress_blocks_v(streams, [0, 50, 101], [4096, 10], 4096)
    with pytest.raises(api.InvalidInput):
        api.decompress_blocks_v(streams, [1, 50, 100], [4096, 10], 4096)
    with pytest.raises(api.InvalidInput):
        api.decompress_blocks_v(streams, [0, 70, 60], [4096, 10], 4096)


def test_decode_workspace_does_not_grow_with_the_capacity():
    """redux::decompress writes to an unbounded sink with O(1) state (lib.rs:113-120): a decode capacity of 1 GiB asks for a
    bounded workspace (a reciprocal table of 2^20 entries + slack, not one entry per byte of capacity)."""
    L = _lib.lib()
    p = _lib.Params(8, 30, 32)
    assert L.redux_decode_workspace_bytes(C.byref(p), 1, 1 << 30) <= (64 << 20)
    assert L.redux_decode_workspace_bytes(C.byref(p), 1, 0xFFFFFF00) <= (64 << 20)
    assert L.redux_decode_workspace_bytes(C.byref(p), 65536, 65536) <= (4 << 20)
