"""The 9 known-answer tests of /root/reference/src/bitio/tests.rs, asserted on the C oracle
and on the pure-Python restatement (exact bytes, get_count semantics, sticky Eof)."""
import ctypes as C

import numpy as np
import pytest

from oracle import cbind as ox
from oracle import redux_ref as rr


class CWriter:
    def __init__(self, cap=64):
        self.buf = np.zeros(cap, dtype=np.uint8)
        self.h = ox.lib().ox_bitwriter_new(self.buf.ctypes.data, cap)

    def write_bits(self, s, b):
        return ox.lib().ox_write_bits(self.h, s, b)

    def flush_bits(self):
        return ox.lib().ox_flush_bits(self.h)

    def get_count(self):
        return ox.lib().ox_bitwriter_count(self.h)

    def data(self):
        return self.buf[: self.get_count()].tobytes()


class CReader:
    def __init__(self, data):
        self.buf = np.frombuffer(bytes(data), dtype=np.uint8).copy() if data else np.zeros(1, dtype=np.uint8)
        self.h = ox.lib().ox_bitreader_new(self.buf.ctypes.data, len(data))

    def read_bits(self, b):
        r = C.c_size_t()
        st = ox.lib().ox_read_bits(self.h, b, C.byref(r))
        return ("Eof" if st == ox.EOF else "Err%d" % st) if st else r.value

    def get_count(self):
        return ox.lib().ox_bitreader_count(self.h)


class PyWriter:
    def __init__(self, cap=64):
        self.w = rr.BitWriter()

    def write_bits(self, s, b):
        self.w.write_bits(s, b)
        return 0

    def flush_bits(self):
        self.w.flush_bits()
        return 0

    def get_count(self):
        return self.w.get_count()

    def data(self):
        return bytes(self.w.out)


class PyReader:
    def __init__(self, data):
        self.r = rr.BitReader(data)

    def read_bits(self, b):
        try:
            return self.r.read_bits(b)
        except rr.Eof:
            return "Eof"

    def get_count(self):
        return self.r.get_count()


WRITERS = [CWriter, PyWriter]
READERS = [CReader, PyReader]


@pytest.mark.parametrize("W", WRITERS)
def test_write_empty(W):  # tests.rs:8-18
    w = W()
    assert w.get_count() == 0
    assert w.flush_bits() == 0
    assert w.get_count() == 0
    assert w.data() == b""


@pytest.mark.parametrize("W", WRITERS)
def test_write_bytes(W):  # tests.rs:20-34
    w = W()
    for i, v in enumerate([1, 2, 3]):
        assert w.get_count() == i
        assert w.write_bits(v, 8) == 0
    assert w.get_count() == 3
    assert w.data() == bytes([1, 2, 3])


@pytest.mark.parametrize("W", WRITERS)
def test_write_bits(W):  # tests.rs:36-66
    w = W()
    bits = [1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1]
    counts_after = {0: 0, 6: 0, 7: 1, 8: 1, 14: 1, 15: 2}
    for i, b in enumerate(bits):
        assert w.write_bits(b, 1) == 0
        if i in counts_after:
            assert w.get_count() == counts_after[i]
    assert w.data() == bytes([0b10101010, 0b1111])


@pytest.mark.parametrize("W", WRITERS)
def test_write_mixed(W):  # tests.rs:68-102
    w = W()
    for b in [1, 0, 1, 0, 1, 0, 1]:
        assert w.write_bits(b, 1) == 0
    assert w.get_count() == 0
    assert w.write_bits(0, 1) == 0
    assert w.get_count() == 1
    assert w.write_bits(0x00, 8) == 0
    assert w.get_count() == 2
    for b in [0, 0, 0, 0, 1, 1, 1]:
        assert w.write_bits(b, 1) == 0
        assert w.get_count() == 2
    assert w.write_bits(1, 1) == 0
    assert w.get_count() == 3
    assert w.write_bits(0xF0, 8) == 0
    assert w.get_count() == 4
    assert w.data() == bytes([0xAA, 0x00, 0x0F, 0xF0])


@pytest.mark.parametrize("W", WRITERS)
def test_write_flush(W):  # tests.rs:104-128
    w = W()
    assert w.flush_bits() == 0 and w.get_count() == 0
    for b in [1, 0, 1, 0]:
        assert w.write_bits(b, 1) == 0
        assert w.get_count() == 0
    assert w.flush_bits() == 0 and w.get_count() == 1
    assert w.write_bits(0, 1) == 0 and w.get_count() == 1
    assert w.flush_bits() == 0 and w.get_count() == 2
    assert w.flush_bits() == 0 and w.get_count() == 2
    assert w.data() == bytes([0xA0, 0x00])


@pytest.mark.parametrize("R", READERS)
def test_read_eof(R):  # tests.rs:130-141
    r = R(b"")
    assert r.get_count() == 0
    for n in (1, 8, 1, 8):
        assert r.read_bits(n) == "Eof"
    assert r.get_count() == 0


@pytest.mark.parametrize("R", READERS)
def test_read_bytes(R):  # tests.rs:143-156
    r = R(bytes([1, 2, 3]))
    for i, v in enumerate([1, 2, 3]):
        assert r.get_count() == i
        assert r.read_bits(8) == v
    assert r.get_count() == 3
    assert r.read_bits(8) == "Eof"
    assert r.get_count() == 3


@pytest.mark.parametrize("R", READERS)
def test_read_bits(R):  # tests.rs:158-184
    r = R(bytes([0b10101010, 0b1111]))
    assert r.get_count() == 0
    got = []
    for i in range(16):
        got.append(r.read_bits(1))
        assert r.get_count() == (1 if i < 8 else 2)
    assert got == [1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1]
    assert r.read_bits(8) == "Eof"
    assert r.get_count() == 2


@pytest.mark.parametrize("R", READERS)
def test_read_mixed(R):  # tests.rs:186-218
    r = R(bytes([0xAA, 0x00, 0x0F, 0xF0]))
    assert [r.read_bits(1) for _ in range(8)] == [1, 0, 1, 0, 1, 0, 1, 0]
    assert r.get_count() == 1
    assert r.read_bits(8) == 0 and r.get_count() == 2
    assert r.read_bits(1) == 0 and r.get_count() == 3
    assert [r.read_bits(1) for _ in range(7)] == [0, 0, 0, 1, 1, 1, 1]
    assert r.get_count() == 3
    assert r.read_bits(8) == 0xF0 and r.get_count() == 4
    assert r.read_bits(8) == "Eof" and r.get_count() == 4


def test_invalid_widths():  # bitio/mod.rs:79-81, :149-151
    w = CWriter()
    assert w.write_bits(2, 1) == ox.INVALID_INPUT
    assert w.write_bits(0, 65) == ox.INVALID_INPUT
    r = CReader(b"\x00")
    assert r.read_bits(65) == "Err%d" % ox.INVALID_INPUT
    with pytest.raises(rr.InvalidInput):
        rr.BitWriter().write_bits(2, 1)
