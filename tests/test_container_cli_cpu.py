"""Container format and CLI argument handling (host logic only, no GPU call)."""
import numpy as np
import pytest

from redux_amd import api, cli, container


def test_pack_unpack_roundtrip():
    streams = np.arange(10, dtype=np.uint8)
    offs = np.array([0, 3, 3, 10], dtype=np.uint64)
    blob = container.pack(streams, offs, (8, 30, 32), 65536, 3 * 65536 - 5)
    P, bs, total, o2, payload = container.unpack(blob)
    assert P.triple() == (8, 30, 32) and bs == 65536 and total == 3 * 65536 - 5
    assert o2.tolist() == offs.tolist() and payload.tobytes() == streams.tobytes()
    assert blob[:4] == b"RDXB" and len(blob) == 32 + 12 + 10


def test_unpack_rejects_malformed():
    blob = container.pack(np.zeros(4, np.uint8), np.array([0, 4], np.uint64), (8, 30, 32), 65536, 10)
    with pytest.raises(api.Eof):
        container.unpack(blob[:20])
    with pytest.raises(api.Eof):
        container.unpack(blob[:-1])
    with pytest.raises(api.InvalidInput):
        container.unpack(b"XXXX" + blob[4:])
    bad = bytearray(blob)
    bad[6] = 3  # freq_bits 3 < symbol_bits + 2: Parameters::new rejects it
    with pytest.raises(api.InvalidInput):
        container.unpack(bytes(bad))
    bad = bytearray(blob)
    bad[16] = 2  # nblocks inconsistent with total length
    with pytest.raises(api.InvalidInput):
        container.unpack(bytes(bad))


def test_cli_usage_and_open_errors(tmp_path, capsys):  # src/main.rs:84-106
    assert cli.main([]) == 1
    assert "Usage: redux (-c | -d)" in capsys.readouterr().err
    assert cli.main(["-i", "x"]) == 1          # neither -c nor -d
    assert cli.main(["-c", "-i"]) == 1         # missing value
    assert cli.main(["-c", "--bogus"]) == 1
    assert cli.main(["-c", "-i", str(tmp_path / "missing")]) == 2
    assert "Error while opening input file" in capsys.readouterr().err
    src = tmp_path / "in.bin"
    src.write_bytes(b"abc")
    assert cli.main(["-c", "-i", str(src), "-o", str(tmp_path / "no_dir" / "out")]) == 2
    assert cli.parse(["-d", "-o", "o", "-i", "i", "--block-size", "65536"]) == \
        {"compress": False, "input": "i", "output": "o", "block_size": 65536}


def test_unpack_rejects_oversized_block_size():
    # a crafted 40-byte file must not be able to make the decoder allocate gigabytes
    blob = container.HEADER.pack(container.MAGIC, container.VERSION, 8, 30, 32, 0xFFFFFFFF, 0, 1, 1) + b"\x01\0\0\0" + b"\xAA"
    with pytest.raises(api.InvalidInput):
        container.unpack(blob)
    with pytest.raises(api.InvalidInput):
        container.compress_bytes(b"abc", block_size=(1 << 30) + 1)


def test_cli_block_size_range_is_a_usage_error():
    assert cli.parse(["-c", "--block-size", str(1 << 33)]) is None   # used to be truncated by ctypes
    assert cli.parse(["-c", "--block-size", "-1"]) is None
    assert cli.parse(["-c", "--block-size", str(1 << 30)])["block_size"] == 1 << 30
    assert cli.main(["-c", "--block-size", str(1 << 33)]) == 1


def test_header_wellformedness_decides_container_vs_raw_stream(tmp_path, capsys):
    blob = container.pack(np.zeros(4, np.uint8), np.array([0, 4], np.uint64), (8, 30, 32), 65536, 10)
    assert container.header_is_wellformed(blob)
    assert container.header_is_wellformed(blob[:-3])          # truncated BODY: still a container (unpack reports Eof)
    assert not container.header_is_wellformed(blob[:20])      # truncated header: cannot be told from a raw stream
    assert not container.header_is_wellformed(b"XXXX" + blob[4:])
    bad = bytearray(blob)
    bad[6] = 3                                               # Parameters::new rejects the triple
    assert not container.header_is_wellformed(bytes(bad))
    bad = bytearray(blob)
    bad[16] = 2                                              # block count does not match the declared length
    assert not container.header_is_wellformed(bytes(bad))
    # the CLI reports a damaged container as a decompression error (exit 3) without touching the GPU:
    # unpack() fails before any device call
    src = tmp_path / "trunc.rdxb"
    src.write_bytes(blob[:-3])
    assert cli.main(["-d", "-i", str(src), "-o", str(tmp_path / "out")]) == 3
    assert "Decompression error" in capsys.readouterr().err
