"""Block container: makes the multi-block output of the GPU path a self-describing file.

The reference has no container (one stream per call, src/lib.rs:102); SURVEY.md 8(f).1 asks
for one so that blocked output can be decoded again.  Every payload stays byte-identical to
`redux::compress(block)`.

Layout (little-endian):
    0   4  magic  b"RDXB"
    4   1  version (1)
    5   3  symbol_bits, freq_bits, code_bits      (Parameters::new arguments, src/model/mod.rs:63)
    8   4  block_size
   12   4  reserved (0)
   16   8  nblocks
   24   8  total uncompressed length
   32  4*nblocks   compressed size of each block
   ..  payloads, concatenated in block order
"""
import struct

import numpy as np

from . import api

MAGIC = b"RDXB"
VERSION = 1
HEADER = struct.Struct("<4sBBBBIIQQ")
# A header field, not a promise: a crafted 40-byte file must not make the decoder allocate
# gigabytes.  The container never holds blocks above 1 GiB (the CLI refuses larger ones), and
# the decode capacity is additionally capped by the total length the header declares.
MAX_BLOCK_SIZE = 1 << 30


def pack(streams, offsets, params, block_size, total_len):
    """streams: dense uint8 array; offsets: uint64[nblocks+1]."""
    P = api._params_of(params)
    offs = np.asarray(offsets, dtype=np.uint64)
    sizes = np.diff(offs.astype(np.int64))
    if (sizes < 0).any() or (sizes > 0xFFFFFFFF).any():
        raise api.InvalidInput()
    head = HEADER.pack(MAGIC, VERSION, P.symbol_bits, P.freq_bits, P.code_bits, block_size, 0, len(sizes), total_len)
    return head + sizes.astype("<u4").tobytes() + np.asarray(streams, dtype=np.uint8)[: int(offs[-1])].tobytes()


def unpack(buf):
    """-> (Parameters, block_size, total_len, offsets uint64[nblocks+1], payload uint8 array).
    Malformed containers raise InvalidInput, truncated ones Eof (src/lib.rs:57-64)."""
    b = memoryview(buf)
    if len(b) < HEADER.size:
        raise api.Eof()
    magic, ver, sb, fb, cb, block_size, _res, nblocks, total = HEADER.unpack_from(b, 0)
    if magic != MAGIC or ver != VERSION or block_size == 0 or block_size > MAX_BLOCK_SIZE:
        raise api.InvalidInput()
    P = api.Parameters(sb, fb, cb)
    if nblocks != (1 if total == 0 else (total + block_size - 1) // block_size):
        raise api.InvalidInput()
    end_sizes = HEADER.size + 4 * nblocks
    if len(b) < end_sizes:
        raise api.Eof()
    sizes = np.frombuffer(b, dtype="<u4", count=nblocks, offset=HEADER.size).astype(np.uint64)
    offsets = np.zeros(nblocks + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(sizes)
    if len(b) < end_sizes + int(offsets[-1]):
        raise api.Eof()
    payload = np.frombuffer(b, dtype=np.uint8, count=int(offsets[-1]), offset=end_sizes)
    return P, block_size, total, offsets, payload


def header_is_wellformed(buf):
    """True when the first 32 bytes are a consistent container header: magic, version, a triple
    Parameters::new accepts, a block size in range and a block count that matches the declared
    length.  The CLI uses it to tell a container from a raw reference stream that happens to begin
    with the same four bytes: a well-formed header followed by a damaged or truncated body is a
    damaged CONTAINER (unpack's error is reported), not a raw stream."""
    b = memoryview(buf)
    if len(b) < HEADER.size:
        return False
    magic, ver, sb, fb, cb, block_size, res, nblocks, total = HEADER.unpack_from(b, 0)
    if magic != MAGIC or ver != VERSION or res != 0 or not 0 < block_size <= MAX_BLOCK_SIZE:
        return False
    try:
        api.Parameters.new(sb, fb, cb)
    except api.Error:
        return False
    return nblocks == (1 if total == 0 else (total + block_size - 1) // block_size)


def compress_bytes(data, block_size=65536, params=(8, 30, 32)):
    """bytes -> container bytes (every block coded on the GPU)."""
    if not 0 < block_size <= MAX_BLOCK_SIZE:
        raise api.InvalidInput()
    out, offs, _ = api.compress_blocks(data, block_size, params)
    return pack(out, offs, params, block_size, len(data))


def decompress_bytes(buf):
    """container bytes -> original bytes."""
    P, block_size, total, offsets, payload = unpack(buf)
    # A stream of s bytes can decode to far more than s bytes (64 KiB of one symbol is ~400 bytes),
    # so only the declared total bounds the capacity; but every block's stream has at least one
    # byte, so a header that declares more blocks than there are payload bytes is malformed.
    nb = len(offsets) - 1
    if len(payload) < nb:
        raise api.InvalidInput()
    cap = max(1, min(block_size, total))  # one short block never needs block_size bytes of capacity
    try:
        out, sizes, _ = api.decompress_blocks(payload, offsets, cap, P)
    except MemoryError:  # a header can declare far more output than this machine holds: malformed for our purposes
        raise api.InvalidInput()
    expect = [min(block_size, total - b * block_size) for b in range(nb)] if total else [0]
    if [int(x) for x in sizes] != expect:
        raise api.InvalidInput()
    if total == nb * cap:
        return out.tobytes()
    return b"".join(out[b * cap: b * cap + int(sizes[b])].tobytes() for b in range(nb))
