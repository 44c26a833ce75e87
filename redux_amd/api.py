"""Host-side mirror of the reference's public interface for the block-coding path.

Reference surface (file:line under the reference checkout) and what stands for it here:
  redux::Error / Result            src/lib.rs:57-98      -> Error, Eof, InvalidInput, IoError
  model::Parameters::new           src/model/mod.rs:63   -> Parameters(symbol, frequency, code)
  model::AdaptiveTreeModel::new    adaptive_tree.rs:36   -> AdaptiveTreeModel(params)
  redux::compress / decompress     src/lib.rs:102,113    -> compress / decompress (file-like in, file-like out)
  (new) block API                  SURVEY.md 8(b)        -> compress_blocks / decompress_blocks,
                                                            DeviceEncoder / DeviceDecoder (HBM-resident)

Every byte of every stream is produced by the gfx950 kernels behind include/redux_hip.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Params as _CParams


# ---- src/lib.rs:57-98 -------------------------------------------------------------------
class Error(Exception):
    """redux::Error"""
    status = None


class Eof(Error):
    """Error::Eof -- "Unexpected end of file" (src/lib.rs:69)"""
    status = _lib.EOF

    def __str__(self):
        return "Unexpected end of file"


class InvalidInput(Error):
    """Error::InvalidInput (src/lib.rs:70)"""
    status = _lib.INVALID_INPUT

    def __str__(self):
        return "Invalid data found while processing input"


class IoError(Error):
    """Error::IoError -- on this path: a HIP runtime failure"""
    status = _lib.IO_ERROR


class OutputTooSmall(IoError):
    status = _lib.OUTPUT_TOO_SMALL


class Unsupported(Error):
    """Valid Parameters that the device path does not implement (no CPU fallback exists)."""
    status = _lib.UNSUPPORTED


_BY_STATUS = {c.status: c for c in (Eof, InvalidInput, IoError, OutputTooSmall, Unsupported)}


def _raise(status, what=""):
    if status != _lib.OK:
        raise _BY_STATUS.get(status, Error)(what or f"status {status}")


# ---- src/model/mod.rs:32-81 -------------------------------------------------------------
class Parameters:
    """model::Parameters: same eleven fields, same validation rule (mod.rs:64)."""

    def __init__(self, symbol, frequency, code):
        if _lib.lib().redux_params_check(symbol, frequency, code) != _lib.OK:
            raise InvalidInput()
        self.symbol_bits = symbol
        self.symbol_eof = 1 << symbol
        self.symbol_count = (1 << symbol) + 1
        self.freq_bits = frequency
        self.freq_max = (1 << frequency) - 1
        self.code_bits = code
        self.code_min = 0
        self.code_one_fourth = 1 << (code - 2)
        self.code_half = 2 << (code - 2)
        self.code_three_fourths = 3 << (code - 2)
        self.code_max = (1 << code) - 1

    @classmethod
    def new(cls, symbol, frequency, code):
        return cls(symbol, frequency, code)

    def _c(self):
        return _CParams(self.symbol_bits, self.freq_bits, self.code_bits)

    def triple(self):
        return (self.symbol_bits, self.freq_bits, self.code_bits)


class AdaptiveTreeModel:
    """model::AdaptiveTreeModel::new(Parameters) -- the model the device implements.  The
    object only carries the parameters: the tree itself lives in LDS, one per block."""

    def __init__(self, params):
        self.params = params

    @classmethod
    def new(cls, params):
        return cls(params)

    def parameters(self):
        return self.params


def _params_of(model_or_params):
    if isinstance(model_or_params, AdaptiveTreeModel):
        return model_or_params.params
    if isinstance(model_or_params, Parameters):
        return model_or_params
    if isinstance(model_or_params, (tuple, list)):
        return Parameters(*model_or_params)
    raise TypeError("expected Parameters, AdaptiveTreeModel or a (symbol, frequency, code) triple")


MAX_BLOCK_BYTES = 0xFFFFFF00  # largest single block / whole stream the C ABI takes (redux_compress)


def version():
    return _lib.lib().redux_version().decode()


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data if a.size else None


# ---- block API, host buffers ------------------------------------------------------------
def compress_blocks(data, block_size, params=(8, 30, 32)):
    """Per-block redux::compress on the GPU.  Returns (dense streams as uint8 array,
    offsets uint64[nblocks+1], status int32[nblocks]); raises on the first non-OK block."""
    P = _params_of(params)
    a = _u8(data)
    L = _lib.lib()
    cp = P._c()
    _raise(L.redux_device_supports(C.byref(cp)))
    if block_size <= 0:
        raise InvalidInput()
    nb = L.redux_block_count(len(a), block_size)
    cap = L.redux_encode_bound(C.byref(cp), len(a), block_size)
    out = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(nb + 1, dtype=np.uint64)
    status = np.zeros(nb, dtype=np.int32)
    st = L.redux_encode_blocks(C.byref(cp), _ptr(a), len(a), block_size, out.ctypes.data, cap, offs.ctypes.data,
                               status.ctypes.data)
    _raise(st)
    return out[: int(offs[-1])], offs, status


def decompress_blocks(streams, offsets, block_size, params=(8, 30, 32), check=True):
    """Per-block redux::decompress on the GPU.  Returns (out uint8[nblocks*block_size],
    sizes uint32[nblocks], status int32[nblocks]); block b occupies out[b*block_size:][:sizes[b]]."""
    P = _params_of(params)
    a = _u8(streams)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    nb = len(offs) - 1
    if nb < 0 or (nb >= 0 and len(offs) and (int(offs[-1]) > len(a) or bool((offs[1:] < offs[:-1]).any()))):
        raise InvalidInput()  # (the C call reads streams[offsets[b] .. offsets[b + 1]) from caller memory)
    L = _lib.lib()
    cp = P._c()
    _raise(L.redux_device_supports(C.byref(cp)))
    out = np.empty(nb * block_size, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    st = L.redux_decode_blocks(C.byref(cp), _ptr(a), offs.ctypes.data, nb, block_size, out.ctypes.data, out.size,
                               sizes.ctypes.data, status.ctypes.data)
    if check:
        _raise(st)
    return out, sizes, status


# ---- several GPUs behind the host-pointer calls ------------------------------------------------
def host_set_devices(device_ids):
    """redux_host_set_devices: later compress_blocks / decompress_blocks calls deal their chunks round-robin over one
    context per entry of device_ids (an id may repeat); [] = back to HIP's current device."""
    ids = np.ascontiguousarray(device_ids, dtype=np.int32)
    _raise(_lib.lib().redux_host_set_devices(ids.ctypes.data if ids.size else None, int(ids.size)))


def host_chunk_plan(nblocks, block_size, ncontexts=1, decode=False):
    """(blocks per chunk, number of chunks) a host-pointer call uses; chunk k runs on context k % ncontexts."""
    cb, nc = C.c_uint64(), C.c_uint64()
    _raise(_lib.lib().redux_host_chunk_plan(nblocks, block_size, ncontexts, 1 if decode else 0, C.byref(cb), C.byref(nc)))
    return cb.value, nc.value


def host_set_chunk_bytes(min_bytes=0, max_bytes=0):
    """Test hook: chunk size limits of the host-pointer pipeline (0, 0 = the defaults)."""
    _raise(_lib.lib().redux_host_set_chunk_bytes(min_bytes, max_bytes))


# ---- many independent inputs in one call (tests/corpora.rs:32-85 codes file by file) --------
BLOCK_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u4"), ("index", "<u4")])  # redux_block
BLOCK_IDLE = 0xFFFFFFFF  # REDUX_BLOCK_IDLE


def block_table_v(offsets, lengths, block_size):
    """redux_block_table_v: the block table (launch order) of inputs at `offsets` with `lengths`."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    L = _lib.lib()
    ne = L.redux_block_table_v(off.ctypes.data, ln.ctypes.data, len(ln), block_size, None)  # entries: blocks + idle lanes
    tab = np.zeros(ne, dtype=BLOCK_DTYPE)
    L.redux_block_table_v(off.ctypes.data, ln.ctypes.data, len(ln), block_size, tab.ctypes.data)
    return tab


def compress_blocks_v(inputs, block_size, params=(8, 30, 32)):
    """redux::compress of every block of every input (a list of bytes-like objects), ONE launch for all of
    them; each input is cut into blocks on its own.  Returns (dense streams uint8, offsets
    uint64[nblocks+1], status int32[nblocks], first_block int64[len(inputs)+1]): input i owns blocks
    first_block[i] .. first_block[i+1]-1."""
    P = _params_of(params)
    L = _lib.lib()
    cp = P._c()
    _raise(L.redux_device_supports(C.byref(cp)))
    if block_size <= 0 or len(inputs) == 0:
        raise InvalidInput()
    arrs = [_u8(x) for x in inputs]
    lens = np.array([len(a) for a in arrs], dtype=np.uint64)
    offs_in = np.zeros(len(arrs), dtype=np.uint64)
    offs_in[1:] = np.cumsum(lens)[:-1]
    flat = np.concatenate(arrs) if int(lens.sum()) else np.zeros(0, dtype=np.uint8)
    counts = np.array([L.redux_block_count(int(n), block_size) for n in lens], dtype=np.int64)
    first = np.zeros(len(arrs) + 1, dtype=np.int64)
    first[1:] = np.cumsum(counts)
    nb = int(first[-1])
    cap = nb * L.redux_encode_slot_bytes(C.byref(cp), block_size)
    out = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(nb + 1, dtype=np.uint64)
    status = np.zeros(nb, dtype=np.int32)
    _raise(L.redux_encode_blocks_v(C.byref(cp), _ptr(flat), offs_in.ctypes.data, lens.ctypes.data, len(arrs), block_size,
                                   out.ctypes.data, cap, offs.ctypes.data, status.ctypes.data))
    return out[: int(offs[-1])], offs, status, first


def decompress_blocks_v(streams, offsets, lengths, block_size, params=(8, 30, 32), check=True):
    """The inverse of compress_blocks_v: `lengths[i]` is the decoded size of input i.  Returns (list of
    uint8 arrays, sizes uint32[nblocks], status int32[nblocks])."""
    P = _params_of(params)
    L = _lib.lib()
    cp = P._c()
    _raise(L.redux_device_supports(C.byref(cp)))
    a = _u8(streams)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lengths, dtype=np.uint64)
    if block_size <= 0 or len(lens) == 0:
        raise InvalidInput()
    nb = L.redux_block_count_v(lens.ctypes.data, len(lens), block_size)
    if nb != len(offs) - 1:
        raise InvalidInput()
    # the C call copies streams[offsets[0] .. offsets[nb]) from caller memory: they must lie inside `streams`, in order
    if int(offs[0]) != 0 or int(offs[-1]) > len(a) or bool((offs[1:] < offs[:-1]).any()):
        raise InvalidInput()
    out_off = np.zeros(len(lens), dtype=np.uint64)
    out_off[1:] = np.cumsum(lens)[:-1]
    out = np.zeros(max(1, int(lens.sum())), dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    st = L.redux_decode_blocks_v(C.byref(cp), _ptr(a), offs.ctypes.data, out.ctypes.data, out_off.ctypes.data, lens.ctypes.data,
                                 len(lens), block_size, sizes.ctypes.data, status.ctypes.data)
    if check:
        _raise(st)
    return [out[int(o): int(o) + int(n)] for o, n in zip(out_off, lens)], sizes, status


# ---- src/lib.rs:102-120: whole-stream drop-ins --------------------------------------------
def compress(istream, ostream, model):
    """redux::compress(istream, ostream, model) -> (bytes_in, bytes_out).  The whole input is
    one block, so the stream equals the reference's; it is coded by one GPU lane."""
    P = _params_of(model)
    a = _u8(istream.read())
    L = _lib.lib()
    cp = P._c()
    cap = L.redux_encode_bound(C.byref(cp), len(a), max(len(a), 1))
    _raise(L.redux_device_supports(C.byref(cp)))
    out = np.empty(cap, dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    _raise(L.redux_compress(C.byref(cp), _ptr(a), len(a), out.ctypes.data, cap, C.byref(bi), C.byref(bo)))
    ostream.write(out[: bo.value].tobytes())
    return (bi.value, bo.value)


def decompress(istream, ostream, model, max_output=None):
    """redux::decompress(istream, ostream, model) -> (bytes_in, bytes_out)."""
    P = _params_of(model)
    a = _u8(istream.read())
    L = _lib.lib()
    cp = P._c()
    _raise(L.redux_device_supports(C.byref(cp)))
    # The reference writes to an unbounded io::Write (src/lib.rs:113); the C ABI wants a capacity.
    # With no explicit max_output the capacity grows until the stream fits (or the ABI's one-block
    # limit of 0xFFFFFF00 bytes is reached): a highly compressible stream (2 MiB of zeros is
    # ~500 bytes) must not fail because of a guess.
    cap = max_output if max_output is not None else max(64 * len(a), 1 << 20)
    while True:
        cap = min(cap, MAX_BLOCK_BYTES)
        out = np.empty(cap, dtype=np.uint8)
        bi, bo = C.c_uint64(), C.c_uint64()
        st = L.redux_decompress(C.byref(cp), _ptr(a), len(a), out.ctypes.data, cap, C.byref(bi), C.byref(bo))
        if st == _lib.OUTPUT_TOO_SMALL and max_output is None and cap < MAX_BLOCK_BYTES:
            cap *= 8
            continue
        _raise(st)
        break
    ostream.write(out[: bo.value].tobytes())
    return (bi.value, bo.value)


# ---- block API, HBM-resident (torch only provides memory + streams) ------------------------
def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("redux_amd device API needs a GPU (torch.cuda.is_available() is False)")
    return torch


def _stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(method):
    """The C ABI launches on HIP's CURRENT device and on the stream it is handed.  A coder object
    built for cuda:1 while cuda:0 is current would run its kernels on GPU 0 against GPU 1's memory,
    so every launching method makes the object's device current first (torch's current stream of
    that device is then the one passed down)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with _torch().cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapper


class DeviceEncoder:
    """Reusable encoder for inputs of up to max_in_len bytes already resident in HBM.
    Allocates once (workspace, dense output, offsets, status); encode() only enqueues kernels
    on torch's current stream."""

    def __init__(self, params, block_size, max_in_len, device="cuda:0"):
        torch = _torch()
        self.P = _params_of(params)
        self.cp = self.P._c()
        L = _lib.lib()
        _raise(L.redux_device_supports(C.byref(self.cp)))
        self.block_size = int(block_size)
        self.max_in_len = int(max_in_len)
        self.nblocks_max = L.redux_block_count(self.max_in_len, self.block_size)
        self.ws_bytes = L.redux_encode_workspace_bytes(C.byref(self.cp), self.max_in_len, self.block_size)
        self.out_cap = L.redux_encode_bound(C.byref(self.cp), self.max_in_len, self.block_size)
        self.device = torch.device(device)
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        self.ws_off = (-self.ws.data_ptr()) % 256
        self.out = torch.empty(self.out_cap, dtype=torch.uint8, device=self.device)
        self.offsets = torch.zeros(self.nblocks_max + 1, dtype=torch.int64, device=self.device)
        self.status = torch.zeros(self.nblocks_max, dtype=torch.int32, device=self.device)
        self.summary = torch.zeros(2, dtype=torch.int32, device=self.device)

    def _ws_ptr(self):
        return C.c_void_p(self.ws.data_ptr() + self.ws_off)

    @_on_device
    def encode_slots(self, d_in):
        """Phase 1 only: the coder kernel (padded slots + sizes inside the workspace)."""
        torch = _torch()
        n = d_in.numel()
        assert d_in.dtype == torch.uint8 and d_in.is_contiguous() and n <= self.max_in_len
        st = _lib.lib().redux_encode_slots_dev(C.byref(self.cp), C.c_void_p(d_in.data_ptr()), n, self.block_size,
                                               C.c_void_p(self.status.data_ptr()), self._ws_ptr(), self.ws_bytes,
                                               _stream_ptr(torch))
        _raise(st)

    @_on_device
    def compact(self, n):
        """Phase 2 only: scan + gather into the dense output."""
        torch = _torch()
        self.summary.zero_()
        st = _lib.lib().redux_compact_slots_dev(C.byref(self.cp), n, self.block_size, C.c_void_p(self.out.data_ptr()),
                                                self.out_cap, C.c_void_p(self.offsets.data_ptr()),
                                                C.c_void_p(self.status.data_ptr()),
                                                C.c_void_p(self.summary.data_ptr()), self._ws_ptr(), self.ws_bytes,
                                                _stream_ptr(torch))
        _raise(st)

    @_on_device
    def encode(self, d_in):
        """Full pass, stream-ordered: returns (out, offsets[nblocks+1], status, summary) views
        of this encoder's buffers (valid until the next call)."""
        torch = _torch()
        n = d_in.numel()
        assert d_in.dtype == torch.uint8 and d_in.is_contiguous() and n <= self.max_in_len
        self.summary.zero_()
        st = _lib.lib().redux_encode_blocks_dev(C.byref(self.cp), C.c_void_p(d_in.data_ptr()), n, self.block_size,
                                                C.c_void_p(self.out.data_ptr()), self.out_cap,
                                                C.c_void_p(self.offsets.data_ptr()), C.c_void_p(self.status.data_ptr()),
                                                C.c_void_p(self.summary.data_ptr()), self._ws_ptr(), self.ws_bytes,
                                                _stream_ptr(torch))
        _raise(st)
        nb = _lib.lib().redux_block_count(n, self.block_size)
        return self.out, self.offsets[: nb + 1], self.status[:nb], self.summary


class DeviceDecoder:
    """Reusable decoder for up to max_blocks blocks resident in HBM."""

    def __init__(self, params, block_size, max_blocks, device="cuda:0"):
        torch = _torch()
        self.P = _params_of(params)
        self.cp = self.P._c()
        L = _lib.lib()
        _raise(L.redux_device_supports(C.byref(self.cp)))
        self.block_size = int(block_size)
        self.max_blocks = int(max_blocks)
        self.ws_bytes = L.redux_decode_workspace_bytes(C.byref(self.cp), self.max_blocks, self.block_size)
        self.device = torch.device(device)
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        self.ws_off = (-self.ws.data_ptr()) % 256
        self.out = torch.empty(self.max_blocks * self.block_size, dtype=torch.uint8, device=self.device)
        self.sizes = torch.zeros(self.max_blocks, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(self.max_blocks, dtype=torch.int32, device=self.device)
        self.summary = torch.zeros(2, dtype=torch.int32, device=self.device)

    @_on_device
    def decode(self, d_streams, d_offsets):
        torch = _torch()
        nb = d_offsets.numel() - 1
        assert nb <= self.max_blocks and d_offsets.dtype == torch.int64 and d_streams.dtype == torch.uint8
        self.summary.zero_()
        st = _lib.lib().redux_decode_blocks_dev(C.byref(self.cp), C.c_void_p(d_streams.data_ptr()),
                                                C.c_void_p(d_offsets.data_ptr()), nb, self.block_size,
                                                C.c_void_p(self.out.data_ptr()), self.out.numel(),
                                                C.c_void_p(self.sizes.data_ptr()), C.c_void_p(self.status.data_ptr()),
                                                C.c_void_p(self.summary.data_ptr()),
                                                C.c_void_p(self.ws.data_ptr() + self.ws_off), self.ws_bytes,
                                                _stream_ptr(torch))
        _raise(st)
        return self.out[: nb * self.block_size], self.sizes[:nb], self.status[:nb], self.summary


class DeviceStaticCoder:
    """The coder core under a fixed frequency table (include/redux_hip.h
    "static-table model").  cum: 258 cumulative frequencies, cum[0] = 0, strictly increasing,
    cum[257] = total <= freq_max; symbol 256 is EOF.  Same buffers as DeviceEncoder/DeviceDecoder."""

    def __init__(self, params, cum, block_size, max_in_len, device="cuda:0"):
        torch = _torch()
        self.P = _params_of(params)
        self.cp = self.P._c()
        L = _lib.lib()
        self.cum = (C.c_uint32 * 258)(*[int(x) for x in cum])
        _raise(L.redux_static_table_check(C.byref(self.cp), self.cum))
        self.block_size = int(block_size)
        self.max_in_len = int(max_in_len)
        self.nblocks_max = L.redux_block_count(self.max_in_len, self.block_size)
        self.ws_bytes = L.redux_static_encode_workspace_bytes(C.byref(self.cp), self.max_in_len, self.block_size)
        self.out_cap = L.redux_static_encode_bound(C.byref(self.cp), self.max_in_len, self.block_size)
        self.device = torch.device(device)
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        self.ws_off = (-self.ws.data_ptr()) % 256
        self.out = torch.empty(self.out_cap, dtype=torch.uint8, device=self.device)
        self.offsets = torch.zeros(self.nblocks_max + 1, dtype=torch.int64, device=self.device)
        self.status = torch.zeros(self.nblocks_max, dtype=torch.int32, device=self.device)
        self.summary = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.dec_out = None

    @_on_device
    def encode(self, d_in):
        torch = _torch()
        n = d_in.numel()
        assert d_in.dtype == torch.uint8 and d_in.is_contiguous() and n <= self.max_in_len
        self.summary.zero_()
        st = _lib.lib().redux_static_encode_blocks_dev(
            C.byref(self.cp), self.cum, C.c_void_p(d_in.data_ptr()), n, self.block_size, C.c_void_p(self.out.data_ptr()),
            self.out_cap, C.c_void_p(self.offsets.data_ptr()), C.c_void_p(self.status.data_ptr()),
            C.c_void_p(self.summary.data_ptr()), C.c_void_p(self.ws.data_ptr() + self.ws_off), self.ws_bytes,
            _stream_ptr(torch))
        _raise(st)
        nb = _lib.lib().redux_block_count(n, self.block_size)
        return self.out, self.offsets[: nb + 1], self.status[:nb], self.summary

    @_on_device
    def decode(self, d_streams, d_offsets):
        torch = _torch()
        nb = d_offsets.numel() - 1
        assert nb <= self.nblocks_max and d_offsets.dtype == torch.int64 and d_streams.dtype == torch.uint8
        if self.dec_out is None:
            self.dec_out = torch.empty(self.nblocks_max * self.block_size, dtype=torch.uint8, device=self.device)
            self.dec_sizes = torch.zeros(self.nblocks_max, dtype=torch.int32, device=self.device)
            self.dec_status = torch.zeros(self.nblocks_max, dtype=torch.int32, device=self.device)
            self.dec_summary = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.dec_summary.zero_()
        st = _lib.lib().redux_static_decode_blocks_dev(
            C.byref(self.cp), self.cum, C.c_void_p(d_streams.data_ptr()), C.c_void_p(d_offsets.data_ptr()), nb,
            self.block_size, C.c_void_p(self.dec_out.data_ptr()), self.dec_out.numel(),
            C.c_void_p(self.dec_sizes.data_ptr()), C.c_void_p(self.dec_status.data_ptr()),
            C.c_void_p(self.dec_summary.data_ptr()), _stream_ptr(torch))
        _raise(st)
        return self.dec_out[: nb * self.block_size], self.dec_sizes[:nb], self.dec_status[:nb], self.dec_summary


# ---- synthetic workloads (BASELINE.json configs 2 and 5) ------------------------------------
def gen_iid(nbytes, seed=0x5EED0001, first_byte=0, device="cuda:0", out=None):
    torch = _torch()
    t = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
    with torch.cuda.device(t.device):
        _raise(_lib.lib().redux_gen_iid_dev(C.c_void_p(t.data_ptr()), nbytes, first_byte, seed, _stream_ptr(torch)))
    return t


def gen_zipf(nbytes, seed=0x5EED0005, first_byte=0, device="cuda:0", out=None):
    torch = _torch()
    t = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
    with torch.cuda.device(t.device):
        _raise(_lib.lib().redux_gen_zipf_dev(C.c_void_p(t.data_ptr()), nbytes, first_byte, seed, _stream_ptr(torch)))
    return t


def zipf_thresholds():
    p = _lib.lib().redux_zipf_thresholds()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()
