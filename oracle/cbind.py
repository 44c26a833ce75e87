"""ctypes binding of oracle/libredux_oracle.so (the C restatement) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, EOF, INVALID_INPUT, IO_ERROR = 0, 1, 2, 3
LINEAR, TREE = 0, 1


def build(force=False):
    """Compile the C oracle with gcc (a no-op when the .so is newer than its sources)."""
    so = os.path.join(_HERE, "libredux_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("redux_oracle.c", "redux_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libredux_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class Params(C.Structure):
    _fields_ = [
        ("symbol_bits", C.c_size_t), ("symbol_eof", C.c_size_t), ("symbol_count", C.c_size_t),
        ("freq_bits", C.c_size_t), ("freq_max", C.c_uint64), ("code_bits", C.c_size_t),
        ("code_min", C.c_uint64), ("code_one_fourth", C.c_uint64), ("code_half", C.c_uint64),
        ("code_three_fourths", C.c_uint64), ("code_max", C.c_uint64),
    ]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p, i32p, u64p, szp = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_uint64, C.c_size_t))
        L.ox_params_new.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(Params)]
        L.ox_bitreader_new.restype = C.c_void_p
        L.ox_bitreader_new.argtypes = [C.c_void_p, C.c_size_t]
        L.ox_bitreader_free.argtypes = [C.c_void_p]
        L.ox_bitreader_count.restype = C.c_uint64
        L.ox_bitreader_count.argtypes = [C.c_void_p]
        L.ox_read_bits.argtypes = [C.c_void_p, C.c_size_t, szp]
        L.ox_bitwriter_new.restype = C.c_void_p
        L.ox_bitwriter_new.argtypes = [C.c_void_p, C.c_size_t]
        L.ox_bitwriter_free.argtypes = [C.c_void_p]
        L.ox_bitwriter_count.restype = C.c_uint64
        L.ox_bitwriter_count.argtypes = [C.c_void_p]
        L.ox_write_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.ox_flush_bits.argtypes = [C.c_void_p]
        L.ox_model_new.restype = C.c_void_p
        L.ox_model_new.argtypes = [C.c_int, C.POINTER(Params)]
        L.ox_model_free.argtypes = [C.c_void_p]
        L.ox_model_total_frequency.restype = C.c_uint64
        L.ox_model_total_frequency.argtypes = [C.c_void_p]
        L.ox_model_get_frequency.argtypes = [C.c_void_p, C.c_size_t, u64p, u64p]
        L.ox_model_get_symbol.argtypes = [C.c_void_p, C.c_uint64, szp, u64p, u64p]
        L.ox_model_get_freq_table.argtypes = [C.c_void_p, u64p, u64p]
        for f in (L.ox_compress, L.ox_decompress):
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                          C.c_int, u64p, u64p]
        L.ox_compress_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.ox_decompress_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                           C.c_int]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, status):
        super().__init__({EOF: "Eof", INVALID_INPUT: "InvalidInput", IO_ERROR: "IoError"}.get(status, str(status)))
        self.status = status


def params_new(sym, freq, code):
    p = Params()
    st = lib().ox_params_new(sym, freq, code, C.byref(p))
    if st:
        raise OracleError(st)
    return p


def _as_u8(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def compress(data, params=(8, 30, 32), model=TREE, cap=None):
    """redux::compress (src/lib.rs:102): returns (stream bytes, (bytes_in, bytes_out))."""
    a = _as_u8(data)
    cap = cap if cap is not None else len(a) * 2 + 1024
    out = np.empty(cap, dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    st = lib().ox_compress(a.ctypes.data, len(a), out.ctypes.data, cap, params[0], params[1], params[2], model,
                           C.byref(bi), C.byref(bo))
    if st:
        raise OracleError(st)
    return out[: bo.value].tobytes(), (bi.value, bo.value)


def decompress(data, params=(8, 30, 32), model=TREE, cap=None):
    """redux::decompress (src/lib.rs:113)."""
    a = _as_u8(data)
    cap = cap if cap is not None else max(len(a) * 64, 1 << 16)
    out = np.empty(cap, dtype=np.uint8)
    bi, bo = C.c_uint64(), C.c_uint64()
    st = lib().ox_decompress(a.ctypes.data, len(a), out.ctypes.data, cap, params[0], params[1], params[2], model,
                             C.byref(bi), C.byref(bo))
    if st:
        raise OracleError(st)
    return out[: bo.value].tobytes(), (bi.value, bo.value)


def _static_call(fn, data, cum, params, cap):
    a = _as_u8(data)
    out = np.empty(cap, dtype=np.uint8)
    tab = np.ascontiguousarray(np.asarray(cum, dtype=np.uint64))
    bi, bo = C.c_uint64(), C.c_uint64()
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    st = fn(a.ctypes.data, len(a), out.ctypes.data, cap, params[0], params[1], params[2], tab.ctypes.data,
            C.byref(bi), C.byref(bo))
    if st:
        raise OracleError(st)
    return out[: bo.value].tobytes(), (bi.value, bo.value)


def compress_static(data, cum, params=(8, 30, 32), cap=None):
    """compress under the static-table model (ox_compress_static; not in the reference)."""
    n = len(_as_u8(data))
    return _static_call(lib().ox_compress_static, data, cum, params, cap if cap is not None else n * 5 + 1024)


def decompress_static(data, cum, params=(8, 30, 32), cap=None):
    n = len(_as_u8(data))
    return _static_call(lib().ox_decompress_static, data, cum, params, cap if cap is not None else max(n * 64, 1 << 16))


def block_count(n, block_size):
    return 1 if n == 0 else (n + block_size - 1) // block_size


def compress_blocks(data, block_size, params=(8, 30, 32), model=TREE, nthreads=1, slot=None):
    """Per-block independent compress; returns (list of per-block streams, status array)."""
    a = _as_u8(data)
    nb = block_count(len(a), block_size)
    slot = slot if slot is not None else block_size + block_size // 4 + 1024
    out = np.empty(nb * slot, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    lib().ox_compress_blocks(a.ctypes.data, len(a), block_size, out.ctypes.data, slot, sizes.ctypes.data,
                             status.ctypes.data, params[0], params[1], params[2], model, nthreads)
    streams = [out[b * slot: b * slot + int(sizes[b])].tobytes() for b in range(nb)]
    return streams, status


def compress_blocks_raw(a, block_size, params=(8, 30, 32), model=TREE, nthreads=1):
    """Timing-friendly variant: numpy in, (slot buffer, sizes, status, slot) out, no per-block copies."""
    a = _as_u8(a)
    nb = block_count(len(a), block_size)
    slot = block_size + block_size // 4 + 1024
    out = np.empty(nb * slot, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    status = np.zeros(nb, dtype=np.int32)
    lib().ox_compress_blocks(a.ctypes.data, len(a), block_size, out.ctypes.data, slot, sizes.ctypes.data,
                             status.ctypes.data, params[0], params[1], params[2], model, nthreads)
    return out, sizes, status, slot
