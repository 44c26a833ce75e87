"""ctypes loader of the C ABI declared in include/redux_hip.h.

There is no fallback of any kind: if libredux_hip.so is missing or a symbol is absent this
module raises, and every public function of redux_amd fails with it.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, in-tree.  REDUX_LIB names another build to load instead: how the A/B tools (tools/run_variants.sh,
# tools/small_grid.py ...) time a variant without ever overwriting the product library (an interrupted run used to leave
# a foreign library in its place that needs_build() would not replace).
LIB_PATH = os.environ.get("REDUX_LIB") or os.path.join(HERE, "libredux_hip.so")

OK, EOF, INVALID_INPUT, IO_ERROR, OUTPUT_TOO_SMALL, UNSUPPORTED = 0, 1, 2, 3, 4, 5


class Params(C.Structure):
    _fields_ = [("symbol_bits", C.c_uint32), ("freq_bits", C.c_uint32), ("code_bits", C.c_uint32)]


_PP = C.POINTER(Params)
_V = C.c_void_p
_U64 = C.c_uint64
_U32 = C.c_uint32

# name -> (restype, argtypes): exactly the declarations of include/redux_hip.h
SIGNATURES = {
    "redux_version": (C.c_char_p, []),
    "redux_source_hash": (C.c_char_p, []),
    "redux_encode_kernel_name": (C.c_char_p, [_PP, _V, _U64, _U32]),
    "redux_decode_kernel_name": (C.c_char_p, [_PP, _V, _U32]),
    "redux_decode_kernel_name_n": (C.c_char_p, [_PP, _V, _U32, _U64]),
    "redux_debug_rcp_check": (C.c_int, [_U64, _U64, C.POINTER(C.c_double)]),
    "redux_debug_role_book": (C.c_int, [_PP, _U64, _U32, C.POINTER(_U64), C.POINTER(_U64)]),
    "redux_params_check": (C.c_int, [_U32, _U32, _U32]),
    "redux_device_supports": (C.c_int, [_PP]),
    "redux_block_count": (_U64, [_U64, _U32]),
    "redux_encode_slot_bytes": (_U64, [_PP, _U32]),
    "redux_encode_bound": (_U64, [_PP, _U64, _U32]),
    "redux_encode_workspace_bytes": (_U64, [_PP, _U64, _U32]),
    "redux_decode_workspace_bytes": (_U64, [_PP, _U64, _U32]),
    "redux_encode_blocks": (C.c_int, [_PP, _V, _U64, _U32, _V, _U64, _V, _V]),
    "redux_decode_blocks": (C.c_int, [_PP, _V, _V, _U64, _U32, _V, _U64, _V, _V]),
    "redux_block_count_v": (_U64, [_V, _U64, _U32]),
    "redux_block_table_v": (_U64, [_V, _V, _U64, _U32, _V]),
    "redux_encode_blocks_v": (C.c_int, [_PP, _V, _V, _V, _U64, _U32, _V, _U64, _V, _V]),
    "redux_decode_blocks_v": (C.c_int, [_PP, _V, _V, _V, _V, _V, _U64, _U32, _V, _V]),
    "redux_encode_blocks_v_dev": (C.c_int, [_PP, _V, _U64, _V, _U64, _U64, _U32, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_decode_blocks_v_dev": (C.c_int, [_PP, _V, _V, _V, _U64, _U64, _U32, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_compress": (C.c_int, [_PP, _V, _U64, _V, _U64, C.POINTER(_U64), C.POINTER(_U64)]),
    "redux_decompress": (C.c_int, [_PP, _V, _U64, _V, _U64, C.POINTER(_U64), C.POINTER(_U64)]),
    "redux_host_release": (C.c_int, []),
    "redux_host_set_devices": (C.c_int, [_V, _U32]),
    "redux_host_chunk_plan": (C.c_int, [_U64, _U32, _U32, C.c_int, C.POINTER(_U64), C.POINTER(_U64)]),
    "redux_host_set_chunk_bytes": (C.c_int, [_U64, _U64]),
    "redux_host_allocations": (_U64, []),
    "redux_host_resident_bytes": (_U64, []),
    "redux_host_trace": (_U64, [C.POINTER(C.c_double), _U64]),
    "redux_encode_blocks_dev": (C.c_int, [_PP, _V, _U64, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_decode_blocks_dev": (C.c_int, [_PP, _V, _V, _U64, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_encode_slots_dev": (C.c_int, [_PP, _V, _U64, _U32, _V, _V, _U64, _V]),
    "redux_compact_slots_dev": (C.c_int, [_PP, _U64, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_static_table_check": (C.c_int, [_PP, C.POINTER(_U32)]),
    "redux_static_encode_bound": (_U64, [_PP, _U64, _U32]),
    "redux_static_encode_workspace_bytes": (_U64, [_PP, _U64, _U32]),
    "redux_static_encode_blocks_dev": (C.c_int, [_PP, C.POINTER(_U32), _V, _U64, _U32, _V, _U64, _V, _V, _V, _V, _U64, _V]),
    "redux_static_decode_blocks_dev": (C.c_int, [_PP, C.POINTER(_U32), _V, _V, _U64, _U32, _V, _U64, _V, _V, _V, _V]),
    "redux_gen_iid_dev": (C.c_int, [_V, _U64, _U64, _U64, _V]),
    "redux_gen_zipf_dev": (C.c_int, [_V, _U64, _U64, _U64, _V]),
    "redux_zipf_thresholds": (C.POINTER(_U32), []),
}

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m redux_amd.build` "
                "(there is no CPU fallback for the redux hot path)")
        # One HIP runtime per process: torch ships its own libamdhip64.so (SONAME
        # libamdhip64.so.7).  Loaded first, it satisfies this library's NEEDED entry, so both
        # share a runtime (and device pointers / streams are interchangeable).  Loaded second,
        # torch would bring a second runtime that cannot open the device.  A consumer without
        # torch (plain C/C++) gets the system ROCm runtime through the library's RUNPATH.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB
