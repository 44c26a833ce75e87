"""The Rust binding (rust/src/hip.rs) against the C header it binds (include/redux_hip.h).

No Rust toolchain exists in this image, so the shim cannot be compiled here; what can be checked
mechanically is that its `extern "C"` block declares exactly what the header declares: name,
number of arguments, and for every argument and the return type the kind (pointer / integer) and
the integer width -- the properties an FFI call gets silently wrong."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_RS = open(os.path.join(ROOT, "rust", "src", "hip.rs")).read()
HEADER = open(os.path.join(ROOT, "include", "redux_hip.h")).read()

RUST_TYPES = {"u8": ("int", 8), "u32": ("int", 32), "i32": ("int", 32), "u64": ("int", 64), "c_int": ("int", 32)}
C_TYPES = {"uint8_t": ("int", 8), "uint32_t": ("int", 32), "int32_t": ("int", 32), "uint64_t": ("int", 64),
           "int": ("int", 32)}


def rust_kind(t):
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(\w+)$", t)
    if m:
        return ("ptr", "const" if m.group(1) == "const" else "mut", m.group(2))
    return RUST_TYPES[t]


def c_kind(t):
    t = " ".join(t.replace("*", " * ").split())
    if "*" in t:
        base = t.replace("*", "").replace("const", "").strip()
        return ("ptr", "const" if "const" in t else "mut", base)
    return C_TYPES[t.replace("const", "").strip()]


def rust_externs():
    block = re.search(r'extern "C" \{(.*?)\n\}', HIP_RS, re.S).group(1)
    out = {}
    for m in re.finditer(r"fn (\w+)\((.*?)\)\s*(?:->\s*([\w\s\*]+))?;", block, re.S):
        args = [a.split(":", 1)[1] for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = ([rust_kind(a) for a in args], rust_kind(m.group(3)) if m.group(3) else None)
    return out


def header_decls():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*([\w\s\*]+?)\s*\b(redux_\w+)\s*\(([^;{]*?)\)\s*;", text, re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        alist = [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")]
        kinds = []
        for a in alist:
            a = re.sub(r"\b\w+$", "", a.strip()) if not a.strip().endswith("*") else a   # drop the parameter name
            kinds.append(c_kind(a))
        out[name] = (kinds, c_kind(ret))
    return out


STRUCT_MAP = {"ReduxParams": "redux_params"}


def same(rk, ck):
    if rk[0] != ck[0]:
        return False
    if rk[0] == "int":
        return rk[1] == ck[1]
    # pointers: constness and pointee must agree (void* never appears in the bound subset)
    if rk[1] != ck[1]:
        return False
    rbase, cbase = rk[2], ck[2]
    if rbase in STRUCT_MAP:
        return STRUCT_MAP[rbase] == cbase
    return RUST_TYPES[rbase] == C_TYPES[cbase]


def test_every_extern_matches_the_header():
    rust, hdr = rust_externs(), header_decls()
    assert len(rust) >= 12 and {"redux_encode_blocks", "redux_decode_blocks", "redux_compress", "redux_decompress",
                                 "redux_encode_blocks_v", "redux_decode_blocks_v"} <= set(rust)
    for name, (rargs, rret) in rust.items():
        assert name in hdr, f"{name} is bound in hip.rs but not declared in include/redux_hip.h"
        cargs, cret = hdr[name]
        assert len(rargs) == len(cargs), f"{name}: {len(rargs)} arguments in hip.rs, {len(cargs)} in the header"
        for i, (rk, ck) in enumerate(zip(rargs, cargs)):
            assert same(rk, ck), f"{name} argument {i}: hip.rs {rk} vs header {ck}"
        assert same(rret, cret), f"{name} return: hip.rs {rret} vs header {cret}"


def test_the_parser_sees_the_whole_header():
    # every symbol the ctypes loader binds (tests/test_abi_cpu.py checks those against the library)
    # must also be found by this file's header parser: otherwise a match above could be vacuous
    from redux_amd import _lib
    hdr = header_decls()
    assert set(_lib.SIGNATURES) <= set(hdr), sorted(set(_lib.SIGNATURES) - set(hdr))
    args, ret = hdr["redux_encode_blocks"]
    assert [a[0] for a in args] == ["ptr", "ptr", "int", "int", "ptr", "int", "ptr", "ptr"] and ret == ("int", 32)
    assert hdr["redux_block_count"] == ([("int", 64), ("int", 32)], ("int", 64))


def test_repr_c_struct_matches_redux_params():
    fields = re.search(r"#\[repr\(C\)\]\s*pub struct ReduxParams \{(.*?)\}", HIP_RS, re.S).group(1)
    rust = [(n, t) for n, t in re.findall(r"(\w+):\s*(\w+),", fields)]
    c = re.search(r"typedef struct redux_params \{(.*?)\} redux_params;", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S), re.S).group(1)
    cf = [(n, t) for t, n in re.findall(r"(\w+)\s+(\w+);", c)]
    assert rust == [(n, {"uint32_t": "u32"}[t]) for n, t in cf] and len(rust) == 3


def test_status_mapping_covers_the_header_enum():
    enum = dict((n, int(v)) for n, v in re.findall(r"(REDUX_\w+)\s*=\s*(\d+)", HEADER) if not n.startswith("REDUX_V_"))
    assert enum == {"REDUX_OK": 0, "REDUX_EOF": 1, "REDUX_INVALID_INPUT": 2, "REDUX_IO_ERROR": 3,
                    "REDUX_OUTPUT_TOO_SMALL": 4, "REDUX_UNSUPPORTED": 5}
    body = re.search(r"fn status\(st: c_int\).*?\n\}", HIP_RS, re.S).group(0)
    arms = dict(re.findall(r"(\d+|_) => (\w+)", body))
    assert arms["0"] == "Ok" and "Error::Eof" in body and "Error::InvalidInput" in body
    assert re.search(r"1 => Err\(Error::Eof\)", body) and re.search(r"2 => Err\(Error::InvalidInput\)", body)
    assert re.search(r"4 => Err\(Error::IoError", body) and re.search(r"5 => Err\(Error::IoError", body) and "_ => Err(Error::IoError" in body


def test_bound_symbols_are_exported_by_the_library():
    from redux_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    for name in rust_externs():
        getattr(L, name)
