"""In-tree build of libredux_hip.so (hipcc, gfx950 only).  `python -m redux_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libredux_hip.so")
SOURCES = ["redux_hip.hip"]
DEPS = sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "redux_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def extra_flags():
    """Additional compile flags of a variant build (REDUX_EXTRA_FLAGS, e.g. "-DREDUX_AB"): part of the source hash, so a
    profile of a variant is never taken for a profile of the product."""
    return os.environ.get("REDUX_EXTRA_FLAGS", "").split()


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources, the C header and the compile flags, in a fixed order: what
    redux_source_hash() returns for a library built from them.  Profiles record it, and bench.py only borrows a profiled
    figure (HBM traffic, instruction counts) when the library it loaded was built from the same sources the same way."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(os.path.join(CSRC, d), "rb").read())
    h.update(" ".join(FLAGS + extra_flags()).encode())
    return h.hexdigest()[:16]


def build_lib(force=False, verbose=False):
    """Cross-compiles for gfx950 (works without a GPU).  Returns the .so path."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + extra_flags() + ['-DREDUX_SOURCE_HASH="%s"' % source_hash(), "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
