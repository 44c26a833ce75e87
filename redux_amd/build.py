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


def build_lib(force=False, verbose=False):
    """Cross-compiles for gfx950 (works without a GPU).  Returns the .so path."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
