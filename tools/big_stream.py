import io, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import redux_amd as rx
from oracle import cbind as ox
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
rng = np.random.default_rng(5)
text = open("tests/golden/corpora/large/bible.txt", "rb").read()
data = (text * (n // 2 // len(text) + 1))[: n // 2] + bytes((rng.integers(0, 256, n - n // 2, dtype=np.uint8) >> 3))
P = (8, 30, 32)
o = io.BytesIO()
t0 = time.time()
rx.compress(io.BytesIO(data), o, rx.AdaptiveTreeModel.new(rx.Parameters.new(*P)))
t1 = time.time()
comp = o.getvalue()
print("compress", n, "->", len(comp), "bytes", round(n / (t1 - t0) / 1e6, 2), "MB/s", flush=True)
import ctypes as C
from redux_amd import _lib
cp = _lib.Params(*P)
print("workspace MiB", _lib.lib().redux_encode_workspace_bytes(C.byref(cp), n, n) >> 20, "resident MiB", _lib.lib().redux_host_resident_bytes() >> 20, flush=True)
t0 = time.time()
want, _ = ox.compress(data, P)
print("oracle", round(n / (time.time() - t0) / 1e6, 2), "MB/s", "equal", want == comp, flush=True)
assert want == comp
