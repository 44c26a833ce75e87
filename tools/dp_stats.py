#!/usr/bin/env python3
"""How often the core wave of k_decode_pair waits for its aux wave (a -DREDUX_DP_STATS build as redux_amd/libredux_hip.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redux_amd as rx  # noqa: E402
from redux_amd import _lib  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
BLOCK = 65536
n = nb * BLOCK
d_in = rx.gen_iid(n)
enc = rx.DeviceEncoder((8, 30, 32), BLOCK, n)
out, offs, st, sm = enc.encode(d_in)
total = int(offs[-1].item())
dec = rx.DeviceDecoder((8, 30, 32), BLOCK, nb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dec.decode(out[:total], offs)
torch.cuda.synchronize()
e0.record()
d_out = dec.decode(out[:total], offs)[0]
e1.record()
torch.cuda.synchronize()
assert torch.equal(d_out, d_in)
s = (C.c_uint64 * 8)()
L = C.CDLL(_lib.LIB_PATH)
L.redux_debug_dp_stats(s)
steps = s[4] or 1
print(f"decode {e0.elapsed_time(e1):.2f} ms; workgroup 7 over {steps} steps (2 launches): ack waits {s[0]} ({s[0] / steps:.3f}/step), "
      f"polls per wait {s[1] / max(s[0], 1):.1f}; window waits {s[2]} ({s[2] / steps:.3f}/step), polls per wait {s[3] / max(s[2], 1):.1f}")
