#!/usr/bin/env python3
"""Regenerates redux_amd/csrc/zipf_table.inc: the 256-entry u32 inverse-CDF table of the
Zipf(alpha=1.2) synthetic workload (BASELINE.json config 5)."""
import os
from decimal import Decimal, getcontext

getcontext().prec = 80
alpha = Decimal("1.2")
w = [Decimal(r) ** (-alpha) for r in range(1, 257)]
tot = sum(w)
acc = Decimal(0)
th = []
for r in range(256):
    acc += w[r]
    th.append(min(int((acc / tot) * (1 << 32)), (1 << 32) - 1))
th[-1] = (1 << 32) - 1
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "redux_amd", "csrc", "zipf_table.inc")
with open(out, "w") as f:
    f.write("// thresholds[r-1] = floor(2^32 * CDF(r)) for P(r) ~ r^-1.2, r = 1..256 (last = 2^32-1).\n")
    f.write("// Generated with 80-digit decimals by tools/make_zipf_table.py; committed so host and device agree.\n")
    for i in range(0, 256, 8):
        f.write("    " + ", ".join("0x%08Xu" % t for t in th[i:i + 8]) + ",\n")
