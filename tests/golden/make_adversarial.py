#!/usr/bin/env python3
"""Builds the adversarial INPUTS under tests/golden/adversarial/ that force the coder's rare
data-dependent paths (expected outputs always come from the oracle at test time):

  pending_80_<params>.bin   symbols obtained by DECODING the bitstream 80 00 00 ... : every
  pending_7f_<params>.bin   symbol's interval contains the midpoint of the code range, so E3
                            steps (codec.rs:75-82) pile up pending bits without an E1/E2 to
                            flush them: runs of hundreds of pending bits, far beyond one
                            32-bit append.  Same for 7F FF FF ....
  width1_8_14_16.bin        greedy search (frozen model, freq_bits 14 / code_bits 16) for
                            symbols that make low == high after narrowing (all code_bits bits
                            shared: k = code_bits), the case the closed-form renormalisation
                            handles through its careful path.

Uses the pure-Python restatement (oracle/redux_ref.py).  Run: python tests/golden/make_adversarial.py
"""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import redux_ref as rr  # noqa: E402

OUT = os.path.join(HERE, "adversarial")
os.makedirs(OUT, exist_ok=True)


def decode_prefix(stream, params, n):
    model = rr.AdaptiveTreeModel(rr.Parameters(*params))
    codec = rr.Codec(model)
    inp = rr.BitReader(stream)
    out = bytearray()
    for _ in range(n):
        sym = codec.decompress_symbol(inp)
        if sym == model.parameters().symbol_eof:
            break
        out.append(sym)
    return bytes(out)


def stats(data, params):
    """(#symbols with low == high after narrowing, longest pending run) when encoding data."""
    model = rr.AdaptiveTreeModel(rr.Parameters(*params))
    c = rr.Codec(model)
    p = model.parameters()
    out = rr.BitWriter()
    hits = 0
    maxpend = 0
    for sym in data:
        count = model.total_frequency()
        snap = copy.deepcopy(model)
        lo, hi = snap.get_frequency(sym)
        rng = c.high - c.low + 1
        if c.low + rng * hi // count - 1 == c.low + rng * lo // count:
            hits += 1
        c.compress_symbol(sym, out)
        maxpend = max(maxpend, c.pending)
    return hits, maxpend


def width1_search(params, nsym):
    """Saturate one symbol until the model freezes, then greedily pick rare symbols that make
    the interval collapse to a single value as often as possible."""
    P = rr.Parameters(*params)
    model = rr.AdaptiveTreeModel(P)
    c = rr.Codec(model)
    out = rr.BitWriter()
    data = bytearray()
    while model.total_frequency() < P.freq_max:
        c.compress_symbol(0, out)
        data.append(0)
    import random
    rnd = random.Random(20261003)
    hits = 0
    for _ in range(nsym):
        count = model.total_frequency()
        rng = c.high - c.low + 1
        # the saturated symbol keeps low at 0 and the range above half; rare symbols move low
        # around, and now and then leave an interval that straddles the midpoint with a range
        # below 2*count -- there a frequency-1 symbol can get a single code value
        pick = 0 if rnd.random() < 0.7 else rnd.randrange(1, 256)
        if rng < 2 * count:
            for s in range(1, 256):
                lo, hi = model._range(s)
                if c.low + rng * hi // count - 1 == c.low + rng * lo // count:
                    pick = s
                    hits += 1
                    break
        c.compress_symbol(pick, out)
        data.append(pick)
    return bytes(data), hits


def main():
    for params in [(8, 30, 32), (8, 14, 16), (8, 22, 24)]:
        tag = "%d_%d_%d" % params
        for name, stream in (("80", b"\x80" + b"\x00" * 6000), ("7f", b"\x7f" + b"\xff" * 6000)):
            d = decode_prefix(stream, params, 6000)
            print("pending_%s_%s: %d symbols, (low==high, max pending) = %s" % (name, tag, len(d), stats(d, params)))
            open(os.path.join(OUT, "pending_%s_%s.bin" % (name, tag)), "wb").write(d)
    d, hits = width1_search((8, 14, 16), 6000)
    print("width1_8_14_16: %d symbols, low==high hits %d, stats %s" % (len(d), hits, stats(d, (8, 14, 16))))
    open(os.path.join(OUT, "width1_8_14_16.bin"), "wb").write(d)


if __name__ == "__main__":
    main()
