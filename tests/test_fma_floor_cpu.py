"""scale_div<.., NONZERO> (redux_amd/csrc/redux_coder.hpp) takes floor((R1+1)*f/c), f >= 1, as the low dword of
fma(Y, f, 2^52 - 0.5) with Y = fma(R1, rc, rc) and rc = RN(1/c) + 4 ulp (k_fill_rc).  This replays both fused operations with exact rationals
(float(Fraction) is correctly rounded, ties to even) against the integer quotient of codec.rs:59-60, on the cases the
argument in the header singles out -- exact multiples of c, quotients just below an integer, the largest range, and f = 0 (which the form excludes) --
and on random ones.  No GPU and no oracle involved: it pins the arithmetic identity the kernels rely on."""
import random
import struct
from fractions import Fraction

MAGIC = Fraction(2**52) - Fraction(1, 2)


def rc_of(c):
    r = float(Fraction(1, c))  # correctly rounded 1/c
    return struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", r))[0] + 4))[0]


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def device_quotient(R1, f, c):
    rc = rc_of(c)
    Y = fma(float(R1), rc, rc)
    z = float(Fraction(Y) * f + MAGIC)
    assert (2.0**52 <= z < 2.0**53) or f == 0
    return struct.unpack("<q", struct.pack("<d", z))[0] & 0xFFFFFFFF


def check(R1, f, c):
    assert device_quotient(R1, f, c) == ((R1 + 1) * f // c) & 0xFFFFFFFF, (R1, f, c)


def test_exact_multiples_and_near_misses():
    rnd = random.Random(5)
    for c in [257, 258, 511, 512, 513, 1000, 4096, 65536, 65793, 131071]:  # !FIXUP: count < 2^17
        for _ in range(300):
            f = rnd.randrange(1, c)
            # x = (R1+1)*f an exact multiple of c, one below, one above
            n = rnd.randrange(1, 2**32 // c)
            for R1 in {n * c - 1, max(0, n * c - 2), n * c, 2**32 - 1, 2**30, 2**30 - 1}:
                if R1 < 2**32:
                    check(R1, f, c)
        for R1 in (0, 1, 2**16, 2**31, 2**32 - 1):
            for f in (1, c - 1, c // 2):
                if (R1 + 1) * f >= c:  # (the coder's interval is never narrower than the count: the quotient is >= 1)
                    check(R1, f, c)


def test_zero_is_excluded():
    """f = 0: 0 + 2^52 - 0.5 is representable one binade down and its low dword is all ones -- the low end of a range
    (cum(0) = 0) therefore keeps the multiply + convert form."""
    assert device_quotient(2**31, 0, 257) == 0xFFFFFFFF


def test_random_cases():
    rnd = random.Random(6)
    for _ in range(20000):
        c = rnd.randrange(257, 2**17)
        check(rnd.randrange(2**30, 2**32), rnd.randrange(1, c), c)


def test_estimate_is_q_or_q_plus_one_for_large_counts():
    """FIXUP == true (count up to 2^30 - 1): the estimate may be one too large, never too small, and the remainder test
    of scale_div corrects it."""
    rnd = random.Random(7)
    for _ in range(20000):
        c = rnd.randrange(2**17, 2**30)
        R1, f = rnd.randrange(2**30, 2**32), rnd.randrange(1, c)
        q = device_quotient(R1, f, c)
        true = (R1 + 1) * f // c
        assert q in (true & 0xFFFFFFFF, (true + 1) & 0xFFFFFFFF), (R1, f, c)
        r = (R1 * f + f - q * c) & 0xFFFFFFFF
        assert ((q - (1 if r >= c else 0)) & 0xFFFFFFFF) == true & 0xFFFFFFFF
