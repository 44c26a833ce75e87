"""src/model/tests.rs restated: linear == tree at the 14 parameter triples, encode and decode
direction, same iteration counts (long enough to reach the freq_max freeze at small widths),
error cases for out-of-range symbol / value; plus Parameters::new validation (mod.rs:64)."""
import pytest

from oracle import cbind as ox
from oracle import redux_ref as rr

TRIPLES = [
    (4, 10, 16, 10000), (4, 14, 16, 10000), (4, 22, 24, 100000), (4, 24, 30, 100000), (4, 30, 32, 200000),
    (8, 10, 16, 10000), (8, 14, 16, 10000), (8, 22, 24, 100000), (8, 24, 30, 100000), (8, 30, 32, 200000),
    (12, 14, 16, 10000), (12, 22, 24, 100000), (12, 24, 30, 100000), (12, 30, 32, 200000),
]


@pytest.mark.parametrize("bits,freq,code,iters", TRIPLES)
@pytest.mark.parametrize("decode", [0, 1])
def test_compare_models(bits, freq, code, iters, decode):  # tests.rs:95-251
    L = ox.lib()
    L.ox_selftest_models.restype = __import__("ctypes").c_int64
    every = 1 if bits <= 8 and iters <= 10000 else 997
    assert L.ox_selftest_models(bits, freq, code, iters, 0xC0FFEE + bits * 100 + freq, decode, every) == -1


def test_python_models_agree_with_c():
    import ctypes as C
    import random
    rnd = random.Random(5)
    for (sb, fb, cb) in [(4, 10, 16), (8, 10, 16), (8, 30, 32)]:
        p = ox.params_new(sb, fb, cb)
        cm = ox.lib().ox_model_new(ox.TREE, C.byref(p))
        pm = rr.AdaptiveTreeModel(rr.Parameters(sb, fb, cb))
        pl = rr.AdaptiveLinearModel(rr.Parameters(sb, fb, cb))
        lo, hi = C.c_uint64(), C.c_uint64()
        for _ in range(3000):
            s = rnd.randrange((1 << sb) + 1)
            assert ox.lib().ox_model_get_frequency(cm, s, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == pm.get_frequency(s) == pl.get_frequency(s)
            assert ox.lib().ox_model_total_frequency(cm) == pm.total_frequency() == pl.total_frequency()
        ox.lib().ox_model_free(cm)


@pytest.mark.parametrize("s,f,c,ok", [
    (8, 30, 32, True), (8, 14, 16, True), (8, 22, 24, True), (1, 3, 5, True), (8, 10, 12, True), (8, 31, 33, True),
    (0, 14, 16, False), (8, 9, 16, False), (8, 14, 15, False), (8, 32, 34, False), (8, 31, 34, False),
])
def test_parameters_validation(s, f, c, ok):  # model/mod.rs:64
    if ok:
        p = ox.params_new(s, f, c)
        q = rr.Parameters(s, f, c)
        for k in ("symbol_eof", "symbol_count", "freq_max", "code_one_fourth", "code_half", "code_three_fourths", "code_max"):
            assert getattr(p, k) == getattr(q, k)
        assert p.symbol_eof == 1 << s and p.freq_max == (1 << f) - 1 and p.code_max == (1 << c) - 1
    else:
        with pytest.raises(ox.OracleError):
            ox.params_new(s, f, c)
        with pytest.raises(rr.InvalidInput):
            rr.Parameters(s, f, c)
