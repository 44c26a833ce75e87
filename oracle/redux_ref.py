"""Pure-Python restatement of peterbudai/redux 0.3.0 (second, independent oracle).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/redux_oracle.c, straight from the
Rust text, so that a transcription slip in either one shows up as a disagreement between
them (tests/test_oracle_cross.py).  Pure-Python loops: small inputs only.

Parity status: the reference holds no compressed golden vector and cannot be run in this
image (no rustc/cargo); see oracle/redux_oracle.h.  Citations are file:line under
/root/reference.
"""

U64 = (1 << 64) - 1


class Eof(Exception):
    """src/lib.rs:59"""


class InvalidInput(Exception):
    """src/lib.rs:61"""


class IoError(Exception):
    """src/lib.rs:63"""


class Parameters:
    """src/model/mod.rs:32-81"""

    def __init__(self, symbol, frequency, code):
        if symbol < 1 or frequency < symbol + 2 or code < frequency + 2 or 64 < code + frequency:
            raise InvalidInput()  # :64-65
        self.symbol_bits = symbol
        self.symbol_eof = 1 << symbol
        self.symbol_count = (1 << symbol) + 1
        self.freq_bits = frequency
        self.freq_max = (1 << frequency) - 1
        self.code_bits = code
        self.code_min = 0
        self.code_one_fourth = 1 << (code - 2)
        self.code_half = 2 << (code - 2)
        self.code_three_fourths = 3 << (code - 2)
        self.code_max = (1 << code) - 1


class BitReader:
    """src/bitio/mod.rs:54-121"""

    def __init__(self, data):
        self.data = bytes(data)
        self.pos = 0
        self.byte = 0
        self.bits = 0
        self.count = 0

    def get_count(self):
        return self.count

    def read_bits(self, bits):
        if bits > 64:  # :79
            raise InvalidInput()
        result = 0
        while bits > 0:
            if self.bits >= bits:  # :85
                result = (result << bits) & U64
                result |= self.byte >> (self.bits - bits)
                self.bits -= bits
                self.byte &= (1 << self.bits) - 1
                bits = 0
            elif self.bits > 0:  # :94
                result = (result << self.bits) & U64
                result |= self.byte
                bits -= self.bits
                self.byte = 0
                self.bits = 0
            else:  # :103
                if self.pos >= len(self.data):
                    raise Eof()
                self.byte = self.data[self.pos]
                self.pos += 1
                self.count += 1
                self.bits = 8
        return result


class BitWriter:
    """src/bitio/mod.rs:124-199"""

    def __init__(self, cap=None):
        self.out = bytearray()
        self.cap = cap
        self.byte = 0
        self.bits = 0
        self.count = 0

    def get_count(self):
        return self.count

    def write_bits(self, symbol, bits):
        if bits > 64 or (symbol >> bits) > 0:  # :149
            raise InvalidInput()
        while bits > 0:
            if self.bits + bits <= 8:  # :154
                if self.bits > 0:
                    self.byte = (self.byte << bits) & 0xFF
                self.byte |= symbol & 0xFF
                self.bits += bits
                bits = 0
                symbol = 0
            elif self.bits < 8:  # :164
                num = 8 - self.bits
                if self.bits > 0:
                    self.byte = (self.byte << num) & 0xFF
                self.byte |= (symbol >> (bits - num)) & 0xFF
                self.bits += num
                bits -= num
                symbol &= (1 << bits) - 1
            if self.bits == 8:  # :176
                self.flush_bits()

    def flush_bits(self):
        if self.bits > 0:  # :184
            self.byte = (self.byte << (8 - self.bits)) & 0xFF
            if self.cap is not None and len(self.out) >= self.cap:
                raise IoError()
            self.out.append(self.byte)
            self.count += 1
            self.byte = 0
            self.bits = 0


def _last_one(x):
    return x & -x  # adaptive_tree.rs:27-31


class AdaptiveLinearModel:
    """src/model/adaptive_linear.rs"""

    def __init__(self, p):
        self.params = p
        self.freq = list(range(p.symbol_count + 1))  # :23-28 (freq[0]=0, freq[i]=i)

    def parameters(self):
        return self.params

    def total_frequency(self):
        return self.freq[self.params.symbol_count]  # :47-49

    def _update(self, symbol):  # :33-39
        if self.total_frequency() < self.params.freq_max:
            for i in range(symbol + 1, len(self.freq)):
                self.freq[i] += 1

    def get_frequency(self, symbol):  # :51-59
        if symbol > self.params.symbol_eof:
            raise InvalidInput()
        res = (self.freq[symbol], self.freq[symbol + 1])
        self._update(symbol)
        return res

    def get_symbol(self, value):  # :61-70
        for i in range(len(self.freq) - 1):
            if value < self.freq[i + 1]:
                res = (i, self.freq[i], self.freq[i + 1])
                self._update(i)
                return res
        raise InvalidInput()


class StaticModel:
    """Not in the reference: the Model interface (src/model/mod.rs) over a fixed cumulative table
    cum[0..=symbol_count].  No update."""

    def __init__(self, p, cum):
        cum = [int(x) for x in cum]
        if len(cum) != p.symbol_count + 1 or cum[0] != 0 or cum[-1] > p.freq_max:
            raise InvalidInput()
        if any(b <= a for a, b in zip(cum, cum[1:])):
            raise InvalidInput()
        self.params = p
        self.cum = cum

    def parameters(self):
        return self.params

    def total_frequency(self):
        return self.cum[-1]

    def get_frequency(self, symbol):
        if symbol > self.params.symbol_eof:
            raise InvalidInput()
        return (self.cum[symbol], self.cum[symbol + 1])

    def get_symbol(self, value):
        for i in range(len(self.cum) - 1):
            if value < self.cum[i + 1]:
                return (i, self.cum[i], self.cum[i + 1])
        raise InvalidInput()


class AdaptiveTreeModel:
    """src/model/adaptive_tree.rs"""

    def __init__(self, p):
        self.params = p
        self.tree = [_last_one(i) for i in range(p.symbol_count + 1)]  # :38-45
        self.count = p.symbol_count  # :39

    def parameters(self):
        return self.params

    def total_frequency(self):
        return self.count  # :100-103

    def _single(self, symbol):  # :51-59
        i = symbol
        s = self.tree[0]
        while i > 0:
            s += self.tree[i]
            i -= _last_one(i)
        return s

    def _range(self, symbol):  # :63-80
        sumh = 0
        suml = 0
        h = symbol + 1
        l = symbol
        while h != l:
            if h > l:
                sumh += self.tree[h]
                h -= _last_one(h)
            else:
                suml += self.tree[l]
                l -= _last_one(l)
        sumr = self._single(h)
        return (suml + sumr, sumh + sumr)

    def _update(self, symbol):  # :83-92
        if self.total_frequency() < self.params.freq_max:
            i = symbol
            while i <= self.params.symbol_count:
                self.tree[i] += 1
                i += _last_one(i)
            self.count += 1

    def get_frequency(self, symbol):  # :105-113
        if symbol > self.params.symbol_eof:
            raise InvalidInput()
        res = self._range(symbol)
        self._update(symbol + 1)
        return res

    def get_symbol(self, value):  # :115-136
        m = self.params.symbol_eof
        i = 0
        v = value
        while m > 0 and i < self.params.symbol_eof:
            ti = i + m
            tv = self.tree[ti]
            if v >= tv:
                i = ti
                v -= tv
            m >>= 1
        l, h = self._range(i)
        if value >= h:
            raise InvalidInput()
        self._update(i + 1)
        return (i, l, h)


class Codec:
    """src/codec.rs"""

    def __init__(self, m):  # :28-36
        self.low = m.parameters().code_min
        self.high = m.parameters().code_max
        self.pending = 0
        self.extra = m.parameters().code_bits
        self.model = m

    def _put_bit(self, bit, output):  # :39-46
        output.write_bits(1 if bit else 0, 1)
        while self.pending > 0:
            output.write_bits(0 if bit else 1, 1)
            self.pending -= 1

    def _get_bit(self, inp):  # :49-52
        self.pending = ((self.pending << 1) & U64) | inp.read_bits(1)

    def compress_symbol(self, symbol, output):  # :55-101
        p = self.model.parameters()
        count = self.model.total_frequency()
        low, high = self.model.get_frequency(symbol)
        rng = self.high - self.low + 1
        self.high = self.low + (rng * high // count) - 1
        self.low = self.low + (rng * low // count)
        while True:
            if self.high < p.code_half:
                self._put_bit(False, output)
                if symbol == p.symbol_eof:
                    self.extra -= 1
            elif self.low >= p.code_half:
                self._put_bit(True, output)
                if symbol == p.symbol_eof:
                    self.extra -= 1
            elif self.low >= p.code_one_fourth and self.high < p.code_three_fourths:
                self.pending += 1
                self.low -= p.code_one_fourth
                self.high -= p.code_one_fourth
                if symbol == p.symbol_eof:
                    self.extra -= 1
            else:
                break
            self.high = ((self.high << 1) + 1) & p.code_max
            self.low = (self.low << 1) & p.code_max
        if symbol == p.symbol_eof:
            while self.extra > 0:
                mask = self.low & p.code_half
                self._put_bit(mask != 0, output)
                self.low = (self.low << 1) & p.code_max
                self.extra -= 1
            output.flush_bits()

    def compress_stream(self, inp, output):  # :104-120
        p = self.model.parameters()
        while True:
            try:
                symbol = inp.read_bits(p.symbol_bits)
            except Eof:
                symbol = p.symbol_eof
            self.compress_symbol(symbol, output)
            if symbol == p.symbol_eof:
                break

    def decompress_symbol(self, inp):  # :123-161
        p = self.model.parameters()
        while self.extra > 0:
            self._get_bit(inp)
            self.extra -= 1
        rng = self.high - self.low + 1
        count = self.model.total_frequency()
        value = ((((self.pending - self.low + 1) & U64) * count - 1) & U64) // rng
        symbol, low, high = self.model.get_symbol(value)
        self.high = self.low + (rng * high // count) - 1
        self.low = self.low + (rng * low // count)
        if symbol == p.symbol_eof:
            return symbol
        while True:
            if self.high < p.code_half:
                pass
            elif self.low >= p.code_half:
                self.pending -= p.code_half
                self.low -= p.code_half
                self.high -= p.code_half
            elif self.low >= p.code_one_fourth and self.high < p.code_three_fourths:
                self.pending -= p.code_one_fourth
                self.low -= p.code_one_fourth
                self.high -= p.code_one_fourth
            else:
                break
            self.low = self.low << 1
            self.high = (self.high << 1) + 1
            self._get_bit(inp)
        return symbol

    def decompress_stream(self, inp, output):  # :164-176
        p = self.model.parameters()
        while True:
            symbol = self.decompress_symbol(inp)
            if symbol == p.symbol_eof:
                break
            output.write_bits(symbol, p.symbol_bits)


def compress(data, model):
    """src/lib.rs:102-109 -> (stream, (bytes_in, bytes_out))"""
    codec = Codec(model)
    inp = BitReader(data)
    out = BitWriter()
    codec.compress_stream(inp, out)
    return bytes(out.out), (inp.get_count(), out.get_count())


def decompress(data, model):
    """src/lib.rs:113-120"""
    codec = Codec(model)
    inp = BitReader(data)
    out = BitWriter()
    codec.decompress_stream(inp, out)
    return bytes(out.out), (inp.get_count(), out.get_count())
