// redux_any.hpp -- the coder for EVERY parameter triple Parameters::new accepts (with
// symbol_bits <= 16), gfx950 device code.  SURVEY.md section 8(f).3.
//
// The fast kernels of redux_hip.hip cover symbol_bits == 8, code_bits <= 32 (the reference CLI's
// fixed (8, 30, 32) and its two other tested widths).  Everything else -- 4- or 12-bit symbols,
// code_bits up to 62 -- runs here: one LANE per block, u64 interval state, true 64-bit division
// (code_bits + freq_bits <= 64, so range * freq fits u64: model/mod.rs:64), the Fenwick tree
// of adaptive_tree.rs as u32 in global memory (2^symbol_bits + 2 entries per block), bit I/O
// through a one-byte buffer exactly like bitio/mod.rs.  It favours being obviously the same
// algorithm over speed: this path exists for completeness of the drop-in, the roofline work
// is in the fast kernels.  Citations are file:line under the reference checkout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {
namespace any {

enum { OK = 0, ERR_EOF = 1, ERR_INVALID = 2, ERR_IO = 3, ERR_SMALL = 4 };

struct Params { // model/mod.rs:63-81
    uint32_t symbol_bits, freq_bits, code_bits;
    uint32_t symbol_eof, symbol_count;
    uint64_t freq_max, one_fourth, half, three_fourths, code_max;
};

__host__ __device__ inline Params make_params(uint32_t sb, uint32_t fb, uint32_t cb)
{
    Params p;
    p.symbol_bits   = sb;
    p.freq_bits     = fb;
    p.code_bits     = cb;
    p.symbol_eof    = 1u << sb;
    p.symbol_count  = (1u << sb) + 1u;
    p.freq_max      = (1ull << fb) - 1;
    p.one_fourth    = 1ull << (cb - 2);
    p.half          = 2ull << (cb - 2);
    p.three_fourths = 3ull << (cb - 2);
    p.code_max      = cb >= 64 ? ~0ull : (1ull << cb) - 1;
    return p;
}

// ---- bitio/mod.rs:78-120: MSB-first reader over memory, one byte buffered ----
struct BitReader {
    const uint8_t *in;
    uint64_t       len, pos;
    uint32_t       byte, bits;
    uint64_t       count; // bytes fetched (ByteCount, :71-75)
};

__device__ inline int read_bits(BitReader &r, uint32_t bits, uint64_t &out)
{
    uint64_t result = 0;
    while (bits > 0) {
        if (r.bits >= bits) { // :85-93
            result = (result << bits) | (r.byte >> (r.bits - bits));
            r.bits -= bits;
            r.byte &= (1u << r.bits) - 1u;
            bits = 0;
        } else if (r.bits > 0) { // :94-102
            result = (result << r.bits) | r.byte;
            bits -= r.bits;
            r.byte = 0;
            r.bits = 0;
        } else { // :104-111: fetch one byte; none left => Eof (the bits gathered so far are lost)
            if (r.pos >= r.len)
                return ERR_EOF;
            r.byte = r.in[r.pos++];
            r.count += 1;
            r.bits = 8;
        }
    }
    out = result;
    return OK;
}

// ---- bitio/mod.rs:148-198: MSB-first writer into a bounded buffer ----
struct BitWriter {
    uint8_t *out;
    uint64_t cap, pos;
    uint32_t byte, bits;
    uint64_t count; // bytes written
};

__device__ inline int flush_bits(BitWriter &w) // :183-198
{
    if (w.bits > 0) {
        if (w.pos >= w.cap)
            return ERR_SMALL; // the reference's write_all would fail with IoError here
        w.out[w.pos++] = (uint8_t)(w.byte << (8 - w.bits));
        w.count += 1;
        w.byte = 0;
        w.bits = 0;
    }
    return OK;
}

__device__ inline int write_bits(BitWriter &w, uint64_t symbol, uint32_t bits) // :148-181
{
    while (bits > 0) {
        if (w.bits + bits <= 8) { // :154-163
            w.byte = ((w.byte << bits) | (uint32_t)symbol) & 0xFFu;
            w.bits += bits;
            bits = 0;
        } else { // :164-174 (w.bits < 8 always holds here: a full byte is flushed at once)
            const uint32_t num = 8 - w.bits;
            w.byte             = ((w.byte << num) | (uint32_t)(symbol >> (bits - num))) & 0xFFu;
            w.bits += num;
            bits -= num;
            symbol &= (1ull << bits) - 1;
        }
        if (w.bits == 8) { // :176-178
            const int e = flush_bits(w);
            if (e)
                return e;
        }
    }
    return OK;
}

// ---- model/adaptive_tree.rs ----
struct Tree {
    uint32_t *v;     // symbol_count + 1 entries (:38); values < 2^31 since freq_bits <= 31
    uint64_t  count; // cached total (:15)
};

__device__ inline uint32_t last_one(uint32_t x) { return x & (0u - x); } // :27-31

__device__ inline void tree_new(Tree &t, uint32_t *mem, const Params &p) // :36-48
{
    t.v    = mem;
    t.v[0] = 0;
    for (uint32_t i = 1; i <= p.symbol_count; i++)
        t.v[i] = last_one(i);
    t.count = p.symbol_count;
}

__device__ inline uint64_t tree_single(const Tree &t, uint32_t symbol) // :51-59
{
    uint64_t sum = t.v[0];
    for (uint32_t i = symbol; i > 0; i -= last_one(i))
        sum += t.v[i];
    return sum;
}

__device__ inline void tree_range(const Tree &t, uint32_t symbol, uint64_t &lo, uint64_t &hi) // :63-80
{
    uint64_t sumh = 0, suml = 0;
    uint32_t h = symbol + 1, l = symbol;
    while (h != l) {
        if (h > l) {
            sumh += t.v[h];
            h -= last_one(h);
        } else {
            suml += t.v[l];
            l -= last_one(l);
        }
    }
    const uint64_t sumr = tree_single(t, h);
    lo = suml + sumr;
    hi = sumh + sumr;
}

__device__ inline void tree_update(Tree &t, uint32_t symbol, const Params &p) // :83-92
{
    if (t.count < p.freq_max) {
        for (uint32_t i = symbol; i <= p.symbol_count; i += last_one(i))
            t.v[i] += 1;
        t.count += 1;
    }
}

__device__ inline int tree_get_symbol(Tree &t, uint64_t value, const Params &p, uint32_t &sym, uint64_t &lo,
                                      uint64_t &hi) // :115-136
{
    uint32_t m = p.symbol_eof, i = 0;
    uint64_t v = value;
    while (m > 0 && i < p.symbol_eof) {
        const uint32_t ti = i + m;
        const uint64_t tv = t.v[ti];
        if (v >= tv) {
            i = ti;
            v -= tv;
        }
        m >>= 1;
    }
    tree_range(t, i, lo, hi);
    if (value >= hi)
        return ERR_INVALID; // :130
    tree_update(t, i + 1, p);
    sym = i;
    return OK;
}

// ---- codec.rs ----
struct Codec {
    uint64_t low, high, pending;
    uint32_t extra;
};

__device__ inline void codec_new(Codec &c, const Params &p) // :28-36
{
    c.low     = 0;
    c.high    = p.code_max;
    c.pending = 0;
    c.extra   = p.code_bits;
}

__device__ inline int put_bit(Codec &c, uint32_t bit, BitWriter &w) // :39-46
{
    int e = write_bits(w, bit, 1);
    if (e)
        return e;
    while (c.pending > 0) {
        e = write_bits(w, bit ^ 1u, 1);
        if (e)
            return e;
        c.pending -= 1;
    }
    return OK;
}

__device__ inline int compress_symbol(Codec &c, Tree &t, const Params &p, uint32_t symbol, BitWriter &w) // :55-101
{
    const uint64_t count = t.count; // read BEFORE the update (:56)
    uint64_t       lo, hi;
    tree_range(t, symbol, lo, hi); // get_frequency (:105-113): symbol <= eof by construction
    tree_update(t, symbol + 1, p);
    const uint64_t range = c.high - c.low + 1;       // :58
    c.high               = c.low + (range * hi / count) - 1; // :59
    c.low                = c.low + (range * lo / count);     // :60
    const bool is_eof    = symbol == p.symbol_eof;
    for (;;) { // :62-89
        int e = OK;
        if (c.high < p.half) {
            e = put_bit(c, 0, w);
        } else if (c.low >= p.half) {
            e = put_bit(c, 1, w);
        } else if (c.low >= p.one_fourth && c.high < p.three_fourths) {
            c.pending += 1;
            c.low -= p.one_fourth;
            c.high -= p.one_fourth;
        } else {
            break;
        }
        if (e)
            return e;
        if (is_eof)
            c.extra -= 1; // :66-82: every renormalisation step of the EOF symbol
        c.high = ((c.high << 1) + 1) & p.code_max; // :87
        c.low  = (c.low << 1) & p.code_max;        // :88
    }
    if (is_eof) { // :91-99
        while (c.extra > 0) {
            const int e = put_bit(c, (c.low & p.half) != 0, w);
            if (e)
                return e;
            c.low = (c.low << 1) & p.code_max;
            c.extra -= 1;
        }
        return flush_bits(w);
    }
    return OK;
}

__device__ inline int compress_stream(const Params &p, uint32_t *tree_mem, const uint8_t *in, uint64_t in_len, uint8_t *out,
                                      uint64_t out_cap, uint64_t &bytes_in, uint64_t &bytes_out) // :104-120 + lib.rs:102-109
{
    Tree t;
    tree_new(t, tree_mem, p);
    Codec c;
    codec_new(c, p);
    BitReader r = {in, in_len, 0, 0, 0, 0};
    BitWriter w = {out, out_cap, 0, 0, 0, 0};
    int       st = OK;
    for (;;) {
        uint64_t  sym;
        const int e = read_bits(r, p.symbol_bits, sym);
        if (e == ERR_EOF)
            sym = p.symbol_eof; // :108
        st = compress_symbol(c, t, p, (uint32_t)sym, w);
        if (st || sym == p.symbol_eof)
            break;
    }
    bytes_in  = r.count;
    bytes_out = w.count;
    return st;
}

__device__ inline int get_bit(Codec &c, BitReader &r) // :49-52
{
    uint64_t  b;
    const int e = read_bits(r, 1, b);
    if (e)
        return e;
    c.pending = (c.pending << 1) | b;
    return OK;
}

__device__ inline int decompress_symbol(Codec &c, Tree &t, const Params &p, BitReader &r, uint32_t &symbol) // :123-161
{
    while (c.extra > 0) { // :124-127
        const int e = get_bit(c, r);
        if (e)
            return e;
        c.extra -= 1;
    }
    const uint64_t range = c.high - c.low + 1;
    const uint64_t count = t.count;
    const uint64_t value = ((c.pending - c.low + 1) * count - 1) / range; // :131
    uint64_t       lo, hi;
    const int      e = tree_get_symbol(t, value, p, symbol, lo, hi);
    if (e)
        return e;
    c.high = c.low + (range * hi / count) - 1; // :133
    c.low  = c.low + (range * lo / count);     // :134
    if (symbol == p.symbol_eof)                // :136-138
        return OK;
    for (;;) { // :140-158
        if (c.high < p.half) {
        } else if (c.low >= p.half) {
            c.pending -= p.half;
            c.low -= p.half;
            c.high -= p.half;
        } else if (c.low >= p.one_fourth && c.high < p.three_fourths) {
            c.pending -= p.one_fourth;
            c.low -= p.one_fourth;
            c.high -= p.one_fourth;
        } else {
            break;
        }
        c.low  = c.low << 1;
        c.high = (c.high << 1) + 1;
        const int e2 = get_bit(c, r);
        if (e2)
            return e2;
    }
    return OK;
}

__device__ inline int decompress_stream(const Params &p, uint32_t *tree_mem, const uint8_t *in, uint64_t in_len, uint8_t *out,
                                        uint64_t out_cap, uint64_t &bytes_in, uint64_t &bytes_out) // :164-176 + lib.rs:113-120
{
    Tree t;
    tree_new(t, tree_mem, p);
    Codec c;
    codec_new(c, p);
    BitReader r = {in, in_len, 0, 0, 0, 0};
    BitWriter w = {out, out_cap, 0, 0, 0, 0};
    int       st = OK;
    for (;;) {
        uint32_t sym;
        st = decompress_symbol(c, t, p, r, sym);
        if (st || sym == p.symbol_eof)
            break;
        st = write_bits(w, sym, p.symbol_bits); // :171; no flush at the end (lib.rs:113-120): a partial byte is dropped
        if (st)
            break;
    }
    bytes_in  = r.count;
    bytes_out = w.count;
    return st;
}

} // namespace any
} // namespace redux
