/*
 * redux_oracle.c -- statement-by-statement CPU restatement of peterbudai/redux 0.3.0
 * (bit I/O, Parameters, AdaptiveLinearModel, AdaptiveTreeModel, Codec, compress /
 * decompress).  TEST INFRASTRUCTURE ONLY -- see redux_oracle.h for who may load this and
 * for what pins it ("parity pinned by source restatement + the reference's bit-I/O /
 * differential / round-trip assertions; no compressed golden vector exists upstream").
 *
 * Deliberately slow and literal: bit-at-a-time coder, u64 state, true `/`, one virtual
 * call per bit -- the same shape as the Rust so each line can be checked against it.
 * Citations are file:line under /root/reference.
 */
#include "redux_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * src/model/mod.rs
 * ==================================================================================== */

/* src/model/mod.rs:63-81 */
int ox_params_new(size_t symbol, size_t frequency, size_t code, ox_params *out)
{
    /* :64 */
    if (symbol < 1 || frequency < symbol + 2 || code < frequency + 2 || 64 < code + frequency)
        return OX_INVALID_INPUT;
    out->symbol_bits        = symbol;                              /* :68 */
    out->symbol_eof         = (size_t)1 << symbol;                 /* :69 */
    out->symbol_count       = ((size_t)1 << symbol) + 1;           /* :70 */
    out->freq_bits          = frequency;                           /* :71 */
    out->freq_max           = ((uint64_t)1 << frequency) - 1;      /* :72 */
    out->code_bits          = code;                                /* :73 */
    out->code_min           = 0;                                   /* :74 */
    out->code_one_fourth    = (uint64_t)1 << (code - 2);           /* :75 */
    out->code_half          = (uint64_t)2 << (code - 2);           /* :76 */
    out->code_three_fourths = (uint64_t)3 << (code - 2);           /* :77 */
    out->code_max           = ((uint64_t)1 << code) - 1;           /* :78 */
    return OX_OK;
}

/* ======================================================================================
 * src/bitio/mod.rs
 * ==================================================================================== */

/* :33-52 BitBuffer { bytes:[u8;1], bits, count } */
typedef struct {
    uint8_t  bytes0;
    size_t   bits;
    uint64_t count;
} ox_bitbuffer;

struct ox_bitreader {
    ox_bitbuffer   buffer;
    const uint8_t *input; /* the io::Read: a Cursor over memory */
    size_t         len, pos;
};

struct ox_bitwriter {
    ox_bitbuffer buffer;
    uint8_t     *output; /* the io::Write: a bounded memory sink; full => IoError */
    size_t       cap, pos;
};

ox_bitreader *ox_bitreader_new(const uint8_t *data, size_t len) /* :63 */
{
    ox_bitreader *r = (ox_bitreader *)calloc(1, sizeof *r);
    r->input = data;
    r->len   = len;
    return r;
}
void     ox_bitreader_free(ox_bitreader *r) { free(r); }
uint64_t ox_bitreader_count(const ox_bitreader *r) { return r->buffer.count; } /* :71-75 */

/* :78-120 */
int ox_read_bits(ox_bitreader *self, size_t bits, size_t *out)
{
    if (bits > sizeof(size_t) * 8) /* :79 */
        return OX_INVALID_INPUT;

    size_t result = 0;
    while (bits > 0) { /* :84 */
        if (self->buffer.bits >= bits) { /* :85 */
            /* bits <= 8 here, so the shifts are well defined */
            result <<= bits;                                                    /* :87 */
            result |= (size_t)self->buffer.bytes0 >> (self->buffer.bits - bits); /* :88 */
            self->buffer.bits -= bits;                                          /* :90 */
            self->buffer.bytes0 &= (uint8_t)((1u << self->buffer.bits) - 1);    /* :91 */
            bits = 0;                                                           /* :93 */
        } else if (self->buffer.bits > 0) { /* :94 */
            result <<= self->buffer.bits;           /* :96 */
            result |= (size_t)self->buffer.bytes0;  /* :97 */
            bits -= self->buffer.bits;              /* :99 */
            self->buffer.bytes0 = 0;                /* :101 */
            self->buffer.bits   = 0;                /* :102 */
        } else {
            /* :105 self.input.read(&mut bytes): Ok(0) => Eof, Ok(_) => one byte fetched */
            if (self->pos >= self->len)
                return OX_EOF; /* :106-108; count NOT bumped, Eof is sticky */
            self->buffer.bytes0 = self->input[self->pos++];
            self->buffer.count += 1; /* :110 */
            self->buffer.bits = 8;   /* :111 */
        }
    }
    *out = result;
    return OX_OK;
}

ox_bitwriter *ox_bitwriter_new(uint8_t *data, size_t cap) /* :133 */
{
    ox_bitwriter *w = (ox_bitwriter *)calloc(1, sizeof *w);
    w->output = data;
    w->cap    = cap;
    return w;
}
void     ox_bitwriter_free(ox_bitwriter *w) { free(w); }
uint64_t ox_bitwriter_count(const ox_bitwriter *w) { return w->buffer.count; } /* :141-145 */

/* :183-198 */
int ox_flush_bits(ox_bitwriter *self)
{
    if (self->buffer.bits > 0) {                                                   /* :184 */
        self->buffer.bytes0 = (uint8_t)(self->buffer.bytes0 << (8 - self->buffer.bits)); /* :185 */
        if (self->pos >= self->cap) /* write_all fails => IoError (:192-194) */
            return OX_IO_ERROR;
        self->output[self->pos++] = self->buffer.bytes0; /* :186 */
        self->buffer.count += 1;                         /* :188 */
        self->buffer.bytes0 = 0;                         /* :189 */
        self->buffer.bits   = 0;                         /* :190 */
    }
    return OX_OK;
}

/* :148-181 */
int ox_write_bits(ox_bitwriter *self, size_t symbol, size_t bits)
{
    /* :149  (bits > 64) || (symbol >> bits > 0).  Rust release builds mask the shift
     * amount, so bits == 64 behaves as `symbol >> 0`; restated as such. */
    if (bits > sizeof(size_t) * 8)
        return OX_INVALID_INPUT;
    if ((bits == 64 ? symbol : (symbol >> bits)) > 0)
        return OX_INVALID_INPUT;

    while (bits > 0) { /* :153 */
        if (self->buffer.bits + bits <= 8) { /* :154 */
            if (self->buffer.bits > 0)                                   /* :156 */
                self->buffer.bytes0 = (uint8_t)(self->buffer.bytes0 << bits);
            self->buffer.bytes0 |= (uint8_t)symbol;                      /* :159 */
            self->buffer.bits += bits;                                   /* :160 */
            bits   = 0;                                                  /* :162 */
            symbol = 0;                                                  /* :163 */
        } else if (self->buffer.bits < 8) { /* :164 */
            size_t num = 8 - self->buffer.bits;                          /* :165 */
            if (self->buffer.bits > 0)                                   /* :167 */
                self->buffer.bytes0 = (uint8_t)(self->buffer.bytes0 << num);
            self->buffer.bytes0 |= (uint8_t)(symbol >> (bits - num));    /* :170 */
            self->buffer.bits += num;                                    /* :171 */
            bits -= num;                                                 /* :173 */
            symbol &= (bits >= 64) ? ~(size_t)0 : (((size_t)1 << bits) - 1); /* :174 */
        }
        if (self->buffer.bits == 8) { /* :176 */
            int e = ox_flush_bits(self);
            if (e)
                return e;
        }
    }
    return OX_OK;
}

/* ======================================================================================
 * src/model/adaptive_linear.rs and src/model/adaptive_tree.rs behind one "trait object"
 * ==================================================================================== */

struct ox_model {
    int       kind;
    ox_params params;
    uint64_t *v;     /* linear: freq[symbol_count+1]; tree: tree[symbol_count+1] */
    size_t    vlen;
    uint64_t  count; /* tree only: cached total (adaptive_tree.rs:15) */
};

static size_t last_one(size_t x) { return x & (~x + 1); } /* adaptive_tree.rs:27-31 */

/* ---- linear (adaptive_linear.rs) ---- */
static uint64_t linear_total(const ox_model *m) { return m->v[m->params.symbol_count]; } /* :47-49 */

static void linear_update(ox_model *m, size_t symbol) /* :33-39 */
{
    if (linear_total(m) < m->params.freq_max)
        for (size_t i = symbol + 1; i < m->vlen; i++)
            m->v[i] += 1;
}

static int linear_get_frequency(ox_model *m, size_t symbol, uint64_t *lo, uint64_t *hi) /* :51-59 */
{
    if (symbol > m->params.symbol_eof)
        return OX_INVALID_INPUT;
    *lo = m->v[symbol];
    *hi = m->v[symbol + 1];
    linear_update(m, symbol);
    return OX_OK;
}

static int linear_get_symbol(ox_model *m, uint64_t value, size_t *sym, uint64_t *lo, uint64_t *hi) /* :61-70 */
{
    for (size_t i = 0; i < m->vlen - 1; i++) {
        if (value < m->v[i + 1]) {
            *sym = i;
            *lo  = m->v[i];
            *hi  = m->v[i + 1];
            linear_update(m, i);
            return OX_OK;
        }
    }
    return OX_INVALID_INPUT;
}

/* ---- tree (adaptive_tree.rs) ---- */
static uint64_t tree_get_frequency_single(const ox_model *m, size_t symbol) /* :51-59 */
{
    size_t   i   = symbol;
    uint64_t sum = m->v[0];
    while (i > 0) {
        sum += m->v[i];
        i -= last_one(i);
    }
    return sum;
}

static void tree_get_frequency_range(const ox_model *m, size_t symbol, uint64_t *lo, uint64_t *hi) /* :63-80 */
{
    uint64_t sumh = 0, suml = 0;
    size_t   h = symbol + 1, l = symbol;
    while (h != l) {
        if (h > l) {
            sumh += m->v[h];
            h -= last_one(h);
        } else {
            suml += m->v[l];
            l -= last_one(l);
        }
    }
    uint64_t sumr = tree_get_frequency_single(m, h);
    *lo = suml + sumr;
    *hi = sumh + sumr;
}

static void tree_update(ox_model *m, size_t symbol) /* :83-92 */
{
    if (m->count < m->params.freq_max) { /* total_frequency() == self.count (:100-103) */
        size_t i = symbol;
        while (i <= m->params.symbol_count) {
            m->v[i] += 1;
            i += last_one(i);
        }
        m->count += 1;
    }
}

static int tree_get_frequency(ox_model *m, size_t symbol, uint64_t *lo, uint64_t *hi) /* :105-113 */
{
    if (symbol > m->params.symbol_eof)
        return OX_INVALID_INPUT;
    tree_get_frequency_range(m, symbol, lo, hi);
    tree_update(m, symbol + 1);
    return OX_OK;
}

static int tree_get_symbol(ox_model *m, uint64_t value, size_t *sym, uint64_t *lo, uint64_t *hi) /* :115-136 */
{
    size_t   mm = m->params.symbol_eof;
    size_t   i  = 0;
    uint64_t v  = value;
    while (mm > 0 && i < m->params.symbol_eof) { /* :119 */
        size_t   ti = i + mm;
        uint64_t tv = m->v[ti];
        if (v >= tv) {
            i = ti;
            v -= tv;
        }
        mm >>= 1;
    }
    uint64_t l, h;
    tree_get_frequency_range(m, i, &l, &h); /* :129 */
    if (value >= h)                         /* :130 */
        return OX_INVALID_INPUT;
    tree_update(m, i + 1); /* :133 */
    *sym = i;
    *lo  = l;
    *hi  = h;
    return OX_OK;
}

/* ---- static-table model: not in the reference.  The Model trait (src/model/mod.rs) with a fixed
 * cumulative table v[0..=symbol_count]: get_frequency(s) = (v[s], v[s+1]), get_symbol(value) =
 * the s with v[s] <= value < v[s+1], total_frequency() = v[symbol_count], no update.  The codec
 * that drives it is the reference's (pinned as the header says); the model is this build's own
 * (SURVEY.md section 8(f).4), so for it "parity" means device == this restatement. */
static int static_get_frequency(ox_model *m, size_t symbol, uint64_t *lo, uint64_t *hi)
{
    if (symbol > m->params.symbol_eof)
        return OX_INVALID_INPUT;
    *lo = m->v[symbol];
    *hi = m->v[symbol + 1];
    return OX_OK;
}

static int static_get_symbol(ox_model *m, uint64_t value, size_t *sym, uint64_t *lo, uint64_t *hi)
{
    for (size_t i = 0; i + 1 < m->vlen; i++)
        if (value < m->v[i + 1]) {
            *sym = i;
            *lo  = m->v[i];
            *hi  = m->v[i + 1];
            return OX_OK;
        }
    return OX_INVALID_INPUT;
}

/* cum: symbol_count + 1 entries (one per data symbol and EOF, then the total).  NULL if the table is not usable. */
ox_model *ox_model_new_static(const ox_params *p, const uint64_t *cum)
{
    const size_t n = p->symbol_count + 1; /* symbol_count counts EOF (mod.rs:70) */
    if (cum[0] != 0 || cum[n - 1] > p->freq_max)
        return NULL;
    for (size_t i = 0; i + 1 < n; i++)
        if (cum[i + 1] <= cum[i])
            return NULL;
    ox_model *m = (ox_model *)calloc(1, sizeof *m);
    m->kind   = OX_MODEL_STATIC;
    m->params = *p;
    m->vlen   = n;
    m->v      = (uint64_t *)malloc(n * sizeof(uint64_t));
    memcpy(m->v, cum, n * sizeof(uint64_t));
    return m;
}

ox_model *ox_model_new(int kind, const ox_params *p)
{
    ox_model *m = (ox_model *)calloc(1, sizeof *m);
    m->kind   = kind;
    m->params = *p;
    m->vlen   = p->symbol_count + 1; /* adaptive_linear.rs:23, adaptive_tree.rs:38 */
    m->v      = (uint64_t *)calloc(m->vlen, sizeof(uint64_t));
    if (kind == OX_MODEL_LINEAR) {
        for (size_t i = 1; i < m->vlen; i++) /* adaptive_linear.rs:26-28 */
            m->v[i] = (uint64_t)i;
    } else {
        m->count = (uint64_t)p->symbol_count; /* adaptive_tree.rs:39 */
        for (size_t i = 0; i < m->vlen; i++)  /* adaptive_tree.rs:43-45 */
            m->v[i] = (uint64_t)last_one(i);
    }
    return m;
}

void ox_model_free(ox_model *m)
{
    if (m) {
        free(m->v);
        free(m);
    }
}

uint64_t ox_model_total_frequency(const ox_model *m)
{
    if (m->kind == OX_MODEL_STATIC)
        return m->v[m->vlen - 1];
    return m->kind == OX_MODEL_LINEAR ? linear_total(m) : m->count;
}

int ox_model_get_frequency(ox_model *m, size_t symbol, uint64_t *low, uint64_t *high)
{
    if (m->kind == OX_MODEL_STATIC)
        return static_get_frequency(m, symbol, low, high);
    return m->kind == OX_MODEL_LINEAR ? linear_get_frequency(m, symbol, low, high)
                                      : tree_get_frequency(m, symbol, low, high);
}

int ox_model_get_symbol(ox_model *m, uint64_t value, size_t *symbol, uint64_t *low, uint64_t *high)
{
    if (m->kind == OX_MODEL_STATIC)
        return static_get_symbol(m, value, symbol, low, high);
    return m->kind == OX_MODEL_LINEAR ? linear_get_symbol(m, value, symbol, low, high)
                                      : tree_get_symbol(m, value, symbol, low, high);
}

/* adaptive_linear.rs:73-79, adaptive_tree.rs:139-145 */
void ox_model_get_freq_table(const ox_model *m, uint64_t *lows, uint64_t *highs)
{
    for (size_t i = 0; i < m->params.symbol_count; i++) {
        if (m->kind == OX_MODEL_LINEAR) {
            lows[i]  = m->v[i];
            highs[i] = m->v[i + 1];
        } else {
            lows[i]  = tree_get_frequency_single(m, i);
            highs[i] = tree_get_frequency_single(m, i + 1);
        }
    }
}

/* ======================================================================================
 * src/codec.rs
 * ==================================================================================== */

/* :11-24 */
typedef struct {
    uint64_t  low;
    uint64_t  high;
    uint64_t  pending;
    size_t    extra;
    ox_model *model;
} ox_codec;

static void codec_new(ox_codec *c, ox_model *m) /* :28-36 */
{
    c->low     = m->params.code_min;
    c->high    = m->params.code_max;
    c->pending = 0;
    c->extra   = m->params.code_bits;
    c->model   = m;
}

static int codec_put_bit(ox_codec *self, int bit, ox_bitwriter *output) /* :39-46 */
{
    int e = ox_write_bits(output, bit ? 1 : 0, 1);
    if (e)
        return e;
    while (self->pending > 0) {
        e = ox_write_bits(output, bit ? 0 : 1, 1);
        if (e)
            return e;
        self->pending -= 1;
    }
    return OX_OK;
}

static int codec_get_bit(ox_codec *self, ox_bitreader *input) /* :49-52 */
{
    size_t b;
    int    e = ox_read_bits(input, 1, &b);
    if (e)
        return e;
    self->pending = (self->pending << 1) | (uint64_t)b;
    return OX_OK;
}

static int codec_compress_symbol(ox_codec *self, size_t symbol, ox_bitwriter *output) /* :55-101 */
{
    const ox_params *p = &self->model->params;
    int              e;

    uint64_t count = ox_model_total_frequency(self->model); /* :56 (read BEFORE the update) */
    uint64_t low, high;
    e = ox_model_get_frequency(self->model, symbol, &low, &high); /* :57 */
    if (e)
        return e;
    uint64_t range = self->high - self->low + 1;          /* :58 */
    self->high     = self->low + (range * high / count) - 1; /* :59 */
    self->low      = self->low + (range * low / count);      /* :60 */

    for (;;) { /* :62 */
        if (self->high < p->code_half) { /* :63 */
            e = codec_put_bit(self, 0, output);
            if (e)
                return e;
            if (symbol == p->symbol_eof)
                self->extra -= 1;
        } else if (self->low >= p->code_half) { /* :69 */
            e = codec_put_bit(self, 1, output);
            if (e)
                return e;
            if (symbol == p->symbol_eof)
                self->extra -= 1;
        } else if (self->low >= p->code_one_fourth && self->high < p->code_three_fourths) { /* :75 */
            self->pending += 1;
            self->low -= p->code_one_fourth;
            self->high -= p->code_one_fourth;
            if (symbol == p->symbol_eof)
                self->extra -= 1;
        } else {
            break; /* :84 */
        }
        self->high = ((self->high << 1) + 1) & p->code_max; /* :87 */
        self->low  = (self->low << 1) & p->code_max;        /* :88 */
    }

    if (symbol == p->symbol_eof) { /* :91 */
        while (self->extra > 0) {
            uint64_t mask = self->low & p->code_half; /* :93 */
            e = codec_put_bit(self, mask != 0, output);
            if (e)
                return e;
            self->low = (self->low << 1) & p->code_max; /* :95 */
            self->extra -= 1;
        }
        e = ox_flush_bits(output); /* :98 */
        if (e)
            return e;
    }
    return OX_OK;
}

static int codec_compress_stream(ox_codec *self, ox_bitreader *input, ox_bitwriter *output) /* :104-120 */
{
    const ox_params *p = &self->model->params;
    for (;;) {
        size_t symbol;
        int    e = ox_read_bits(input, p->symbol_bits, &symbol); /* :106 */
        if (e == OX_EOF)
            symbol = p->symbol_eof; /* :108 */
        else if (e)
            return e; /* :109 */
        e = codec_compress_symbol(self, symbol, output); /* :112 */
        if (e)
            return e;
        if (symbol == p->symbol_eof) /* :114 */
            break;
    }
    return OX_OK;
}

static int codec_decompress_symbol(ox_codec *self, ox_bitreader *input, size_t *out_symbol) /* :123-161 */
{
    const ox_params *p = &self->model->params;
    int              e;

    while (self->extra > 0) { /* :124-127 */
        e = codec_get_bit(self, input);
        if (e)
            return e;
        self->extra -= 1;
    }

    uint64_t range = self->high - self->low + 1;                           /* :129 */
    uint64_t count = ox_model_total_frequency(self->model);                /* :130 */
    uint64_t value = ((self->pending - self->low + 1) * count - 1) / range; /* :131 */
    size_t   symbol;
    uint64_t low, high;
    e = ox_model_get_symbol(self->model, value, &symbol, &low, &high); /* :132 */
    if (e)
        return e;
    self->high = self->low + (range * high / count) - 1; /* :133 */
    self->low  = self->low + (range * low / count);      /* :134 */

    if (symbol == p->symbol_eof) { /* :136-138 */
        *out_symbol = symbol;
        return OX_OK;
    }

    for (;;) { /* :140 */
        if (self->high < p->code_half) {
            /* do nothing */
        } else if (self->low >= p->code_half) { /* :143 */
            self->pending -= p->code_half;
            self->low -= p->code_half;
            self->high -= p->code_half;
        } else if (self->low >= p->code_one_fourth && self->high < p->code_three_fourths) { /* :147 */
            self->pending -= p->code_one_fourth;
            self->low -= p->code_one_fourth;
            self->high -= p->code_one_fourth;
        } else {
            break;
        }
        self->low  = self->low << 1;        /* :155 */
        self->high = (self->high << 1) + 1; /* :156 */
        e = codec_get_bit(self, input);     /* :157 */
        if (e)
            return e;
    }
    *out_symbol = symbol;
    return OX_OK;
}

static int codec_decompress_stream(ox_codec *self, ox_bitreader *input, ox_bitwriter *output) /* :164-176 */
{
    const ox_params *p = &self->model->params;
    for (;;) {
        size_t symbol;
        int    e = codec_decompress_symbol(self, input, &symbol);
        if (e)
            return e;
        if (symbol == p->symbol_eof)
            break;
        e = ox_write_bits(output, symbol, p->symbol_bits); /* :171 */
        if (e)
            return e;
    }
    return OX_OK;
}

/* ======================================================================================
 * src/lib.rs:102-120
 * ==================================================================================== */

int ox_compress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                uint64_t *bytes_in, uint64_t *bytes_out)
{
    ox_params p;
    int       e = ox_params_new(symbol_bits, freq_bits, code_bits, &p);
    if (e)
        return e;
    ox_model    *m = ox_model_new(model_kind, &p);
    ox_codec     c;
    ox_bitreader r;
    ox_bitwriter w;
    memset(&r, 0, sizeof r);
    memset(&w, 0, sizeof w);
    r.input  = in;
    r.len    = in_len;
    w.output = out;
    w.cap    = out_cap;
    codec_new(&c, m);                       /* lib.rs:103 */
    e = codec_compress_stream(&c, &r, &w);  /* lib.rs:107 */
    if (bytes_in)
        *bytes_in = r.buffer.count;         /* lib.rs:108 */
    if (bytes_out)
        *bytes_out = w.buffer.count;
    ox_model_free(m);
    return e;
}

int ox_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                  size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                  uint64_t *bytes_in, uint64_t *bytes_out)
{
    ox_params p;
    int       e = ox_params_new(symbol_bits, freq_bits, code_bits, &p);
    if (e)
        return e;
    ox_model    *m = ox_model_new(model_kind, &p);
    ox_codec     c;
    ox_bitreader r;
    ox_bitwriter w;
    memset(&r, 0, sizeof r);
    memset(&w, 0, sizeof w);
    r.input  = in;
    r.len    = in_len;
    w.output = out;
    w.cap    = out_cap;
    codec_new(&c, m);                         /* lib.rs:114 */
    e = codec_decompress_stream(&c, &r, &w);  /* lib.rs:118 */
    if (bytes_in)
        *bytes_in = r.buffer.count;           /* lib.rs:119 */
    if (bytes_out)
        *bytes_out = w.buffer.count;
    ox_model_free(m);
    return e;
}

/* lib.rs:102-120 with the static-table model */
static int static_run(int decode, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t symbol_bits,
                      size_t freq_bits, size_t code_bits, const uint64_t *cum, uint64_t *bytes_in,
                      uint64_t *bytes_out)
{
    ox_params p;
    int       e = ox_params_new(symbol_bits, freq_bits, code_bits, &p);
    if (e)
        return e;
    ox_model *m = ox_model_new_static(&p, cum);
    if (!m)
        return OX_INVALID_INPUT;
    ox_codec     c;
    ox_bitreader r;
    ox_bitwriter w;
    memset(&r, 0, sizeof r);
    memset(&w, 0, sizeof w);
    r.input  = in;
    r.len    = in_len;
    w.output = out;
    w.cap    = out_cap;
    codec_new(&c, m);
    e = decode ? codec_decompress_stream(&c, &r, &w) : codec_compress_stream(&c, &r, &w);
    if (bytes_in)
        *bytes_in = r.buffer.count;
    if (bytes_out)
        *bytes_out = w.buffer.count;
    ox_model_free(m);
    return e;
}

int ox_compress_static(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t symbol_bits,
                       size_t freq_bits, size_t code_bits, const uint64_t *cum, uint64_t *bytes_in,
                       uint64_t *bytes_out)
{
    return static_run(0, in, in_len, out, out_cap, symbol_bits, freq_bits, code_bits, cum, bytes_in, bytes_out);
}

int ox_decompress_static(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t symbol_bits,
                         size_t freq_bits, size_t code_bits, const uint64_t *cum, uint64_t *bytes_in,
                         uint64_t *bytes_out)
{
    return static_run(1, in, in_len, out, out_cap, symbol_bits, freq_bits, code_bits, cum, bytes_in, bytes_out);
}

/* ======================================================================================
 * Block driver -- the build's own chunking (not in the reference): block b is one
 * independent ox_compress() of in[b*block_size .. min(in_len,(b+1)*block_size)).
 * ==================================================================================== */

typedef struct {
    int             decode;
    const uint8_t  *in;
    uint64_t        in_len;
    uint32_t        block_size;
    uint8_t        *out;
    uint64_t        slot_bytes;
    uint32_t       *sizes;       /* encode: out; decode: in */
    uint32_t       *out_sizes;   /* decode only */
    int32_t        *status;
    size_t          sb, fb, cb;
    int             kind;
    uint64_t        b0, b1;
} blk_job;

static void *blk_worker(void *arg)
{
    blk_job *j = (blk_job *)arg;
    for (uint64_t b = j->b0; b < j->b1; b++) {
        uint64_t bi = 0, bo = 0;
        if (!j->decode) {
            uint64_t off = b * (uint64_t)j->block_size;
            uint64_t len = j->in_len - off < j->block_size ? j->in_len - off : j->block_size;
            int e = ox_compress(j->in + off, (size_t)len, j->out + b * j->slot_bytes,
                                (size_t)j->slot_bytes, j->sb, j->fb, j->cb, j->kind, &bi, &bo);
            j->sizes[b]  = (uint32_t)bo;
            j->status[b] = e;
        } else {
            int e = ox_decompress(j->in + b * j->slot_bytes, j->sizes[b],
                                  j->out + b * (uint64_t)j->block_size, j->block_size,
                                  j->sb, j->fb, j->cb, j->kind, &bi, &bo);
            j->out_sizes[b] = (uint32_t)bo;
            j->status[b]    = e;
        }
    }
    return NULL;
}

static int run_jobs(blk_job *proto, uint64_t nblocks, int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    if ((uint64_t)nthreads > nblocks)
        nthreads = nblocks ? (int)nblocks : 1;
    blk_job   *jobs = (blk_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof *tids);
    uint64_t   per  = (nblocks + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        jobs[t]    = *proto;
        jobs[t].b0 = per * (uint64_t)t < nblocks ? per * (uint64_t)t : nblocks;
        jobs[t].b1 = jobs[t].b0 + per < nblocks ? jobs[t].b0 + per : nblocks;
    }
    if (nthreads == 1) {
        blk_worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; t++)
            pthread_create(&tids[t], NULL, blk_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++)
            pthread_join(tids[t], NULL);
    }
    int worst = OX_OK;
    for (uint64_t b = 0; b < nblocks; b++)
        if (proto->status[b] != OX_OK && worst == OX_OK)
            worst = proto->status[b];
    free(jobs);
    free(tids);
    return worst;
}

int ox_compress_blocks(const uint8_t *in, uint64_t in_len, uint32_t block_size,
                       uint8_t *out, uint64_t slot_bytes, uint32_t *sizes, int32_t *status,
                       size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                       int nthreads)
{
    if (block_size == 0)
        return OX_INVALID_INPUT;
    uint64_t nblocks = in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
    blk_job  j;
    memset(&j, 0, sizeof j);
    j.in = in; j.in_len = in_len; j.block_size = block_size; j.out = out;
    j.slot_bytes = slot_bytes; j.sizes = sizes; j.status = status;
    j.sb = symbol_bits; j.fb = freq_bits; j.cb = code_bits; j.kind = model_kind;
    return run_jobs(&j, nblocks, nthreads);
}

int ox_decompress_blocks(const uint8_t *in, uint64_t slot_bytes, const uint32_t *sizes,
                         uint64_t nblocks, uint8_t *out, uint32_t block_size,
                         uint32_t *out_sizes, int32_t *status,
                         size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                         int nthreads)
{
    blk_job j;
    memset(&j, 0, sizeof j);
    j.decode = 1;
    j.in = in; j.block_size = block_size; j.out = out; j.slot_bytes = slot_bytes;
    j.sizes = (uint32_t *)sizes; j.out_sizes = out_sizes; j.status = status;
    j.sb = symbol_bits; j.fb = freq_bits; j.cb = code_bits; j.kind = model_kind;
    return run_jobs(&j, nblocks, nthreads);
}

/* ======================================================================================
 * Differential driver for src/model/tests.rs:50-93 (compare_models_{encode,decode}_single):
 * runs `iter` random operations on a linear and a tree model side by side and returns -1
 * if every (low,high)/(symbol,low,high), every total and (every `table_every` steps) the
 * full frequency table agreed and the two invalid-input probes were rejected; otherwise
 * the iteration index of the first disagreement (or -2 / -3 for the error probes).
 * The reference draws from an unseeded rand::random; here a seeded splitmix64.
 * ==================================================================================== */
static uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int64_t ox_selftest_models(size_t bits, size_t freq, size_t code, uint64_t iter, uint64_t seed,
                           int decode, uint64_t table_every)
{
    ox_params p;
    if (ox_params_new(bits, freq, code, &p))
        return -4;
    ox_model *lin = ox_model_new(OX_MODEL_LINEAR, &p);
    ox_model *tre = ox_model_new(OX_MODEL_TREE, &p);
    uint64_t *tl = (uint64_t *)malloc(4 * p.symbol_count * sizeof(uint64_t));
    uint64_t *th = tl + p.symbol_count, *ll = th + p.symbol_count, *lh = ll + p.symbol_count;
    int64_t   bad = -1;
    for (uint64_t it = 0; it < iter && bad == -1; it++) {
        if (ox_model_total_frequency(lin) != ox_model_total_frequency(tre)) /* tests.rs:31-35 */
            bad = (int64_t)it;
        if (table_every && it % table_every == 0) { /* tests.rs:37-43 */
            ox_model_get_freq_table(lin, ll, lh);
            ox_model_get_freq_table(tre, tl, th);
            if (memcmp(ll, tl, p.symbol_count * 8) || memcmp(lh, th, p.symbol_count * 8))
                bad = (int64_t)it;
        }
        uint64_t a, b, c, d;
        if (!decode) {
            size_t symbol = (size_t)(sm64(&seed) % (((uint64_t)1 << bits) + 1)); /* tests.rs:19-21 */
            int e1 = ox_model_get_frequency(lin, symbol, &a, &b);
            int e2 = ox_model_get_frequency(tre, symbol, &c, &d);
            if (e1 || e2 || a != c || b != d)
                bad = (int64_t)it;
        } else {
            uint64_t value = sm64(&seed) % ox_model_total_frequency(lin); /* tests.rs:27-29 */
            size_t s1, s2;
            int e1 = ox_model_get_symbol(lin, value, &s1, &a, &b);
            int e2 = ox_model_get_symbol(tre, value, &s2, &c, &d);
            if (e1 || e2 || s1 != s2 || a != c || b != d)
                bad = (int64_t)it;
        }
    }
    if (bad == -1) {
        uint64_t a, b;
        size_t   s;
        if (!decode) { /* tests.rs:65-69 */
            size_t inv = ((size_t)1 << bits) + 1;
            if (ox_model_get_frequency(lin, inv, &a, &b) == OX_OK || ox_model_get_frequency(lin, inv + 1, &a, &b) == OX_OK ||
                ox_model_get_frequency(tre, inv, &a, &b) == OX_OK || ox_model_get_frequency(tre, inv + 1, &a, &b) == OX_OK)
                bad = -2;
        } else { /* tests.rs:88-92 */
            uint64_t inv = ox_model_total_frequency(lin);
            if (ox_model_get_symbol(lin, inv, &s, &a, &b) == OX_OK || ox_model_get_symbol(lin, inv + 1, &s, &a, &b) == OX_OK ||
                ox_model_get_symbol(tre, inv, &s, &a, &b) == OX_OK || ox_model_get_symbol(tre, inv + 1, &s, &a, &b) == OX_OK)
                bad = -3;
        }
    }
    free(tl);
    ox_model_free(lin);
    ox_model_free(tre);
    return bad;
}
