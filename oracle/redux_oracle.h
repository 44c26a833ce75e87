/*
 * redux_oracle.h -- CPU restatement of peterbudai/redux's encode/decode hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (redux_amd/) never
 * links, imports or calls anything in oracle/; it fails loudly if the HIP library is
 * missing.
 *
 * PARITY PINNING.  The reference is Rust and no Rust toolchain exists in this image, so
 * the reference itself cannot be run here.  The reference's own tests hold NO compressed
 * golden byte string.  What pins this restatement:
 *   - bit I/O: the 9 known-answer tests of /root/reference/src/bitio/tests.rs (exact bytes);
 *   - model:   linear == tree differential tests at the 14 parameter triples of
 *              /root/reference/src/model/tests.rs, including the error cases;
 *   - codec:   round-trip identity + byte counts over every corpus file x 2 models x 3
 *              widths (/root/reference/tests/corpora.rs), the doc-test of src/lib.rs:23-39,
 *              the four hand-traced streams of SURVEY.md section 8c, and a second,
 *              independently written pure-Python restatement (oracle/redux_ref.py).
 * The compressed bitstream as such is therefore pinned by source restatement only:
 * "bit-exact" everywhere in this repo means bit-exact to this line-faithful restatement.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef REDUX_ORACLE_H
#define REDUX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:57-64  enum Error { Eof, InvalidInput, IoError } ; 0 = Ok */
enum { OX_OK = 0, OX_EOF = 1, OX_INVALID_INPUT = 2, OX_IO_ERROR = 3 };

/* model kinds: src/model/adaptive_linear.rs, src/model/adaptive_tree.rs */
enum { OX_MODEL_LINEAR = 0, OX_MODEL_TREE = 1, OX_MODEL_STATIC = 2 /* this build's own, see ox_model_new_static */ };

/* src/model/mod.rs:32-59  struct Parameters */
typedef struct {
    size_t   symbol_bits;
    size_t   symbol_eof;
    size_t   symbol_count;
    size_t   freq_bits;
    uint64_t freq_max;
    size_t   code_bits;
    uint64_t code_min;
    uint64_t code_one_fourth;
    uint64_t code_half;
    uint64_t code_three_fourths;
    uint64_t code_max;
} ox_params;

/* src/model/mod.rs:63-81  Parameters::new */
int ox_params_new(size_t symbol, size_t frequency, size_t code, ox_params *out);

/* ---- bit I/O (src/bitio/mod.rs) ------------------------------------------------ */
typedef struct ox_bitreader ox_bitreader;
typedef struct ox_bitwriter ox_bitwriter;

ox_bitreader *ox_bitreader_new(const uint8_t *data, size_t len);      /* :63 */
void          ox_bitreader_free(ox_bitreader *r);
uint64_t      ox_bitreader_count(const ox_bitreader *r);              /* :71 */
int           ox_read_bits(ox_bitreader *r, size_t bits, size_t *result); /* :78 */

ox_bitwriter *ox_bitwriter_new(uint8_t *data, size_t cap);            /* :133 */
void          ox_bitwriter_free(ox_bitwriter *w);
uint64_t      ox_bitwriter_count(const ox_bitwriter *w);              /* :141 */
int           ox_write_bits(ox_bitwriter *w, size_t symbol, size_t bits); /* :148 */
int           ox_flush_bits(ox_bitwriter *w);                         /* :183 */

/* ---- models (src/model/mod.rs:17-29 trait Model) -------------------------------- */
typedef struct ox_model ox_model;

ox_model *ox_model_new(int kind, const ox_params *p);   /* adaptive_linear.rs:21, adaptive_tree.rs:36 */
void      ox_model_free(ox_model *m);
uint64_t  ox_model_total_frequency(const ox_model *m);
int       ox_model_get_frequency(ox_model *m, size_t symbol, uint64_t *low, uint64_t *high);
int       ox_model_get_symbol(ox_model *m, uint64_t value, size_t *symbol, uint64_t *low, uint64_t *high);
/* debug-only get_freq_table (mod.rs:27): fills symbol_count (low,high) pairs */
void      ox_model_get_freq_table(const ox_model *m, uint64_t *lows, uint64_t *highs);

/* ---- whole-stream API (src/lib.rs:102-120) -------------------------------------- */
/* compress: returns status; *bytes_in / *bytes_out are the (u64,u64) tuple of lib.rs:108. */
/* Static-table model (not in the reference): cum has symbol_count + 1
 * entries, cum[0] = 0, strictly increasing, cum[last] = total <= freq_max. */
ox_model *ox_model_new_static(const ox_params *p, const uint64_t *cum);
int ox_compress_static(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t symbol_bits,
                       size_t freq_bits, size_t code_bits, const uint64_t *cum, uint64_t *bytes_in,
                       uint64_t *bytes_out);
int ox_decompress_static(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t symbol_bits,
                         size_t freq_bits, size_t code_bits, const uint64_t *cum, uint64_t *bytes_in,
                         uint64_t *bytes_out);

int ox_compress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                uint64_t *bytes_in, uint64_t *bytes_out);
int ox_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                  size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                  uint64_t *bytes_in, uint64_t *bytes_out);

/* ---- block driver (NOT in the reference: the build's chunking) ----
 * Each block is one independent ox_compress() call with a fresh model.  Block b's
 * stream is written at out + b*slot_bytes; sizes[b] receives its length, status[b] its
 * status.  nthreads>1 uses a static contiguous partition over pthreads. */
int ox_compress_blocks(const uint8_t *in, uint64_t in_len, uint32_t block_size,
                       uint8_t *out, uint64_t slot_bytes, uint32_t *sizes, int32_t *status,
                       size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                       int nthreads);
int ox_decompress_blocks(const uint8_t *in, uint64_t slot_bytes, const uint32_t *sizes,
                         uint64_t nblocks, uint8_t *out, uint32_t block_size,
                         uint32_t *out_sizes, int32_t *status,
                         size_t symbol_bits, size_t freq_bits, size_t code_bits, int model_kind,
                         int nthreads);

/* ---- differential self-test of the two models (src/model/tests.rs:50-93) ---------- */
int64_t ox_selftest_models(size_t bits, size_t freq, size_t code, uint64_t iter, uint64_t seed,
                           int decode, uint64_t table_every);

#ifdef __cplusplus
}
#endif
#endif
