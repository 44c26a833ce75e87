/*
 * redux_hip.h -- C ABI of the MI355X-native block coder that sits behind peterbudai/redux's
 * compress / decompress surface.  Plain pointers and sizes only; no HIP or torch types.
 *
 * Every entry point names the reference interface it replaces (file:line under the
 * reference checkout).  The reference-side binding a maintainer would add (Rust
 * `extern "C"` block + safe wrappers) is shown in INTEGRATION.md.
 *
 * Semantics shared by all calls
 *   - A "block" is block_size consecutive input bytes (the last one may be shorter; an
 *     empty input is ONE empty block).  Each block is coded by a fresh Codec + fresh
 *     AdaptiveTreeModel, so block b's stream is byte-identical to
 *     redux::compress(&mut &in[b*block_size..], .., AdaptiveTreeModel::new(params))
 *     (src/lib.rs:102, src/codec.rs:104, src/model/adaptive_tree.rs:36).
 *   - Status codes mirror src/lib.rs:57-64: 0 Ok, 1 Eof, 2 InvalidInput, 3 IoError (here: a
 *     HIP runtime failure), plus 4 OutputTooSmall and 5 Unsupported (parameters the device
 *     path does not implement: symbol_bits > 16; symbol_bits == 8 with code_bits <= 32 runs on
 *     the fast kernels, 4- and 12-bit symbols with code_bits <= 32 on lock-step kernels of the
 *     same form, every other valid triple on a one-lane-per-block kernel).  There is NO
 *     CPU fallback: Unsupported is returned, never silently served by other code.
 *   - The caller owns every buffer.  Host-pointer calls are synchronous.  `_dev` calls take
 *     device pointers, enqueue on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream), never allocate, never synchronise and keep no pointer after they return.
 *   - Thread-safe.  The `_dev` calls and the geometry helpers keep no state at all.  The
 *     host-pointer calls share ONE lazily created, mutex-guarded context per device (chunk slots
 *     in HBM, pinned staging, streams; grown on demand, buffers above 1 GiB given back when their call ends, freed by
 *     redux_host_release()): calls from several host threads are safe; two that use the same
 *     context run one after the other, two on different devices run concurrently.
 *   - The `_dev` calls launch on HIP's CURRENT device; every pointer must belong to it.
 */
#ifndef REDUX_HIP_H
#define REDUX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:57-64 enum Error (+ two codes the block API needs) */
enum {
    REDUX_OK               = 0,
    REDUX_EOF              = 1, /* Error::Eof: compressed input ended early (bitio/mod.rs:107) */
    REDUX_INVALID_INPUT    = 2, /* Error::InvalidInput (model/mod.rs:65, adaptive_tree.rs:131) */
    REDUX_IO_ERROR         = 3, /* Error::IoError: here a HIP runtime error */
    REDUX_OUTPUT_TOO_SMALL = 4, /* the caller's output buffer / slot cannot hold the result */
    REDUX_UNSUPPORTED      = 5  /* valid Parameters the device path does not implement */
};

/* The three integers of model::Parameters::new (src/model/mod.rs:63); the library derives
 * the other eight fields (mod.rs:67-79) itself. */
typedef struct redux_params {
    uint32_t symbol_bits; /* 1..16 on the device (8: fast kernels) */
    uint32_t freq_bits;
    uint32_t code_bits;   /* <= 32 with symbol_bits 8: fast kernels; else general path */
} redux_params;

/* One entry of a block table (the `_v` calls below): which bytes a block is, and which entry of the
 * per-block result tables is its.  16 bytes, host or device memory. */
typedef struct redux_block {
    uint64_t offset; /* encode: the block's first byte in the input buffer; decode: where its output starts */
    uint32_t length; /* encode: bytes in the block (<= block_size); decode: output capacity (<= block_size) */
    uint32_t index;  /* the block's number: its entry in out_offsets / in_offsets / out_sizes / block_status;
                        REDUX_BLOCK_IDLE: no block, the lane idles (padding, see redux_block_table_v) */
} redux_block;
#define REDUX_BLOCK_IDLE 0xFFFFFFFFu

/* model::Parameters::new validation, src/model/mod.rs:64: OK or INVALID_INPUT. */
int redux_params_check(uint32_t symbol_bits, uint32_t freq_bits, uint32_t code_bits);
/* OK, INVALID_INPUT, or UNSUPPORTED for valid triples outside the device path. */
int redux_device_supports(const redux_params *p);

/* Geometry of the block API. */
uint64_t redux_block_count(uint64_t in_len, uint32_t block_size); /* max(1, ceil(in_len / block_size)) */
/* Worst-case bytes of one block's stream for these parameters (9 bits/symbol while the model
 * cannot freeze inside a block, freq_bits+2 bits/symbol once it can, + 1 KiB). */
uint64_t redux_encode_slot_bytes(const redux_params *p, uint32_t block_size);
/* Dense-output capacity that always suffices: block_count * slot_bytes. */
uint64_t redux_encode_bound(const redux_params *p, uint64_t in_len, uint32_t block_size);
/* Workspace of an encode call on the device: the blocks' slots, the reciprocal table and -- for launches the small-grid
 * kernels take -- the (low, high) pairs: 8 bytes per input byte for at most 2048 blocks of up to 64 KiB; for at most 24,576
 * blocks above 64 KiB (one block of any length included: redux_compress) TWO windows of at most 65,504 symbols per block
 * (1 MiB), at most 2816 MiB in all, because such blocks are coded window by window, the model of one window next to the
 * coder of the one before: the workspace of a long stream is its slot (9/8 of its length) + 2 MiB, whatever its length. */
uint64_t redux_encode_workspace_bytes(const redux_params *p, uint64_t in_len, uint32_t block_size);
uint64_t redux_decode_workspace_bytes(const redux_params *p, uint64_t nblocks, uint32_t block_size);

/* ---- host-pointer, synchronous ------------------------------------------------------
 * Inside, a call is a pipeline over chunks of whole blocks (~64 MiB): CPU threads stage the
 * caller's bytes into pinned memory, H2D, the chunk's kernels on its own stream, D2H -- so the
 * transfers of later chunks hide under the kernels of earlier ones (redux_amd/csrc/redux_host.hpp).
 * No hipMalloc / hipHostMalloc happens in the steady state (same or smaller shapes than before).
 *
 * redux_encode_blocks: replaces one redux::compress call per block (src/lib.rs:102-109).
 *   out          dense concatenation of the per-block streams
 *   out_offsets  nblocks+1 entries; block b's stream is out[out_offsets[b] .. out_offsets[b+1])
 *                (so (bytes_in, bytes_out) of lib.rs:108 = (block length, offsets[b+1]-offsets[b]))
 *   block_status nblocks entries, per-block status (may be NULL)
 * Returns the first non-OK per-block status, or a call-level error. */
int redux_encode_blocks(const redux_params *p, const uint8_t *in, uint64_t in_len, uint32_t block_size,
                        uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, int32_t *block_status);

/* redux_decode_blocks: replaces one redux::decompress call per block (src/lib.rs:113-120).
 *   in / in_offsets  dense streams as produced by redux_encode_blocks
 *   out              block b is written at out + b*block_size (at most block_size bytes)
 *   out_sizes        nblocks entries: decoded length of each block
 * Per-block status: EOF for a truncated stream, INVALID_INPUT for a code value outside the
 * model's total, OUTPUT_TOO_SMALL if a (corrupt) stream decodes past block_size. */
int redux_decode_blocks(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                        uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes,
                        int32_t *block_status);

/* ---- many independent inputs in one call ("_v": a vector of inputs) ----------------------
 * The reference's harness codes every file by itself, one redux::compress / decompress per file
 * (tests/corpora.rs:32-85).  These calls do that for `ninputs` inputs at once: input i is the
 * bytes in[in_off[i] .. in_off[i] + in_len[i]) and is cut into blocks of block_size ON ITS OWN
 * (its last block may be shorter; an empty input is one empty block), so every block's stream
 * equals the one redux_encode_blocks produces when it is called for that input alone.  Blocks
 * are numbered input by input: input i owns blocks first(i) .. first(i) + redux_block_count(
 * in_len[i], block_size) - 1, first(0) = 0.  One launch codes all of them.
 *
 * redux_block_count_v   total number of blocks.
 * redux_block_table_v   the block table of these inputs in LAUNCH order.  A wave (64 consecutive
 *                       entries) runs its fast path for as long as its SHORTEST block lasts, so
 *                       the table puts whole blocks first, 64 to a wave, then the shorter ones by
 *                       decreasing length, and starts a new wave wherever the length has dropped
 *                       by an eighth: the rest of the old wave is filled with IDLE entries
 *                       (index = REDUX_BLOCK_IDLE; on a chip with 1024 wave slots a batch of a
 *                       few hundred blocks has lanes to spare).  entry.index = the block's number
 *                       otherwise.  Returns the number of ENTRIES (>= blocks); `table` may be
 *                       NULL (count only).
 * redux_encode_blocks_v out / out_offsets (total blocks + 1) / block_status (may be NULL) as in
 *                       redux_encode_blocks, in block-number order.
 * redux_decode_blocks_v the inverse: in / in_offsets as produced above; input i's blocks are
 *                       written to out[out_off[i] ..) back to back, at most out_len[i] bytes
 *                       (a block of input i may hold min(block_size, what is left of out_len[i]));
 *                       out_sizes / block_status per block.
 * Device path: symbol_bits == 8 and code_bits <= 32 (other valid triples: REDUX_UNSUPPORTED from
 * these calls only -- call redux_encode_blocks per input instead). */
uint64_t redux_block_count_v(const uint64_t *in_len, uint64_t ninputs, uint32_t block_size);
uint64_t redux_block_table_v(const uint64_t *in_off, const uint64_t *in_len, uint64_t ninputs, uint32_t block_size,
                             redux_block *table);
int redux_encode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                          uint64_t ninputs, uint32_t block_size, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets,
                          int32_t *block_status);
int redux_decode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint8_t *out,
                          const uint64_t *out_off, const uint64_t *out_len, uint64_t ninputs, uint32_t block_size,
                          uint32_t *out_sizes, int32_t *block_status);

/* Whole-stream drop-ins for redux::compress / redux::decompress (src/lib.rs:102,113): the
 * input is ONE block of any length, so the stream equals the reference's for the same bytes.
 * One coder = one chain of dependent symbols: correct but serial (the model is computed by a wave, the
 * interval chain by one lane; the decoder is one wave per stream); the block API is the accelerated path.
 * bytes_in / bytes_out are the (u64,u64) the reference returns.
 * LIMIT: one block is at most 0xFFFFFF00 bytes (symbol index, byte offsets and the consumed-bit count of a lane
 * are 32-bit in the kernels): redux_compress returns REDUX_UNSUPPORTED for a longer input, redux_decompress
 * clamps out_cap to it (a stream that decodes to more comes back REDUX_OUTPUT_TOO_SMALL).  The reference takes
 * any io::Read (src/lib.rs:102); a caller with more than 4 GiB in ONE stream has no parallelism to gain here
 * (~10 MB/s against ~14 MB/s for one CPU thread) and should use the block API or the CPU. */
int redux_compress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                   uint64_t *bytes_in, uint64_t *bytes_out);
int redux_decompress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                     uint64_t *bytes_in, uint64_t *bytes_out);

/* Several GPUs behind the host-pointer calls.  By default a call runs on HIP's current device.
 * redux_host_set_devices(ids, n) (n <= 16; n = 0: back to the default) makes every later
 * redux_encode_blocks / redux_decode_blocks deal its chunks round-robin over n contexts, context i on
 * device ids[i] -- chunk k goes to context k mod n -- each with its own streams, HBM slots and
 * pinned staging, each fed over its own PCIe link by its own host threads; the data starts and ends
 * in host memory, so the devices exchange nothing (no collective).  An id may appear more than once
 * (two contexts on one device: what the one-GPU test box exercises).  The `_v` calls deal their GROUPS of inputs
 * (512 MiB of payload each) the same way, group k on context k mod n.
 * Existing contexts are released by the call.  Returns INVALID_INPUT for an id that is not a device.
 * redux_host_chunk_plan: the chunking such a call uses for nblocks blocks on ncontexts contexts
 * (whole waves of 64 blocks per chunk); host arithmetic only.
 * redux_host_set_chunk_bytes: overrides the chunk size limits (bytes of payload per chunk; 0, 0 = the
 * defaults, 16 MiB .. 128 / 256 MiB): a harness uses it to drive many chunks through a small input. */
int  redux_host_set_devices(const int32_t *device_ids, uint32_t n);
int  redux_host_chunk_plan(uint64_t nblocks, uint32_t block_size, uint32_t ncontexts, int decode, uint64_t *chunk_blocks,
                           uint64_t *nchunks);
int  redux_host_set_chunk_bytes(uint64_t min_bytes, uint64_t max_bytes);

/* Frees every per-device context of the host-pointer calls (they are rebuilt on the next call).
 * redux_host_allocations: hipMalloc + hipHostMalloc calls the contexts have made so far -- a
 * harness checks that it does not move in the steady state. */
int      redux_host_release(void);
uint64_t redux_host_allocations(void);
/* Device memory the contexts hold right now, in bytes.  A context keeps what the chunk pipeline can ask for and gives
 * back slot buffers above 1 GiB when the call that needed them ends (one block of hundreds of MiB, a generous decode
 * capacity); the decoders' workspace does not grow with the capacity at all (a bounded reciprocal table). */
uint64_t redux_host_resident_bytes(void);
/* Diagnostic: the timeline of the last host-pointer call on the current device, four doubles per
 * chunk (seconds since the call began): staging begins, device work enqueued, kernels done, results
 * in caller memory.  Copies up to cap doubles, returns how many there are. */
uint64_t redux_host_trace(double *out, uint64_t cap);

/* ---- device-pointer, stream-ordered --------------------------------------------------
 * Same contracts with every pointer in device memory.  d_workspace must hold
 * redux_encode_workspace_bytes() / redux_decode_workspace_bytes() bytes and be 256-B aligned.
 * d_summary (int32[2], may be NULL): [0] = first non-OK status over all blocks (0 if none),
 * [1] = number of non-OK blocks.  Errors found on the device are reported there and in
 * d_block_status; the return value covers argument and launch errors only. */
int redux_encode_blocks_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                            void *d_out, uint64_t out_cap, void *d_out_offsets /* u64[nblocks+1] */,
                            void *d_block_status /* i32[nblocks] */, void *d_summary /* i32[2] */,
                            void *d_workspace, uint64_t workspace_bytes, void *stream);
int redux_decode_blocks_dev(const redux_params *p, const void *d_in, const void *d_in_offsets /* u64[nblocks+1] */,
                            uint64_t nblocks, uint32_t block_size, void *d_out, uint64_t out_cap,
                            void *d_out_sizes /* u32[nblocks] */, void *d_block_status, void *d_summary,
                            void *d_workspace, uint64_t workspace_bytes, void *stream);

/* The `_v` calls with everything in device memory.  d_table: redux_block[nentries] in launch order
 * (redux_block_table_v builds one; any order is valid, a wave's 64 consecutive entries run in
 * lock-step for as long as its shortest block lasts).  The table is CHECKED on the device before any
 * coder kernel reads it (one thread per entry, tens of microseconds; the kernels then read a copy in the
 * workspace, the caller's table is never written): an entry must be idle (index REDUX_BLOCK_IDLE) or have
 * index < nblocks, an index no other entry has, length <= block_size, offset + length <= in_bytes /
 * out_bytes and, under REDUX_V_ALIGNED16, offset a multiple of 16.  An entry that fails is treated as
 * idle -- nothing is read or written through it --, a block that no valid entry codes comes back with
 * size 0 and status REDUX_INVALID_INPUT, and d_summary[0] is REDUX_INVALID_INPUT (the return value covers
 * argument and launch errors only, as everywhere: the call is stream-ordered).  As the reference's
 * surface never writes out of bounds (bitio/mod.rs:148-198 returns Err), neither does a bad table.  Encode: entry.offset is the block's first
 * byte in d_in (in_bytes = size of that buffer, below 4 GiB), entry.length its size.  Decode:
 * entry.offset is where the block's output starts in d_out (out_bytes = size of that buffer),
 * entry.length the room it has there.  flags: REDUX_V_ALIGNED16 promises that d_in / d_out and
 * every entry.offset are multiples of 16 (the fast kernels need it; without it the call is
 * correct and slow).  nentries = entries of the table (idle ones included), nblocks = blocks.
 * Workspace: redux_encode_workspace_bytes(p, nentries * block_size, block_size) /
 * redux_decode_workspace_bytes(p, nentries, block_size). */
enum { REDUX_V_ALIGNED16 = 1 };
int redux_encode_blocks_v_dev(const redux_params *p, const void *d_in, uint64_t in_bytes, const void *d_table,
                              uint64_t nentries, uint64_t nblocks, uint32_t block_size, uint32_t flags, void *d_out,
                              uint64_t out_cap, void *d_out_offsets /* u64[nblocks+1] */, void *d_block_status /* i32[nblocks] */,
                              void *d_summary, void *d_workspace, uint64_t workspace_bytes, void *stream);
int redux_decode_blocks_v_dev(const redux_params *p, const void *d_in, const void *d_in_offsets /* u64[nblocks+1] */,
                              const void *d_table, uint64_t nentries, uint64_t nblocks, uint32_t block_size, uint32_t flags,
                              void *d_out, uint64_t out_bytes, void *d_out_sizes /* u32[nblocks] */, void *d_block_status,
                              void *d_summary, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* The two phases of redux_encode_blocks_dev, exposed so a harness can time the coder kernel
 * by itself: (1) code every block into its padded slot inside the workspace and record the
 * sizes; (2) scan the sizes and gather the slots into the dense output.  Both take the same
 * (in_len, block_size, d_workspace, workspace_bytes): the layout inside the workspace is a function of
 * the shape AND of workspace_bytes (a workspace without room for the small-launch kernels' pairs area
 * makes a small launch run the full-grid kernels on their layout), so the phases must be told the same. */
int redux_encode_slots_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                           void *d_block_status, void *d_workspace, uint64_t workspace_bytes, void *stream);
int redux_compact_slots_dev(const redux_params *p, uint64_t in_len, uint32_t block_size, void *d_out, uint64_t out_cap,
                            void *d_out_offsets, void *d_block_status, void *d_summary,
                            void *d_workspace, uint64_t workspace_bytes, void *stream);

/* ---- synthetic workloads of BASELINE.json, generated in HBM ----------
 * iid : byte j = byte (j mod 8), little-endian, of splitmix64(seed + j/8).
 * zipf: byte j = rank-1 where rank is drawn by inverse CDF of P(r) ~ r^-1.2, r = 1..256,
 *       from the high 32 bits of splitmix64(seed + j) against the 256-entry u32 table
 *       returned by redux_zipf_thresholds() (thresholds[r-1] = floor(2^32 * CDF(r)) , last = 2^32-1).
 * first_byte lets a rank generate its own shard: byte j of the call is stream byte first_byte + j. */
int redux_gen_iid_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream);
int redux_gen_zipf_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream);
const uint32_t *redux_zipf_thresholds(void);

/* ---- static-table model ------------------------------------------
 * The reference's Codec is generic over its Model trait (src/model/mod.rs; lib.rs:14-15 invites
 * custom models).  The cheapest second model is a fixed table: cum[0..=257] (host memory, 258
 * entries) with cum[0] = 0, cum strictly increasing and cum[257] = total_frequency() <= freq_max.
 * get_frequency(s) = [cum[s], cum[s+1]) for the data symbols 0..255 and for EOF = 256; nothing is
 * updated.  Block b's stream is Codec::compress_stream (codec.rs:104-120) of that block under
 * such a model.  Device path: symbol_bits == 8, code_bits <= 32 (else REDUX_UNSUPPORTED); a table
 * that is not strictly increasing or exceeds freq_max is REDUX_INVALID_INPUT.
 * Same buffers and error reporting as redux_encode_blocks_dev / redux_decode_blocks_dev. */
int      redux_static_table_check(const redux_params *p, const uint32_t *cum);
uint64_t redux_static_encode_bound(const redux_params *p, uint64_t in_len, uint32_t block_size);
uint64_t redux_static_encode_workspace_bytes(const redux_params *p, uint64_t in_len, uint32_t block_size);
int redux_static_encode_blocks_dev(const redux_params *p, const uint32_t *cum, const void *d_in, uint64_t in_len,
                                   uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_offsets,
                                   void *d_block_status, void *d_summary, void *d_workspace,
                                   uint64_t workspace_bytes, void *stream);
int redux_static_decode_blocks_dev(const redux_params *p, const uint32_t *cum, const void *d_in,
                                   const void *d_in_offsets, uint64_t nblocks, uint32_t block_size, void *d_out,
                                   uint64_t out_cap, void *d_out_sizes, void *d_block_status, void *d_summary,
                                   void *stream);

/* Library / build identification: "redux_hip <version> gfx950". */
const char *redux_version(void);
/* sha256 (first 16 hex digits) of the kernel sources + this header the library was built from ("unknown" when the
 * build did not say): a profile records it, and a harness borrows a profiled figure only for the same sources. */
const char *redux_source_hash(void);

/* Which kernel redux_encode_slots_dev / redux_decode_blocks_dev launch for these arguments (the
 * same decision function the launch code uses; d_in / d_out only contribute their alignment).  A static
 * string; "" for invalid arguments.  A harness reports it next to its timings (bench.py's
 * roofline.kernel) instead of assuming the fast path was taken. */
const char *redux_encode_kernel_name(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size);
const char *redux_decode_kernel_name(const redux_params *p, const void *d_out, uint32_t block_size);
/* The decoder also depends on the SIZE of the launch (blocks the lock-step decoder does not take -- above 64 KiB -- run one
 * per wave in launches of at most 1024 blocks, k_decode_wave, and on the cell decoder with u32 nodes, k_decode_cells<8>, in
 * larger ones; 11- and 12-bit symbols keep their bottom tree cells in LDS on small grids):
 * nblocks = blocks (or table entries) of the launch; 0 = a grid that fills the chip, which is what
 * redux_decode_kernel_name answers for. */
const char *redux_decode_kernel_name_n(const redux_params *p, const void *d_out, uint32_t block_size, uint64_t nblocks);

/* Diagnostic, used by the parity tests only: *max_err = max over the integers x in [lo, hi] of
 * |v_rcp_f64(x) * x - 1| evaluated on the device.  The decoder's code-value division
 * (codec.rs:131) multiplies by the raw hardware reciprocal of `range` (an integer in [1, 2^32])
 * and relies on this error staying below 2^-24 over that whole range. */
int redux_debug_rcp_check(uint64_t lo, uint64_t hi, double *max_err);

/* Diagnostic, used by the parity tests only: byte offset and length, inside the encode workspace
 * of (p, in_len, block_size), of the encoder's per-CU role book (see redux_encode.hpp).  Every
 * encode workgroup returns its booking when it ends, so the region reads all-zero after a
 * completed redux_encode_slots_dev / redux_encode_blocks_dev. */
int redux_debug_role_book(const redux_params *p, uint64_t in_len, uint32_t block_size, uint64_t *offset,
                          uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
