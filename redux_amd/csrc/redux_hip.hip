// redux_hip.hip -- the C ABI of include/redux_hip.h over the gfx950 kernels.
//
// One translation unit; the kernels (all hand-written for CDNA4, wave64) live in
//   redux_coder.hpp    device building blocks: LDS tree, interval narrowing, bit output
//   redux_encode.hpp   k_fill_rc, k_encode, k_encode_pair (default encoder)
//   redux_decode.hpp   k_decode, k_decode_lock (default decoder); redux_decode_wave.hpp: k_decode_wave (small launches, whole streams)
//   redux_pack.hpp     k_scan_sizes, k_compact: slots -> dense stream + offsets
//   redux_coop.hpp     k_coop_model, k_coop_chain: small grids, a block's model computed by 64 lanes
//   redux_any.hpp      general Parameters (symbol_bits <= 16, code_bits <= 63), one lane per block
//   redux_gen.hpp      k_encode_gen / k_encode_gen_pair: symbol widths 1 .. 12 other than 8 (code_bits <= 32) in lock-step form
//   redux_decode_cells.hpp  k_decode_cells: their decoder, the tree as cells of four levels
//   redux_synth.hpp    k_gen_iid / k_gen_zipf
//   redux_static.hpp   k_encode_static / k_decode_static: the coder core under a fixed frequency table
// This file holds the general-parameter kernels' launch shims, the workspace geometry and the
// extern "C" entry points.
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC redux_hip.hip -o libredux_hip.so
#include "redux_coder.hpp"
#include "redux_any.hpp"
#include "redux_gen.hpp"
#include "redux_encode.hpp"
#include "redux_decode.hpp"
#include "redux_decode_adaptive.hpp"
#include "redux_decode_wave.hpp"
#include "redux_decode_cells.hpp"
#include "redux_pack.hpp"
#include "redux_table.hpp"
#include "redux_coop.hpp"
#include "redux_synth.hpp"
#include "redux_static.hpp"

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace redux {


// ======================================================================================
// general parameters (redux_any.hpp): one lane per block, tree in the workspace
// ======================================================================================
struct AnyEncArgs {
    const uint8_t *in;
    uint64_t       in_len, nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    uint32_t      *trees;
    uint64_t       tree_words; // u32 entries per block
    uint32_t       block_size, slot_cap;
    uint32_t       sb, fb, cb;
};

__global__ void __launch_bounds__(64) k_encode_any(AnyEncArgs a)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.nblocks)
        return;
    const uint64_t o0  = blk * a.block_size;
    const uint64_t rem = a.in_len > o0 ? a.in_len - o0 : 0;
    const uint64_t len = rem < a.block_size ? rem : a.block_size;
    const any::Params P = any::make_params(a.sb, a.fb, a.cb);
    uint64_t  bi, bo;
    const int st = any::compress_stream(P, a.trees + blk * a.tree_words, a.in + o0, len, a.slots + blk * a.slot_bytes,
                                        a.slot_cap, bi, bo);
    a.sizes[blk]  = (uint32_t)bo;
    a.status[blk] = st;
}

struct AnyDecArgs {
    const uint8_t  *in;
    const uint64_t *in_offsets;
    uint64_t        nblocks;
    uint8_t        *out;
    uint32_t       *out_sizes;
    int32_t        *status;
    uint64_t       *in_used;
    uint32_t       *trees;
    uint64_t        tree_words;
    uint32_t        block_size;
    uint32_t        sb, fb, cb;
};

__global__ void __launch_bounds__(64) k_decode_any(AnyDecArgs a)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.nblocks)
        return;
    const uint64_t o0 = a.in_offsets[blk], o1 = a.in_offsets[blk + 1];
    const any::Params P = any::make_params(a.sb, a.fb, a.cb);
    uint64_t  bi, bo;
    const int st = any::decompress_stream(P, a.trees + blk * a.tree_words, a.in + o0, o1 - o0,
                                          a.out + blk * (uint64_t)a.block_size, a.block_size, bi, bo);
    a.out_sizes[blk] = (uint32_t)bo;
    a.status[blk]    = st;
    if (a.in_used)
        a.in_used[blk] = bi;
}

// ======================================================================================
// host side of the ABI
// ======================================================================================
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Geometry {
    uint64_t nblocks;
    uint64_t slot_bytes; // stride between slots (16-byte multiple, >= cap + 32)
    uint32_t slot_cap;   // usable bytes
    uint32_t rc_n;       // reciprocal table entries
    uint32_t nfreeze;
    bool     u16, fixup;
    bool     any;        // general-parameter path (redux_any.hpp): symbol_bits != 8 or code_bits > 32
    bool     gen;        // ... except 4- and 12-bit symbols with code_bits <= 32: lock-step kernels of redux_gen.hpp
    uint64_t tree_bytes; // any / gen (12-bit symbols): per-block tree in the workspace
    bool     coop;       // small grid: k_coop_model + k_coop_chain (redux_coop.hpp), (low, high) pairs in the workspace
    bool     coop_linear; // ... with fewer than 64 large blocks: linear slots, nblocks + 1 of them (k_coop_chain<.., LINEAR>)
    uint32_t pair_width; // ... in rows of this many lanes
    uint32_t coop_win;   // ... one WINDOW of this many symbols of every block at a time (block_size + 1: whole blocks), coop_nwin of them
    uint32_t coop_nwin;
    // workspace layout (encode)
    uint64_t off_rc, off_sizes, off_mode, off_slots, off_trees, off_table, off_seen, off_cstate, off_pairs, total;
};

static int check_params(const redux_params *p)
{
    if (!p)
        return REDUX_INVALID_INPUT;
    const int st = redux_params_check(p->symbol_bits, p->freq_bits, p->code_bits);
    if (st != REDUX_OK)
        return st;
    if (p->symbol_bits > 16) // a tree of 2^symbol_bits + 2 entries per block
        return REDUX_UNSUPPORTED;
    return REDUX_OK;
}

static bool is_any(const redux_params *p) { return p->symbol_bits != 8 || p->code_bits > 32; }
// Symbol widths 1 .. 12 other than 8 with code_bits <= 32 -- the widths src/model/tests.rs exercises besides 8 (4 and 12)
// and everything between -- have lock-step kernels: the encoders of redux_gen.hpp, the cell decoder of
// redux_decode_cells.hpp.  Their limits, in symbols of a block (a bigger block -- whole-stream mode above all -- is one
// lane's serial chain anyway and runs on the one-lane kernels, which have none):
//   * 4 MiB: the kernels index a reciprocal table by the symbol number and address 64 slots / 64 blocks with 32-bit lane offsets;
//   * symbol_bits >= 9: u16 tree nodes holding lowbit + increments: at most 65535 - 2^(symbol_bits - 1) updates of the model
//     (symbols of a block, or fewer if the model freezes first: 12-bit symbols, 20 frequency bits: blocks of 95,230 bytes);
//   * symbol_bits <= 7: u32 nodes: any block up to the 4 MiB above; a block whose count can pass 2^17 takes the kernels'
//     fix-up instances (quotient fix-ups in scale_div, a 62-bit numerator for the decoder's code value).
static uint64_t gen_updates(const redux_params *p, uint32_t block_size) // increments a tree node of a block can receive
{
    const uint64_t nsym    = (uint64_t)block_size * 8 / p->symbol_bits;
    const uint64_t nfreeze = ((1ull << p->freq_bits) - 1) - ((1ull << p->symbol_bits) + 1);
    return nsym < nfreeze ? nsym : nfreeze;
}
// the cell decoder takes the block: see above
static bool gen_decode_cells(const redux_params *p, uint32_t block_size)
{
    if (p->symbol_bits >= 8)
        return gen_updates(p, block_size) + (1ull << (p->symbol_bits - 1)) <= 65535;
    return true;
}
// ... with the fix-up instance: the count passes 2^17 inside a block.  (By a few symbols only -- a 64 KiB block of 4-bit symbols
// ends at 2^17 + 17 -- is not worth the instance's ~10 %: the plain one stops its lock-step loop there and its per-lane loop,
// which divides exactly, codes the rest.)
static bool gen_needs_fixup(const redux_params *p, uint32_t block_size)
{
    return p->symbol_bits < 8 && (1ull << p->symbol_bits) + 1 + gen_updates(p, block_size) > (1ull << 17) + 64;
}
static bool is_gen(const redux_params *p, uint32_t block_size)
{
    if (p->code_bits > 32 || p->symbol_bits == 8 || p->symbol_bits > 12 || block_size > (1u << 22))
        return false;
    return gen_decode_cells(p, block_size);
}
// 11- and 12-bit symbols: the bottom cells (2^(symbol_bits - 4) of 32 bytes per block: 4 / 8 KiB) live in the workspace, the
// cells above them in LDS: 64 blocks per wave, four waves per CU (redux_decode_cells.hpp).  Measured against keeping everything
// in LDS (which holds 16 blocks of 12-bit symbols per CU, 32 of 11-bit): never slower from 2,048 blocks up, 2 x faster where
// the LDS form needs a second pass.  9- and 10-bit symbols fit LDS with 64 blocks per wave and are faster there
// (profiles/r04_gen_parameters.txt).
static bool gen_decode_in_workspace(const redux_params *p, uint64_t nblocks)
{
    (void)nblocks;
    return p->symbol_bits >= 11;
}
static uint64_t gen_decode_tree_bytes(const redux_params *p) { return (1ull << (p->symbol_bits - 4)) * 32; }
// 8-bit symbols in blocks above 64 KiB (which k_decode_lock's u16 nodes do not hold), in launches too big for one block per
// wave (k_decode_wave): the cell decoder with u32 nodes -- 68 KiB of cells per wave, two waves per CU -- instead of k_decode's
// per-lane control flow.  No block tables (the cell decoder takes blocks in order), blocks of at most 4 MiB (its reciprocal
// table is indexed by the symbol number).
static bool cells8_takes(const redux_params *p, uint32_t block_size, uint64_t nslots, bool table)
{
    // (nslots == 0: "a full grid", redux_decode_kernel_name)
    return p->symbol_bits == 8 && p->code_bits <= 32 && block_size > 65536 && block_size <= (1u << 22) &&
           (nslots == 0 || nslots > kWaveDecMaxBlocks || (nslots > kWaveDecManyBlocks && block_size >= kWaveDecLargeBlock)) && !table;
}
static uint32_t cells8_rc_entries(const redux_params *p, uint32_t block_size)
{
    const uint64_t nfreeze = ((1ull << p->freq_bits) - 1) - 257;
    return (uint32_t)((block_size < nfreeze ? block_size : nfreeze) + 1 + 32);
}
static bool cells8_needs_fixup(const redux_params *p, uint32_t block_size)
{
    return 257ull + cells8_rc_entries(p, block_size) - 33 > (1ull << 17) + 64;
}

static uint64_t slot_cap_for(const redux_params *p, uint32_t block_size)
{
    const uint64_t freq_max = (1ull << p->freq_bits) - 1;
    if (is_any(p)) {
        // General parameters: a symbol of frequency >= 1 out of count <= min(freq_max, K + n)
        // costs at most ceil(log2 count) bits, + 1 for the truncation of codec.rs:59-60, + 1 spare.
        const uint64_t K = (1ull << p->symbol_bits) + 1;
        const uint64_t n = (uint64_t)block_size * 8 / p->symbol_bits + 1; // symbols incl. EOF
        uint32_t       lg = 0;
        while ((1ull << lg) < K + n)
            lg++;
        const uint64_t per = (lg < p->freq_bits ? lg : p->freq_bits) + 2;
        return n * per / 8 + p->code_bits / 8 + 64;
    }
    // Worst case of one block's stream.  While the model never freezes inside a block the
    // adaptive code length is <= 8 bits/symbol + O(256 log N) and the integer truncation of
    // codec.rs:59-60 loses < 1 bit/symbol: 9 bits/symbol.  Once frozen (count == freq_max) a
    // symbol of frequency 1 costs up to freq_bits + 1 bits.
    const uint64_t n        = block_size;
    const bool     freezes  = 257ull + n > freq_max;
    const uint64_t bits     = freezes ? n * (p->freq_bits + 2) : n * 9;
    return bits / 8 + 1024;
}

// static_model: the fixed-table coder (redux_static.hpp).  A symbol of frequency >= 1 out of
// total <= freq_max costs at most freq_bits bits + 1 for the truncation of codec.rs:59-60, + 1
// spare; no reciprocal table, no tree.
static Geometry geometry(const redux_params *p, uint64_t in_len, uint32_t block_size, bool static_model = false, bool allow_coop = true)
{
    Geometry g;
    memset(&g, 0, sizeof g);
    g.nblocks = in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
    const uint64_t cap = static_model ? ((uint64_t)block_size + 1) * (p->freq_bits + 2) / 8 + 1024 : slot_cap_for(p, block_size);
    g.slot_cap   = cap > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)cap;
    // Slot stride: a whole number of 128-byte lines, and an ODD one.  All lanes write their
    // slots at about the same relative offset, so a stride that is a multiple of 2^k lines
    // folds the concurrently written lines onto 1/2^k of the L2 sets and evicts them
    // half-written (measured: 2.6x the stream bytes written to HBM at a stride of 584 lines).
    g.slot_bytes = align_up((uint64_t)g.slot_cap + 32, 128);
    if (((g.slot_bytes / 128) & 1) == 0)
        g.slot_bytes += 128;
    const uint64_t freq_max = (1ull << p->freq_bits) - 1;
    g.nfreeze = (uint32_t)(freq_max - 257);
    const uint64_t maxlen = in_len < block_size ? in_len : block_size;
    g.rc_n  = (uint32_t)((maxlen < g.nfreeze ? maxlen : g.nfreeze) + 1);
    g.u16   = block_size <= 65536;
    g.fixup = (257ull + (uint64_t)(g.rc_n - 1)) >= (1ull << 17);
    g.rc_n += 32; // slack: both coders load their reciprocals a group / a chunk ahead without clamping
    g.any = !static_model && is_any(p);
    if (static_model)
        g.rc_n = 0;
    g.gen = !static_model && is_gen(p, block_size) && 64ull * g.slot_bytes < (1ull << 32); // (64 blocks: implied by is_gen)
    if (g.gen) { // reciprocal table over the symbol count (the trees are in LDS)
        const uint64_t k0      = (1ull << p->symbol_bits) + 1;
        const uint64_t nsym    = maxlen * 8 / p->symbol_bits;
        const uint64_t nfreeze = freq_max - k0;
        g.any     = false;
        g.nfreeze = (uint32_t)(nfreeze < 0xFFFFFFFFull ? nfreeze : 0xFFFFFFFFull);
        g.rc_n    = (uint32_t)((nsym < nfreeze ? nsym : nfreeze) + 1 + 32);
        g.u16     = false;
        g.fixup   = true;
        g.tree_bytes = 0;
    }
    if (g.any) { // no reciprocal table; one tree of 2^symbol_bits + 2 u32 per block
        g.rc_n       = 0;
        g.u16        = false;
        g.fixup      = true;
        g.tree_bytes = align_up(((1ull << p->symbol_bits) + 2) * 4, 256);
    }
    // a grid that leaves most SIMDs idle: the model by 64 lanes per block, the chain by one.  One block of any length --
    // redux_compress, the literal redux::compress -- is such a grid (redux_coop.hpp).
    g.pair_width = g.u16 ? 64u : 1u; // blocks of up to 64 KiB: rows of 64 lanes per symbol; larger ones: block-major (k_coop_model)
    // Blocks of up to 64 KiB: the pairs of whole blocks (8 bytes per input byte).  Larger blocks -- one stream of any length
    // above all -- are coded in windows, one (model, chain) pair of launches per window, so that the pairs area and the
    // reciprocal table hold one window whatever the block length: the largest window whose pairs fit kCoopWindowBytes, at most
    // kCoopWindowMax symbols, the windows together covering block_size + 1 symbols (the last one holds a full block's EOF).
    const uint64_t lanes_total = g.u16 ? (g.nblocks + 63) / 64 * 64 : g.nblocks;
    g.coop_win  = block_size + 1;
    g.coop_nwin = 1;
    if (!g.u16) {
        uint64_t w = kCoopWindowBytes / 2 / (8 * lanes_total); // (two buffers: the model of a window runs next to the chain of the one before)
        w = w > kCoopWindowMax ? kCoopWindowMax : w;
        w = w < 4096 ? 4096 : w;
        if (w < (uint64_t)block_size + 1) {
            g.coop_nwin = (uint32_t)(((uint64_t)block_size + 1 + w - 1) / w);
            g.coop_win  = (uint32_t)((((uint64_t)block_size + 1 + g.coop_nwin - 1) / g.coop_nwin + 31) & ~31ull);
        }
    }
    const uint64_t pair_bytes = (g.u16 ? lanes_total * ((uint64_t)g.coop_win + kCoopSlack) : 2 * lanes_total * coop_block_pitch(g.coop_win)) * 8;
    // fewer than 64 large blocks on the small-grid kernels: linear slots, one per block (a row-major group area is 64 slots
    // big whatever the number of blocks: 230 MiB to code one 3 MiB stream); a lane addresses its slot with 32-bit offsets
    const bool linear = g.nblocks < 64 && !g.u16;
    g.coop = allow_coop && !static_model && !g.any && !g.gen && g.nblocks <= (g.u16 ? kCoopMaxBlocks : kCoopMaxLargeBlocks) && block_size >= kCoopMinBlock &&
             (linear ? g.nblocks : 64ull) * g.slot_bytes < (1ull << 32) && pair_bytes <= (g.u16 ? kCoopMaxPairBytes : kCoopWindowBytes + (64ull << 20));
    if (g.coop) // the reciprocals of one window (+ what the chain wave reads ahead); blocks coded in windows: of two, alternating
        g.rc_n = g.u16 ? g.coop_win + 64 : 2 * ((g.coop_win + 64 + 31) & ~31u);
    g.coop_linear = g.coop && linear;
    g.off_rc    = 0;
    g.off_sizes = align_up(g.off_rc + (uint64_t)g.rc_n * 8, 256);
    g.off_mode  = align_up(g.off_sizes + g.nblocks * 4, 256); // one word: 0 linear slots, != 0 row-major group areas
    g.off_slots = g.off_mode + 256 + kClaimWords * 4; // mode word, then k_encode_pair's role book
    // whole groups of 64 slots (a row-major group area is 64 slots big) + 1 spare slot for the dead lanes of linear mode;
    // giant blocks (one per wave, encode_lanes()) and the small-grid kernels' linear slots: one per block
    const bool     one_each = g.coop_linear || (!g.any && !g.gen && !static_model && !g.coop && 64ull * g.slot_bytes >= (1ull << 32));
    const uint64_t nslots = one_each ? g.nblocks : (g.nblocks + 63) / 64 * 64 + 1;
    g.off_trees = align_up(g.off_slots + nslots * g.slot_bytes + (g.nblocks + 63) / 64 * 128, 256);
    // the checked copy of a `_v_dev` call's block table + the bitmap of block numbers its check uses (redux_table.hpp)
    g.off_table = align_up(g.off_trees + (g.gen ? (g.nblocks + 63) / 64 * 64 : g.nblocks) * g.tree_bytes, 256); // gen: whole waves
    g.off_seen  = g.off_table + align_up(g.nblocks * sizeof(redux_block), 256);
    // small-grid kernels, blocks coded in windows: 8 words of coder state + 256 symbol counts per block, carried between windows
    g.off_cstate = g.off_seen + align_up(table_seen_words(g.nblocks) * 4, 256);
    g.off_pairs  = g.off_cstate + ((g.coop && !g.u16) ? align_up(g.nblocks * (8 + 256) * 4, 256) : 0);
    g.total = g.off_pairs + (g.coop ? pair_bytes : 0);
    return g;
}

// The layout a launch uses in the workspace it was GIVEN.  A workspace sized for a larger input, or for a pipeline of several
// chunks, has no room for the small-grid kernels' pairs area: the launch then runs the full-grid kernels on the layout they
// need (same bytes out, the small-launch speed-up forgone).  Both phases of a call -- the coder and the compaction -- must
// take this decision the same way, so it is made here from (shape, workspace size) alone.
static Geometry geometry_ws(const redux_params *p, uint64_t in_len, uint32_t block_size, uint64_t workspace_bytes)
{
    Geometry g = geometry(p, in_len, block_size);
    if (g.coop && workspace_bytes < g.total)
        g = geometry(p, in_len, block_size, false, false);
    return g;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "redux_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),       \
                    __FILE__, __LINE__);                                                               \
            return REDUX_IO_ERROR;                                                                     \
        }                                                                                              \
    } while (0)

// CUs of HIP's current device (the _dev entry points launch on it and keep no other state): cached per device id
static uint32_t cu_count()
{
    static std::atomic<int> cus[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
        return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return (uint32_t)n;
}

// ---- which kernel a call runs: ONE decision, used by the launch code and reported by
// redux_encode_kernel_name / redux_decode_kernel_name (bench.py's roofline.kernel) ------------
enum class EncKernel { PairCb32, Pair, SingleU16, SingleU16Fixup, SingleU32, Gen, GenPair, Any, CoopCb32, Coop };
enum class DecKernel { LockCb32, Lock, GenericU16, GenericU16Fixup, GenericU32, Cells, CellsFixup, CellsWorkspace, Cells8, Cells8Fixup, Any, Wave, WaveFixup };

// 64 blocks per wave while 64 slots / 64 blocks stay within a 32-bit lane offset; otherwise
// (giant blocks, whole-stream mode) one block per wave.
static uint32_t encode_lanes(const Geometry &g, uint32_t block_size)
{
    return (64ull * g.slot_bytes < (1ull << 32) && 64ull * block_size < (1ull << 32)) ? 64u : 1u;
}

static EncKernel pick_encode_kernel(const Geometry &g, const redux_params *p, bool aligned16, uint32_t block_size)
{
    if (g.gen)
        return p->symbol_bits < 8 ? EncKernel::Gen : EncKernel::GenPair;
    if (g.any)
        return EncKernel::Any;
    bool coop = g.coop;
#ifdef REDUX_AB // A/B timing builds only: REDUX_ENCODE_KERNEL=pair keeps small grids on the pair kernel
    if (getenv("REDUX_ENCODE_KERNEL"))
        coop = false;
#endif
    if (coop)
        return p->code_bits == 32 ? EncKernel::CoopCb32 : EncKernel::Coop;
    bool pair = g.u16 && aligned16 && encode_lanes(g, block_size) == 64;
#ifdef REDUX_AB // A/B timing builds only: REDUX_ENCODE_KERNEL=single pins the one-wave kernel
    const char *force = getenv("REDUX_ENCODE_KERNEL");
    if (force && !strcmp(force, "single"))
        pair = false;
#endif
    // (a u16 tree means blocks of <= 65536 symbols, so count < 2^17: the pair kernel never needs FIXUP)
    if (pair && !g.fixup)
        return p->code_bits == 32 ? EncKernel::PairCb32 : EncKernel::Pair;
    if (g.u16)
        return g.fixup ? EncKernel::SingleU16Fixup : EncKernel::SingleU16;
    return EncKernel::SingleU32;
}

// nslots: blocks (or table entries) of the launch; 0 = unknown (redux_decode_kernel_name: the full-grid choice)
static DecKernel pick_decode_kernel(const Geometry &g, const redux_params *p, uint64_t nslots = 0, uint32_t block_size = 0, bool table = false)
{
    if (g.gen) {
        if (gen_needs_fixup(p, block_size))
            return DecKernel::CellsFixup;
        return gen_decode_in_workspace(p, nslots ? nslots : ~0ull) ? DecKernel::CellsWorkspace : DecKernel::Cells;
    }
    if (g.any)
        return DecKernel::Any;
    if (cells8_takes(p, block_size, nslots, table))
        return cells8_needs_fixup(p, block_size) ? DecKernel::Cells8Fixup : DecKernel::Cells8;
    // blocks the lock-step decoder does not take (u32 counts, count >= 2^17: one block of any length above all,
    // redux_decompress) in a launch that leaves SIMDs idle: one block per wave, the model across the lanes
    // (redux_decode_wave.hpp).  (For u16 blocks it measures 28.9 ms per 64 KiB block against the lock-step decoder's 24.)
    bool wave = nslots != 0 && nslots <= kWaveDecMaxBlocks && !(g.u16 && !g.fixup);
#ifdef REDUX_AB
    if (getenv("REDUX_DECODE_KERNEL"))
        wave = false;
#endif
    if (wave)
        return g.fixup ? DecKernel::WaveFixup : DecKernel::Wave;
    bool lock = g.u16 && !g.fixup;
#ifdef REDUX_AB // A/B timing builds only: REDUX_DECODE_KERNEL=generic pins k_decode
    if (getenv("REDUX_DECODE_KERNEL"))
        lock = false;
#endif
    if (lock)
        return p->code_bits == 32 ? DecKernel::LockCb32 : DecKernel::Lock;
    if (g.u16)
        return g.fixup ? DecKernel::GenericU16Fixup : DecKernel::GenericU16;
    return DecKernel::GenericU32;
}

} // namespace redux

#include "redux_host.hpp"

using namespace redux;

extern "C" {

const char *redux_version(void) { return "redux_hip 0.3.0 gfx950"; }

#ifndef REDUX_SOURCE_HASH
#define REDUX_SOURCE_HASH "unknown"
#endif
const char *redux_source_hash(void) { return REDUX_SOURCE_HASH; }

const char *redux_encode_kernel_name(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return "";
    const Geometry g = geometry(p, in_len, block_size);
    const bool aligned16 = (((uintptr_t)d_in) & 15) == 0 && (block_size & 15) == 0;
    switch (pick_encode_kernel(g, p, aligned16, block_size)) {
    case EncKernel::CoopCb32: return "k_coop_model + k_coop_chain<true> (small grid: model by 64 lanes per block, chain wave + bit-writer wave, code_bits 32)";
    case EncKernel::Coop: return "k_coop_model + k_coop_chain<false> (small grid: model by 64 lanes per block, chain wave + bit-writer wave)";
    case EncKernel::PairCb32: return "k_encode_pair<false, true> (u16 tree, model wave + coder wave, code_bits 32)";
    case EncKernel::Pair: return "k_encode_pair<false, false> (u16 tree, model wave + coder wave)";
    case EncKernel::SingleU16: return "k_encode<true, false> (u16 tree, one wave per 64 blocks)";
    case EncKernel::SingleU16Fixup: return "k_encode<true, true> (u16 tree, one wave per 64 blocks, quotient fix-up)";
    case EncKernel::SingleU32: return "k_encode<false, true> (u32 tree)";
    case EncKernel::Gen: {
        static const char *const names[8] = {"", "k_encode_gen<1>", "k_encode_gen<2>", "k_encode_gen<3>", "k_encode_gen<4>", "k_encode_gen<5>",
                                              "k_encode_gen<6>", "k_encode_gen<7>"};
        return names[p->symbol_bits]; // (lock-step, u32 tree in LDS, one wave per 64 blocks)
    }
    case EncKernel::GenPair: {
        // lock-step, u16 tree in LDS, one workgroup per 64 / 64 / 32 / 16 blocks: three model waves + a coder wave
        static const char *const names[4] = {"k_encode_gen_pair<9>", "k_encode_gen_pair<10>", "k_encode_gen_pair<11>", "k_encode_gen_pair<12>"};
        return names[p->symbol_bits - 9];
    }
    case EncKernel::Any: return "k_encode_any (general parameters, one lane per block)";
    }
    return "";
}

const char *redux_decode_kernel_name(const redux_params *p, const void *d_out, uint32_t block_size)
{
    return redux_decode_kernel_name_n(p, d_out, block_size, 0);
}

const char *redux_decode_kernel_name_n(const redux_params *p, const void *d_out, uint32_t block_size, uint64_t nblocks)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return "";
    (void)d_out; // every decoder takes any alignment (it only picks the store width inside the kernel)
    const Geometry g = geometry(p, block_size, block_size, false, false); // (a decoder: no small-grid encoder, whose windows would size the reciprocal table)
    switch (pick_decode_kernel(g, p, nblocks, block_size)) {
    case DecKernel::LockCb32: return "k_decode_lock<true> (u16 tree, one wave per 64 blocks, code_bits 32)";
    case DecKernel::Lock: return "k_decode_lock<false> (u16 tree, one wave per 64 blocks)";
    case DecKernel::GenericU16: return "k_decode<true, false> (u16 tree, per-lane control flow)";
    case DecKernel::GenericU16Fixup: return "k_decode<true, true> (u16 tree, quotient fix-up)";
    case DecKernel::GenericU32: return "k_decode<false, true> (u32 tree)";
    case DecKernel::CellsFixup: // (the same kernels' instance for counts of 2^17 and more)
    case DecKernel::Cells:
    case DecKernel::CellsWorkspace: {
        // lock-step, the tree as cells of four levels: all of them in LDS (symbol_bits <= 10), or the bottom ones in the workspace
        static const char *const names[13] = {"", "k_decode_cells<1>", "k_decode_cells<2>", "k_decode_cells<3>", "k_decode_cells<4>",
                                               "k_decode_cells<5>", "k_decode_cells<6>", "k_decode_cells<7>", "", "k_decode_cells<9>",
                                               "k_decode_cells<10>", "k_decode_cells<11>", "k_decode_cells<12>"};
        return names[p->symbol_bits];
    }
    case DecKernel::Cells8:
    case DecKernel::Cells8Fixup: return "k_decode_cells<8> (u32 cells, blocks above 64 KiB, one wave per 64 blocks)";
    case DecKernel::Any: return "k_decode_any (general parameters, one lane per block)";
    case DecKernel::Wave:
    case DecKernel::WaveFixup: return "k_decode_wave (one block per wave, cumulative table across the lanes)";
    }
    return "";
}

int redux_params_check(uint32_t symbol, uint32_t frequency, uint32_t code) /* model/mod.rs:64 */
{
    if (symbol < 1 || frequency < symbol + 2 || code < frequency + 2 || 64 < code + frequency)
        return REDUX_INVALID_INPUT;
    return REDUX_OK;
}

int redux_device_supports(const redux_params *p) { return check_params(p); }

uint64_t redux_block_count(uint64_t in_len, uint32_t block_size)
{
    if (block_size == 0)
        return 0;
    return in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
}

uint64_t redux_encode_slot_bytes(const redux_params *p, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    return geometry(p, block_size, block_size).slot_cap;
}

uint64_t redux_encode_bound(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    const Geometry g = geometry(p, in_len, block_size);
    return g.nblocks * (uint64_t)g.slot_cap;
}

uint64_t redux_encode_workspace_bytes(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    return geometry(p, in_len, block_size).total;
}

// d_table != null: the block table of redux_encode_blocks_v_dev (tbl_blocks entries; in_len = bytes of d_in)
static int encode_slots_impl(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                             const redux_block *d_table, uint64_t tbl_blocks, bool tbl_aligned16, void *d_block_status,
                             void *d_workspace, uint64_t workspace_bytes, void *stream, uint64_t nblocks_real = 0)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_workspace || !d_block_status || (in_len && !d_in))
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry_ws(p, d_table ? tbl_blocks * (uint64_t)block_size : in_len, block_size, workspace_bytes);
    if (d_table && (g.gen || g.any || in_len > 0xFFFFFFFFull || tbl_blocks == 0)) // lane offsets into d_in are 32-bit
        return tbl_blocks == 0 ? REDUX_INVALID_INPUT : REDUX_UNSUPPORTED;
    if (workspace_bytes < g.total)
        return REDUX_OUTPUT_TOO_SMALL;
    if (((uintptr_t)d_workspace) & 255)
        return REDUX_INVALID_INPUT;
    hipStream_t s  = (hipStream_t)stream;
    uint8_t    *ws = (uint8_t *)d_workspace;

    if (d_table) { // caller data: the kernels read a checked copy (redux_table.hpp)
        TableCheckArgs ta;
        ta.in         = d_table;
        ta.out        = (redux_block *)(ws + g.off_table);
        ta.nentries   = tbl_blocks;
        ta.nblocks    = nblocks_real;
        ta.bytes      = in_len;
        ta.block_size = block_size;
        ta.aligned16  = tbl_aligned16 ? 1u : 0u;
        ta.seen       = (uint32_t *)(ws + g.off_seen);
        ta.sizes      = (uint32_t *)(ws + g.off_sizes);
        ta.status     = (int32_t *)d_block_status;
        const uint64_t n0 = std::max(nblocks_real, table_seen_words(nblocks_real));
        k_table_prepare<<<(uint32_t)((n0 + 255) / 256), 256, 0, s>>>(ta);
        k_table_check<<<(uint32_t)((tbl_blocks + 255) / 256), 256, 0, s>>>(ta);
        d_table = ta.out;
    }
    HIP_TRY(hipMemsetAsync(ws + g.off_mode, 0, 256 + kClaimWords * 4, s)); // linear slots unless the pair kernel runs (below); empty role book
    if (g.gen) {
        GenEncArgs ga;
        ga.in         = (const uint8_t *)d_in;
        ga.in_len     = in_len;
        ga.nblocks    = g.nblocks;
        ga.slots      = ws + g.off_slots;
        ga.slot_bytes = g.slot_bytes;
        ga.sizes      = (uint32_t *)(ws + g.off_sizes);
        ga.status     = (int32_t *)d_block_status;
        ga.rc         = (const double *)(ws + g.off_rc);
        ga.block_size = block_size;
        ga.slot_cap   = g.slot_cap;
        ga.nfreeze    = g.nfreeze;
        ga.code_bits  = p->code_bits;
        // (64 slots / blocks within a 32-bit lane offset: geometry())
        k_fill_rc_from<<<(g.rc_n + 255) / 256, 256, 0, s>>>((double *)(ws + g.off_rc), g.rc_n, (1u << p->symbol_bits) + 1u);
        const uint32_t grid64 = (uint32_t)((g.nblocks + 63) / 64);
        switch (p->symbol_bits) {
#define REDUX_GEN_ENC(SB) case SB: k_encode_gen<SB><<<grid64, 64, 0, s>>>(ga); break;
#define REDUX_GEN_ENC_PAIR(SB)                                                                                         \
    case SB: k_encode_gen_pair<SB><<<(uint32_t)((g.nblocks + GenTree<SB>::kBlocks - 1) / GenTree<SB>::kBlocks), 256, 0, s>>>(ga); break;
            REDUX_GEN_ENC(1) REDUX_GEN_ENC(2) REDUX_GEN_ENC(3) REDUX_GEN_ENC(4) REDUX_GEN_ENC(5) REDUX_GEN_ENC(6) REDUX_GEN_ENC(7)
            REDUX_GEN_ENC_PAIR(9) REDUX_GEN_ENC_PAIR(10) REDUX_GEN_ENC_PAIR(11) REDUX_GEN_ENC_PAIR(12)
#undef REDUX_GEN_ENC
#undef REDUX_GEN_ENC_PAIR
        default: return REDUX_UNSUPPORTED;
        }
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }
    if (g.any) {
        AnyEncArgs aa;
        aa.in         = (const uint8_t *)d_in;
        aa.in_len     = in_len;
        aa.nblocks    = g.nblocks;
        aa.slots      = ws + g.off_slots;
        aa.slot_bytes = g.slot_bytes;
        aa.sizes      = (uint32_t *)(ws + g.off_sizes);
        aa.status     = (int32_t *)d_block_status;
        aa.trees      = (uint32_t *)(ws + g.off_trees);
        aa.tree_words = g.tree_bytes / 4;
        aa.block_size = block_size;
        aa.slot_cap   = g.slot_cap;
        aa.sb = p->symbol_bits; aa.fb = p->freq_bits; aa.cb = p->code_bits;
        k_encode_any<<<(uint32_t)((g.nblocks + 63) / 64), 64, 0, s>>>(aa);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }

    k_fill_rc<<<(g.rc_n + 255) / 256, 256, 0, s>>>((double *)(ws + g.off_rc), g.rc_n);

    EncArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_len     = in_len;
    a.nblocks    = g.nblocks;
    a.slots      = ws + g.off_slots;
    a.slot_bytes = g.slot_bytes;
    a.sizes      = (uint32_t *)(ws + g.off_sizes);
    a.status     = (int32_t *)d_block_status;
    a.rc         = (const double *)(ws + g.off_rc);
    a.block_size = block_size;
    a.slot_cap   = g.slot_cap;
    a.nfreeze    = g.nfreeze;
    a.code_bits  = p->code_bits;
    a.aligned16  = ((((uintptr_t)d_in) & 15) == 0 && (d_table ? tbl_aligned16 : (block_size & 15) == 0)) ? 1 : 0;
    a.claims     = (uint32_t *)(ws + g.off_mode + 256);
    a.table      = d_table;
    a.pair_width = g.pair_width;
    a.win0 = 0; a.winlen = block_size + 1; a.rc_frozen = 0.0; a.cstate = nullptr; a.cbase = nullptr;
    // 64 blocks per wave while 64 slots / 64 blocks stay within a 32-bit lane offset;
    // otherwise (giant blocks, whole-stream mode) one block per wave.
    a.lanes = encode_lanes(g, block_size);
    const uint32_t grid = (uint32_t)((g.nblocks + a.lanes - 1) / a.lanes);
    const EncKernel which = pick_encode_kernel(g, p, a.aligned16 != 0, block_size);
    // what the pair kernel leaves in the slots (CompactArgs::mode): byte 0x01 -> row-major group
    // areas, 0x02 -> linear slots whose dwords are byte-reversed
    if (which == EncKernel::PairCb32 || which == EncKernel::Pair || which == EncKernel::CoopCb32 || which == EncKernel::Coop)
        HIP_TRY(hipMemsetAsync(ws + g.off_mode, g.coop_linear ? 2 : (1 | 2), 4, s));
    switch (which) {
    case EncKernel::CoopCb32:
    case EncKernel::Coop: {
        uint2         *pairs = (uint2 *)(ws + g.off_pairs);
        const uint32_t cgrid = (uint32_t)((g.nblocks + 63) / 64);
        const bool     cb32  = which == EncKernel::CoopCb32;
        {
            const double r = 1.0 / (double)(257ull + g.nfreeze); // the frozen model's reciprocal, biased as k_fill_rc's
            uint64_t     u;
            memcpy(&u, &r, 8);
            u += 4;
            memcpy(&a.rc_frozen, &u, 8);
        }
        a.winlen = g.coop_win;
        a.cstate = !g.u16 ? (uint32_t *)(ws + g.off_cstate) : nullptr;
        a.cbase  = a.cstate ? a.cstate + g.nblocks * 8 : nullptr;
        if (g.u16) { // whole blocks
            k_coop_model<false><<<(uint32_t)g.nblocks, 64, 0, s>>>(a, pairs);
            if (g.fixup) {
                if (cb32) k_coop_chain<true, true><<<cgrid, 128, 0, s>>>(a, pairs);
                else      k_coop_chain<false, true><<<cgrid, 128, 0, s>>>(a, pairs);
            } else {
                if (cb32) k_coop_chain<true, false><<<cgrid, 128, 0, s>>>(a, pairs);
                else      k_coop_chain<false, false><<<cgrid, 128, 0, s>>>(a, pairs);
            }
            break;
        }
        // Blocks above 64 KiB, window by window.  Window w's chain runs in ONE launch with window w + 1's model (k_coop_step):
        // two pairs buffers and two reciprocal tables, alternating.  The longest block of the launch decides the number of
        // windows: its EOF symbol (symbol number `length`) is the last one coded.
        const uint64_t longest = d_table ? block_size : (in_len < block_size ? in_len : block_size);
        const uint32_t rc_half = g.rc_n / 2;
        const uint64_t pair_half = g.nblocks * coop_block_pitch(g.coop_win);
        double        *rcs[2] = {(double *)(ws + g.off_rc), (double *)(ws + g.off_rc) + rc_half};
        uint2         *prs[2] = {pairs, pairs + pair_half};
        k_coop_model<true><<<(uint32_t)g.nblocks, 64, 0, s>>>(a, prs[0]); // (win0 = 0)
        for (uint32_t w = 0; w < g.coop_nwin && (uint64_t)w * g.coop_win <= longest; w++) {
            EncArgs ac = a, am = a;
            ac.win0 = w * g.coop_win;
            ac.rc   = rcs[w & 1];
            am.win0 = (w + 1) * g.coop_win;
            // (the table of the first window was filled above, as k_fill_rc fills it, in the first half)
            if (w)
                k_fill_rc_from<<<(rc_half + 255) / 256, 256, 0, s>>>(rcs[w & 1], rc_half, 257u + ac.win0);
            const bool more = w + 1 < g.coop_nwin && (uint64_t)am.win0 <= longest; // (a window that only holds EOF symbols has no model)
            const uint2 *pc = prs[w & 1];
            uint2       *pm = prs[(w + 1) & 1];
            const uint32_t sgrid = cgrid + (uint32_t)g.nblocks;
#define REDUX_COOP_LAUNCH(CB, FX, LIN)                                                                                 \
    do {                                                                                                               \
        if (more) k_coop_step<CB, FX, LIN><<<sgrid, 128, 0, s>>>(ac, am, pc, pm, cgrid);                               \
        else      k_coop_chain<CB, FX, LIN, true><<<cgrid, 128, 0, s>>>(ac, pc);                                       \
    } while (0)
            if (g.coop_linear) {
                if (g.fixup) { if (cb32) REDUX_COOP_LAUNCH(true, true, true); else REDUX_COOP_LAUNCH(false, true, true); }
                else         { if (cb32) REDUX_COOP_LAUNCH(true, false, true); else REDUX_COOP_LAUNCH(false, false, true); }
            } else {
                if (g.fixup) { if (cb32) REDUX_COOP_LAUNCH(true, true, false); else REDUX_COOP_LAUNCH(false, true, false); }
                else         { if (cb32) REDUX_COOP_LAUNCH(true, false, false); else REDUX_COOP_LAUNCH(false, false, false); }
            }
#undef REDUX_COOP_LAUNCH
        }
        break;
    }
    case EncKernel::PairCb32: k_encode_pair<false, true><<<grid, 128, 0, s>>>(a); break;
    case EncKernel::Pair: k_encode_pair<false, false><<<grid, 128, 0, s>>>(a); break;
    case EncKernel::SingleU16: k_encode<true, false><<<grid, 64, 0, s>>>(a); break;
    case EncKernel::SingleU16Fixup: k_encode<true, true><<<grid, 64, 0, s>>>(a); break;
    case EncKernel::SingleU32: k_encode<false, true><<<grid, 64, 0, s>>>(a); break;
    case EncKernel::Gen:
    case EncKernel::GenPair:
    case EncKernel::Any: break; // handled above
    }
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_encode_slots_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                           void *d_block_status, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    return encode_slots_impl(p, d_in, in_len, block_size, nullptr, 0, false, d_block_status, d_workspace, workspace_bytes, stream);
}

// scan + gather of the slots a coder kernel left in the workspace laid out by g
static int compact_with(const Geometry &g, void *d_out, uint64_t out_cap, void *d_out_offsets, void *d_block_status,
                        void *d_summary, void *d_workspace, uint64_t workspace_bytes, void *stream,
                        const redux_block *d_table = nullptr, uint64_t nblocks_real = 0)
{
    if (!d_workspace || !d_block_status || !d_out_offsets || !d_out)
        return REDUX_INVALID_INPUT;
    if (workspace_bytes < g.off_pairs) // (the compaction reads nothing behind the slots and trees: a workspace without the pairs area will do)
        return REDUX_OUTPUT_TOO_SMALL;
    hipStream_t s  = (hipStream_t)stream;
    uint8_t    *ws = (uint8_t *)d_workspace;

    ScanArgs sa;
    sa.sizes   = (const uint32_t *)(ws + g.off_sizes);
    sa.status  = (const int32_t *)d_block_status;
    sa.offsets = (uint64_t *)d_out_offsets;
    sa.summary = (int32_t *)d_summary;
    sa.nblocks = d_table ? nblocks_real : g.nblocks; // (g counts slots = table entries; sizes and status are per block)
    if (scan_is_coalesced(sa))
        k_scan_sizes_coalesced<<<1, 1024, 0, s>>>(sa);
    else
        k_scan_sizes<<<1, 1024, 0, s>>>(sa);

    CompactArgs ca;
    ca.slots      = ws + g.off_slots;
    ca.slot_bytes = g.slot_bytes;
    ca.offsets    = (const uint64_t *)d_out_offsets;
    ca.out        = (uint8_t *)d_out;
    ca.out_cap    = out_cap;
    ca.status     = (int32_t *)d_block_status;
    ca.summary    = (int32_t *)d_summary;
    ca.nblocks    = g.nblocks;
    ca.mode       = (const uint32_t *)(ws + g.off_mode);
    ca.cap_rows   = (uint32_t)(g.slot_bytes / 4);
    ca.table      = d_table;
    k_compact<<<(uint32_t)g.nblocks, 256, 0, s>>>(ca);
    if (!g.any && !g.gen && (g.u16 || g.coop)) { // (every launch that may have left row-major group areas: the kernel reads the mode word)
        const uint32_t tiles = (ca.cap_rows + kTileRows - 1) / kTileRows + 1;
        const uint32_t groups = (uint32_t)((g.nblocks + 63) / 64);
        k_compact_rows<<<(groups + 7) / 8 * 8 * tiles, 256, 0, s>>>(ca); // (whole sets of 8 groups: k_compact_rows' XCD-aware mapping)
    }
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_compact_slots_dev(const redux_params *p, uint64_t in_len, uint32_t block_size, void *d_out,
                            uint64_t out_cap, void *d_out_offsets, void *d_block_status, void *d_summary,
                            void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0)
        return REDUX_INVALID_INPUT;
    return compact_with(geometry_ws(p, in_len, block_size, workspace_bytes), d_out, out_cap, d_out_offsets, d_block_status, d_summary,
                        d_workspace, workspace_bytes, stream);
}

int redux_encode_blocks_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                            void *d_out, uint64_t out_cap, void *d_out_offsets, void *d_block_status,
                            void *d_summary, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    int st = redux_encode_slots_dev(p, d_in, in_len, block_size, d_block_status, d_workspace, workspace_bytes, stream);
    if (st != REDUX_OK)
        return st;
    return redux_compact_slots_dev(p, in_len, block_size, d_out, out_cap, d_out_offsets, d_block_status, d_summary,
                                   d_workspace, workspace_bytes, stream);
}

int redux_encode_blocks_v_dev(const redux_params *p, const void *d_in, uint64_t in_bytes, const void *d_table,
                              uint64_t nentries, uint64_t nblocks, uint32_t block_size, uint32_t flags, void *d_out,
                              uint64_t out_cap, void *d_out_offsets, void *d_block_status, void *d_summary, void *d_workspace,
                              uint64_t workspace_bytes, void *stream)
{
    if (!d_table || nblocks == 0 || nentries < nblocks)
        return REDUX_INVALID_INPUT;
    if (nentries > 0xFFFFFFF0ull || nblocks > 0xFFFFFFF0ull || !d_block_status || !d_workspace)
        return REDUX_INVALID_INPUT;
    int st = encode_slots_impl(p, d_in, in_bytes, block_size, (const redux_block *)d_table, nentries,
                               (flags & REDUX_V_ALIGNED16) != 0, d_block_status, d_workspace, workspace_bytes, stream, nblocks);
    if (st != REDUX_OK)
        return st;
    // (from here on the table is its checked copy in the workspace)
    const Geometry g = geometry_ws(p, nentries * (uint64_t)block_size, block_size, workspace_bytes);
    st = compact_with(g, d_out, out_cap, d_out_offsets, d_block_status, d_summary, d_workspace, workspace_bytes, stream,
                      (const redux_block *)((uint8_t *)d_workspace + g.off_table), nblocks);
    if (st != REDUX_OK)
        return st;
    k_table_verdict<<<1, 1, 0, (hipStream_t)stream>>>((const uint32_t *)((uint8_t *)d_workspace + g.off_seen) + table_seen_words(nblocks) - 1,
                                                      (int32_t *)d_summary);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

uint64_t redux_block_count_v(const uint64_t *in_len, uint64_t ninputs, uint32_t block_size)
{
    if (block_size == 0 || (ninputs && !in_len))
        return 0;
    uint64_t n = 0;
    for (uint64_t i = 0; i < ninputs; i++)
        n += redux_block_count(in_len[i], block_size);
    return n;
}

uint64_t redux_block_table_v(const uint64_t *in_off, const uint64_t *in_len, uint64_t ninputs, uint32_t block_size,
                             redux_block *table)
{
    const uint64_t nb = redux_block_count_v(in_len, ninputs, block_size);
    if (nb == 0 || nb > 0xFFFFFFF0ull || !in_off)
        return 0;
    // whole blocks first, in block order; then the short ones (at most one per input), longest first
    std::vector<redux_block> whole, tails;
    uint64_t b = 0;
    for (uint64_t i = 0; i < ninputs; i++) {
        const uint64_t cnt = redux_block_count(in_len[i], block_size);
        for (uint64_t j = 0; j < cnt; j++, b++) {
            const uint64_t o   = j * (uint64_t)block_size;
            const uint64_t rem = in_len[i] - o;
            redux_block    e;
            e.offset = in_off[i] + o;
            e.length = rem < block_size ? (uint32_t)rem : block_size;
            e.index  = (uint32_t)b;
            (e.length == block_size ? whole : tails).push_back(e);
        }
    }
    std::stable_sort(tails.begin(), tails.end(), [](const redux_block &x, const redux_block &y) { return x.length > y.length; });
    // a new wave wherever the length has dropped by an eighth (small blocks: by 512 bytes) since the wave's first entry
    std::vector<redux_block> t(whole);
    const redux_block idle = {0, 0, REDUX_BLOCK_IDLE};
    uint32_t first_len = block_size;
    for (const redux_block &e : tails) {
        const uint32_t slack = first_len / 8 > 512 ? first_len / 8 : 512;
        if (t.size() % 64 == 0)
            first_len = e.length;
        else if (e.length + slack < first_len) {
            while (t.size() % 64)
                t.push_back(idle);
            first_len = e.length;
        }
        t.push_back(e);
    }
    if (table)
        memcpy(table, t.data(), t.size() * sizeof(redux_block));
    return t.size();
}

int redux_encode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len,
                          uint64_t ninputs, uint32_t block_size, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets,
                          int32_t *block_status)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || ninputs == 0 || !in_off || !in_len || !out || !out_offsets)
        return REDUX_INVALID_INPUT;
    if (is_any(p))
        return REDUX_UNSUPPORTED;
    return host::encode_blocks_v(p, in, in_off, in_len, ninputs, block_size, out, out_cap, out_offsets, block_status);
}

int redux_encode_blocks(const redux_params *p, const uint8_t *in, uint64_t in_len, uint32_t block_size,
                        uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, int32_t *block_status)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !out || !out_offsets || (in_len && !in))
        return REDUX_INVALID_INPUT;
    return host::encode_blocks(p, in, in_len, block_size, out, out_cap, out_offsets, block_status); // redux_host.hpp
}

int redux_compress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                   uint64_t *bytes_in, uint64_t *bytes_out) /* src/lib.rs:102-109 */
{
    if (in_len > 0xFFFFFF00ull)
        return REDUX_UNSUPPORTED;
    uint64_t  offs[2] = {0, 0};
    int32_t   st      = 0;
    const int rc = redux_encode_blocks(p, in, in_len, in_len ? (uint32_t)in_len : 1u, out, out_cap, offs, &st);
    if (rc == REDUX_OK) {
        if (bytes_in)
            *bytes_in = in_len;
        if (bytes_out)
            *bytes_out = offs[1];
    }
    return rc;
}

// entries of the decoders' reciprocal table: what the block capacity (or the freeze point) asks for, but at most
// kDecRcWindow + slack -- a decoder of longer blocks computes the rest itself (rc_lookup, redux_decode.hpp)
static uint32_t dec_rc_entries(const Geometry &g)
{
    return (g.gen || g.rc_n <= kDecRcWindow + 32) ? g.rc_n : kDecRcWindow + 32;
}

uint64_t redux_decode_workspace_bytes(const redux_params *p, uint64_t nblocks, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    const Geometry g = geometry(p, block_size, block_size, false, false); // (a decoder: no small-grid encoder, whose windows would size the reciprocal table)
    if (g.gen) // (11- and 12-bit symbols: the bottom cells of the decoder's tree live in the workspace: gen_decode_in_workspace)
        return align_up((uint64_t)g.rc_n * 8, 256) +
               (gen_decode_cells(p, block_size) && gen_decode_in_workspace(p, nblocks) ? (nblocks + 63) / 64 * 64 * gen_decode_tree_bytes(p) : 0);
    if (g.any)
        return (nblocks ? nblocks : 1) * g.tree_bytes;
    // the reciprocal table, then room for the checked copy of a block table and its bitmap (redux_table.hpp)
    const uint64_t rc_n = cells8_takes(p, block_size, nblocks ? nblocks : 1, false) ? std::max<uint64_t>(cells8_rc_entries(p, block_size), dec_rc_entries(g))
                                                                                  : dec_rc_entries(g);
    return align_up(rc_n * 8, 256) + align_up(nblocks * sizeof(redux_block), 256) + align_up(table_seen_words(nblocks) * 4, 256);
}

// d_in_used (optional, u64[nblocks]): bytes of each stream the reader fetched; only
// redux_decompress asks for it (the (u64, u64) of src/lib.rs:119)
static int decode_blocks_dev_impl(const redux_params *p, const void *d_in, const void *d_in_offsets, uint64_t nblocks,
                                  uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_sizes,
                                  void *d_block_status, void *d_summary, void *d_workspace,
                                  uint64_t workspace_bytes, void *stream, void *d_in_used,
                                  const redux_block *d_table = nullptr, bool tbl_aligned16 = false, uint64_t nblocks_real = 0)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_in_offsets || !d_out_sizes || !d_block_status || !d_workspace)
        return REDUX_INVALID_INPUT;
    if (nblocks == 0)
        return REDUX_OK;
    if (!d_table && out_cap < nblocks * (uint64_t)block_size)
        return REDUX_OUTPUT_TOO_SMALL;
    const Geometry g = geometry(p, block_size, block_size, false, false); // (a decoder: no small-grid encoder, whose windows would size the reciprocal table)
    if (workspace_bytes < redux_decode_workspace_bytes(p, nblocks, block_size))
        return REDUX_OUTPUT_TOO_SMALL;
    if (d_table && (g.gen || g.any))
        return REDUX_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (g.gen) {
        GenDecArgs ga;
        ga.in         = (const uint8_t *)d_in;
        ga.in_offsets = (const uint64_t *)d_in_offsets;
        ga.nblocks    = nblocks;
        ga.out        = (uint8_t *)d_out;
        ga.out_sizes  = (uint32_t *)d_out_sizes;
        ga.status     = (int32_t *)d_block_status;
        ga.rc         = (const double *)d_workspace;
        ga.trees      = (uint32_t *)((uint8_t *)d_workspace + align_up((uint64_t)g.rc_n * 8, 256));
        ga.in_used    = (uint64_t *)d_in_used;
        ga.block_size = block_size;
        ga.nfreeze    = g.nfreeze;
        ga.code_bits  = p->code_bits;
        k_fill_rc_from<<<(g.rc_n + 255) / 256, 256, 0, s>>>((double *)d_workspace, g.rc_n, (1u << p->symbol_bits) + 1u);
        const uint32_t grid64 = (uint32_t)((nblocks + 63) / 64);
        switch (pick_decode_kernel(g, p, nblocks, block_size)) {
        case DecKernel::CellsFixup:
            switch (p->symbol_bits) {
#define REDUX_GEN_DEC(SB) case SB: k_decode_cells<SB, 64, false, true><<<grid64, 64, 0, s>>>(ga); break;
                REDUX_GEN_DEC(1) REDUX_GEN_DEC(2) REDUX_GEN_DEC(3) REDUX_GEN_DEC(4) REDUX_GEN_DEC(5) REDUX_GEN_DEC(6) REDUX_GEN_DEC(7)
#undef REDUX_GEN_DEC
            default: return REDUX_UNSUPPORTED;
            }
            break;
        case DecKernel::CellsWorkspace: {
            // every tree starts at all-ones frequencies: a node = its lowbit
            const uint64_t npieces = (uint64_t)grid64 * 64 * gen_decode_tree_bytes(p) / 16;
            k_fill_cells16<<<(uint32_t)((npieces + 255) / 256), 256, 0, s>>>((cl_u32x4 *)ga.trees, npieces);
            if (p->symbol_bits == 11)
                k_decode_cells<11, 64, true><<<grid64, 64, 0, s>>>(ga);
            else
                k_decode_cells<12, 64, true><<<grid64, 64, 0, s>>>(ga);
            break;
        }
        case DecKernel::Cells:
            switch (p->symbol_bits) {
#define REDUX_GEN_DEC(SB, LANES) case SB: k_decode_cells<SB, LANES, false><<<(uint32_t)((nblocks + LANES - 1) / LANES), LANES, 0, s>>>(ga); break;
                REDUX_GEN_DEC(1, 64) REDUX_GEN_DEC(2, 64) REDUX_GEN_DEC(3, 64) REDUX_GEN_DEC(4, 64) REDUX_GEN_DEC(5, 64) REDUX_GEN_DEC(6, 64)
                REDUX_GEN_DEC(7, 64) REDUX_GEN_DEC(9, 64) REDUX_GEN_DEC(10, 64)
#undef REDUX_GEN_DEC
            default: return REDUX_UNSUPPORTED;
            }
            break;
        default: return REDUX_UNSUPPORTED;
        }
        if (d_summary)
            k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }
    if (g.any) {
        AnyDecArgs aa;
        aa.in         = (const uint8_t *)d_in;
        aa.in_offsets = (const uint64_t *)d_in_offsets;
        aa.nblocks    = nblocks;
        aa.out        = (uint8_t *)d_out;
        aa.out_sizes  = (uint32_t *)d_out_sizes;
        aa.status     = (int32_t *)d_block_status;
        aa.in_used    = (uint64_t *)d_in_used;
        aa.trees      = (uint32_t *)d_workspace;
        aa.tree_words = g.tree_bytes / 4;
        aa.block_size = block_size;
        aa.sb = p->symbol_bits; aa.fb = p->freq_bits; aa.cb = p->code_bits;
        k_decode_any<<<(uint32_t)((nblocks + 63) / 64), 64, 0, s>>>(aa);
        if (d_summary)
            k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }
    if (cells8_takes(p, block_size, nblocks, d_table != nullptr)) {
        GenDecArgs ga;
        ga.in         = (const uint8_t *)d_in;
        ga.in_offsets = (const uint64_t *)d_in_offsets;
        ga.nblocks    = nblocks;
        ga.out        = (uint8_t *)d_out;
        ga.out_sizes  = (uint32_t *)d_out_sizes;
        ga.status     = (int32_t *)d_block_status;
        ga.rc         = (const double *)d_workspace;
        ga.trees      = nullptr;
        ga.in_used    = (uint64_t *)d_in_used;
        ga.block_size = block_size;
        ga.nfreeze    = g.nfreeze;
        ga.code_bits  = p->code_bits;
        const uint32_t rc8 = cells8_rc_entries(p, block_size);
        k_fill_rc<<<(rc8 + 255) / 256, 256, 0, s>>>((double *)d_workspace, rc8);
        const uint32_t grid64 = (uint32_t)((nblocks + 63) / 64);
        if (cells8_needs_fixup(p, block_size))
            k_decode_cells<8, 64, false, true><<<grid64, 64, 0, s>>>(ga);
        else
            k_decode_cells<8, 64, false, false><<<grid64, 64, 0, s>>>(ga);
        if (d_summary)
            k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }
    const uint32_t rc_n = dec_rc_entries(g);
    k_fill_rc<<<(rc_n + 255) / 256, 256, 0, s>>>((double *)d_workspace, rc_n);
    const uint32_t *table_failed = nullptr;
    if (d_table) { // caller data: the kernels read a checked copy (redux_table.hpp)
        uint8_t *wt = (uint8_t *)d_workspace + align_up((uint64_t)rc_n * 8, 256);
        TableCheckArgs ta;
        ta.in         = d_table;
        ta.out        = (redux_block *)wt;
        ta.nentries   = nblocks;
        ta.nblocks    = nblocks_real;
        ta.bytes      = out_cap;
        ta.block_size = block_size;
        ta.aligned16  = tbl_aligned16 ? 1u : 0u;
        ta.seen       = (uint32_t *)(wt + align_up(nblocks * sizeof(redux_block), 256));
        ta.sizes      = (uint32_t *)d_out_sizes;
        ta.status     = (int32_t *)d_block_status;
        const uint64_t n0 = std::max(nblocks_real, table_seen_words(nblocks_real));
        k_table_prepare<<<(uint32_t)((n0 + 255) / 256), 256, 0, s>>>(ta);
        k_table_check<<<(uint32_t)((nblocks + 255) / 256), 256, 0, s>>>(ta);
        d_table      = ta.out;
        table_failed = ta.seen + table_seen_words(nblocks_real) - 1;
    }
    DecArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_offsets = (const uint64_t *)d_in_offsets;
    a.nblocks    = nblocks;
    a.out        = (uint8_t *)d_out;
    a.out_sizes  = (uint32_t *)d_out_sizes;
    a.status     = (int32_t *)d_block_status;
    a.rc         = (const double *)d_workspace;
    a.block_size = block_size;
    a.nfreeze    = g.nfreeze;
    a.code_bits  = p->code_bits;
    a.aligned4   = ((((uintptr_t)d_out) & 3) == 0 && (block_size & 3) == 0) ? 1 : 0;
    if (a.aligned4 && (((uintptr_t)d_out) & 15) == 0 && (block_size & 15) == 0)
        a.aligned4 = 2; // 16-byte aligned blocks: the lock-step decoder stages four dwords per store
    if (d_table) // where a block starts is the table's business: all of them 16-byte aligned, or nothing is assumed
        a.aligned4 = (tbl_aligned16 && (((uintptr_t)d_out) & 15) == 0) ? 2 : 0;
    a.in_used    = (uint64_t *)d_in_used;
    a.table      = d_table;
    a.rc_n       = rc_n - 32; // (the last 32 entries are slack for the lock-step decoder's look-ahead)
    const uint32_t grid = (uint32_t)((nblocks + 63) / 64);
    switch (pick_decode_kernel(g, p, nblocks)) {
    case DecKernel::Wave: k_decode_wave<false><<<(uint32_t)nblocks, 64, 0, s>>>(a); break;
    case DecKernel::WaveFixup: k_decode_wave<true><<<(uint32_t)nblocks, 64, 0, s>>>(a); break;
    case DecKernel::LockCb32: k_decode_lock<true><<<grid, 64, 0, s>>>(a); break;
    case DecKernel::Lock: k_decode_lock<false><<<grid, 64, 0, s>>>(a); break;
    case DecKernel::GenericU16: k_decode<true, false><<<grid, 64, 0, s>>>(a); break;
    case DecKernel::GenericU16Fixup: k_decode<true, true><<<grid, 64, 0, s>>>(a); break;
    case DecKernel::GenericU32: k_decode<false, true><<<grid, 64, 0, s>>>(a); break;
    case DecKernel::Cells:
    case DecKernel::CellsFixup:
    case DecKernel::CellsWorkspace:
    case DecKernel::Cells8:
    case DecKernel::Cells8Fixup:
    case DecKernel::Any: break; // handled above
    }
    if (d_summary) // (with a block table nblocks counts its entries: statuses are per block)
        k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, d_table ? nblocks_real : nblocks, (int32_t *)d_summary);
    if (table_failed)
        k_table_verdict<<<1, 1, 0, s>>>(table_failed, (int32_t *)d_summary);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

// Diagnostic (tests only): where the encoder's per-CU role book sits in the workspace.
int redux_debug_role_book(const redux_params *p, uint64_t in_len, uint32_t block_size, uint64_t *offset, uint64_t *bytes)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !offset || !bytes)
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry(p, in_len, block_size);
    *offset          = g.off_mode + 256;
    *bytes           = kClaimWords * 4;
    return REDUX_OK;
}

// Diagnostic (tests only): max |v_rcp_f64(x) * x - 1| over the integers lo..hi, on the device.
int redux_debug_rcp_check(uint64_t lo, uint64_t hi, double *max_err)
{
    unsigned long long *d = nullptr, h = 0;
    HIP_TRY(hipMalloc(&d, 8));
    HIP_TRY(hipMemset(d, 0, 8));
    k_rcp_check<<<4096, 256>>>(lo, hi, d);
    HIP_TRY(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    memcpy(max_err, &h, 8);
    return REDUX_OK;
}

int redux_decode_blocks_dev(const redux_params *p, const void *d_in, const void *d_in_offsets, uint64_t nblocks,
                            uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_sizes,
                            void *d_block_status, void *d_summary, void *d_workspace, uint64_t workspace_bytes,
                            void *stream)
{
    return decode_blocks_dev_impl(p, d_in, d_in_offsets, nblocks, block_size, d_out, out_cap, d_out_sizes,
                                  d_block_status, d_summary, d_workspace, workspace_bytes, stream, nullptr);
}

int redux_decode_blocks_v_dev(const redux_params *p, const void *d_in, const void *d_in_offsets, const void *d_table,
                              uint64_t nentries, uint64_t nblocks, uint32_t block_size, uint32_t flags, void *d_out,
                              uint64_t out_bytes, void *d_out_sizes, void *d_block_status, void *d_summary, void *d_workspace,
                              uint64_t workspace_bytes, void *stream)
{
    if (!d_table || nblocks == 0 || nentries < nblocks || !d_out || nentries > 0xFFFFFFF0ull)
        return REDUX_INVALID_INPUT;
    return decode_blocks_dev_impl(p, d_in, d_in_offsets, nentries, block_size, d_out, out_bytes, d_out_sizes, d_block_status,
                                  d_summary, d_workspace, workspace_bytes, stream, nullptr, (const redux_block *)d_table,
                                  (flags & REDUX_V_ALIGNED16) != 0, nblocks);
}

int redux_decode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint8_t *out,
                          const uint64_t *out_off, const uint64_t *out_len, uint64_t ninputs, uint32_t block_size,
                          uint32_t *out_sizes, int32_t *block_status)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || ninputs == 0 || !in_offsets || !out_off || !out_len || !out_sizes)
        return REDUX_INVALID_INPUT;
    if (is_any(p))
        return REDUX_UNSUPPORTED;
    return host::decode_blocks_v(p, in, in_offsets, out, out_off, out_len, ninputs, block_size, out_sizes, block_status,
                                 decode_blocks_dev_impl);
}

static int decode_blocks_host(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                              uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes,
                              int32_t *block_status, uint64_t *in_used)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !in_offsets || !out_sizes || (nblocks && !out))
        return REDUX_INVALID_INPUT;
    if (nblocks == 0)
        return REDUX_OK;
    if (out_cap < nblocks * (uint64_t)block_size)
        return REDUX_OUTPUT_TOO_SMALL;
    if (in_offsets[nblocks] && !in)
        return REDUX_INVALID_INPUT;
    return host::decode_blocks(p, in, in_offsets, nblocks, block_size, out, out_cap, out_sizes, block_status, in_used,
                               decode_blocks_dev_impl); // redux_host.hpp
}

int redux_decode_blocks(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                        uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes,
                        int32_t *block_status)
{
    return decode_blocks_host(p, in, in_offsets, nblocks, block_size, out, out_cap, out_sizes, block_status, nullptr);
}

int redux_decompress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                     uint64_t *bytes_in, uint64_t *bytes_out) /* src/lib.rs:113-120 */
{
    if (out_cap == 0 || out_cap > 0xFFFFFF00ull)
        out_cap = out_cap ? 0xFFFFFF00ull : 1;
    uint64_t  offs[2] = {0, in_len};
    uint32_t  sz      = 0;
    int32_t   st      = 0;
    uint64_t  used    = 0;
    const int rc = decode_blocks_host(p, in, offs, 1, (uint32_t)out_cap, out, out_cap, &sz, &st, &used);
    if (rc == REDUX_OK) {
        if (bytes_out)
            *bytes_out = sz;
        // input.get_count() (lib.rs:119): the bytes the BitReader fetched -- the decoder reads
        // exactly the bits the encoder wrote, so trailing bytes after the stream are not counted
        if (bytes_in)
            *bytes_in = used;
    }
    return rc;
}

// ---- static-table model (redux_static.hpp) ----------------------------
static int static_check(const redux_params *p, const uint32_t *cum)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (is_any(p)) // 8-bit symbols, code_bits <= 32: the widths the fast coder core covers
        return REDUX_UNSUPPORTED;
    if (!cum || cum[0] != 0)
        return REDUX_INVALID_INPUT;
    for (uint32_t i = 0; i + 1 < kStaticEntries; i++)
        if (cum[i + 1] <= cum[i]) // every symbol, EOF included, must be codable
            return REDUX_INVALID_INPUT;
    if ((uint64_t)cum[kStaticEntries - 1] > (1ull << p->freq_bits) - 1) // total <= freq_max (model/mod.rs)
        return REDUX_INVALID_INPUT;
    return REDUX_OK;
}

static double static_rc(uint32_t total)
{
    double  r = 1.0 / (double)total;
    int64_t b;
    memcpy(&b, &r, 8);
    b += 4; // as k_fill_rc: never below the true quotient (scale_div)
    memcpy(&r, &b, 8);
    return r;
}

int redux_static_table_check(const redux_params *p, const uint32_t *cum) { return static_check(p, cum); }

uint64_t redux_static_encode_bound(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || is_any(p) || block_size == 0)
        return 0;
    const Geometry g = geometry(p, in_len, block_size, true);
    return g.nblocks * (uint64_t)g.slot_cap;
}

uint64_t redux_static_encode_workspace_bytes(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || is_any(p) || block_size == 0)
        return 0;
    return geometry(p, in_len, block_size, true).total;
}

int redux_static_encode_blocks_dev(const redux_params *p, const uint32_t *cum, const void *d_in, uint64_t in_len,
                                   uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_offsets,
                                   void *d_block_status, void *d_summary, void *d_workspace,
                                   uint64_t workspace_bytes, void *stream)
{
    int st = static_check(p, cum);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || (!d_in && in_len) || !d_block_status || !d_workspace)
        return REDUX_INVALID_INPUT;
    if (((uintptr_t)d_workspace) & 255)
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry(p, in_len, block_size, true);
    if (workspace_bytes < g.total)
        return REDUX_OUTPUT_TOO_SMALL;
    if (64ull * g.slot_bytes >= (1ull << 32) || 64ull * block_size >= (1ull << 32)) // 64 slots / blocks within a 32-bit lane offset
        return REDUX_UNSUPPORTED;
    hipStream_t s  = (hipStream_t)stream;
    uint8_t    *ws = (uint8_t *)d_workspace;
    HIP_TRY(hipMemsetAsync(ws + g.off_mode, 0, 256, s)); // linear slots, stream byte order
    StaticEncArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_len     = in_len;
    a.nblocks    = g.nblocks;
    a.slots      = ws + g.off_slots;
    a.slot_bytes = g.slot_bytes;
    a.sizes      = (uint32_t *)(ws + g.off_sizes);
    a.status     = (int32_t *)d_block_status;
    a.rc         = static_rc(cum[kStaticEntries - 1]);
    a.block_size = block_size;
    a.slot_cap   = g.slot_cap;
    a.code_bits  = p->code_bits;
    a.aligned16  = ((((uintptr_t)d_in) & 15) == 0 && (block_size & 15) == 0) ? 1 : 0;
    memcpy(a.tab.cum, cum, sizeof a.tab.cum);
    const uint32_t grid = (uint32_t)((g.nblocks + 63) / 64);
    const bool     solo = grid <= 4u * cu_count(); // as k_decode_static_lock: one wave per SIMD, not two on some
    if (cum[kStaticEntries - 1] >= (1u << 17))
        k_encode_static<true, false><<<grid, 64, 0, s>>>(a);
    else if (p->code_bits == 32 && solo)
        k_encode_static<false, true, true><<<grid, 64, 0, s>>>(a);
    else if (p->code_bits == 32)
        k_encode_static<false, true><<<grid, 64, 0, s>>>(a);
    else
        k_encode_static<false, false><<<grid, 64, 0, s>>>(a);
    HIP_TRY(hipGetLastError());
    return compact_with(g, d_out, out_cap, d_out_offsets, d_block_status, d_summary, d_workspace, workspace_bytes, stream);
}

int redux_static_decode_blocks_dev(const redux_params *p, const uint32_t *cum, const void *d_in,
                                   const void *d_in_offsets, uint64_t nblocks, uint32_t block_size, void *d_out,
                                   uint64_t out_cap, void *d_out_sizes, void *d_block_status, void *d_summary,
                                   void *stream)
{
    int st = static_check(p, cum);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_in_offsets || !d_out_sizes || !d_block_status)
        return REDUX_INVALID_INPUT;
    if (nblocks == 0)
        return REDUX_OK;
    if (out_cap < nblocks * (uint64_t)block_size)
        return REDUX_OUTPUT_TOO_SMALL;
    hipStream_t   s = (hipStream_t)stream;
    StaticDecArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_offsets = (const uint64_t *)d_in_offsets;
    a.nblocks    = nblocks;
    a.out        = (uint8_t *)d_out;
    a.out_sizes  = (uint32_t *)d_out_sizes;
    a.status     = (int32_t *)d_block_status;
    a.rc         = static_rc(cum[kStaticEntries - 1]);
    a.block_size = block_size;
    a.code_bits  = p->code_bits;
    a.aligned4   = ((((uintptr_t)d_out) & 3) == 0 && (block_size & 3) == 0) ? 1 : 0;
    memcpy(a.tab.cum, cum, sizeof a.tab.cum);
    const uint32_t grid = (uint32_t)((nblocks + 63) / 64);
    if (cum[kStaticEntries - 1] >= (1u << 17))
        k_decode_static<true><<<grid, 64, 0, s>>>(a);
    else {
        StaticLockArgs la;
        memset(&la, 0, sizeof la);
        la.d.in         = a.in;
        la.d.in_offsets = a.in_offsets;
        la.d.nblocks    = nblocks;
        la.d.out        = a.out;
        la.d.out_sizes  = a.out_sizes;
        la.d.status     = a.status;
        la.d.rc         = nullptr;
        la.d.block_size = block_size;
        la.d.nfreeze    = 0xFFFFFFFFu;
        la.d.code_bits  = p->code_bits;
        la.d.aligned4   = a.aligned4;
        if (a.aligned4 && (((uintptr_t)d_out) & 15) == 0 && (block_size & 15) == 0)
            la.d.aligned4 = 2;
        la.d.in_used    = nullptr;
        la.d.table      = nullptr;
        la.rc           = a.rc;
        la.tab          = a.tab;
        const bool solo = grid <= 4u * cu_count(); // at most one wave per SIMD: keep the dispatcher from doubling them up
        if (cum[kStaticEntries - 1] <= 65536u) { // get_symbol by direct lookup
            if (solo) {
                if (p->code_bits == 32)
                    k_decode_static_lut<true, 4><<<(grid + 3) / 4, 256, 0, s>>>(la);
                else
                    k_decode_static_lut<false, 4><<<(grid + 3) / 4, 256, 0, s>>>(la);
            } else {
                if (p->code_bits == 32)
                    k_decode_static_lut<true, 8><<<(grid + 7) / 8, 512, 0, s>>>(la);
                else
                    k_decode_static_lut<false, 8><<<(grid + 7) / 8, 512, 0, s>>>(la);
            }
        } else if (p->code_bits == 32) {
            if (solo)
                k_decode_static_lock<true, true><<<grid, 64, 0, s>>>(la);
            else
                k_decode_static_lock<true, false><<<grid, 64, 0, s>>>(la);
        } else {
            if (solo)
                k_decode_static_lock<false, true><<<grid, 64, 0, s>>>(la);
            else
                k_decode_static_lock<false, false><<<grid, 64, 0, s>>>(la);
        }
    }
    if (d_summary)
        k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_host_release(void) { return host::ctx_release_all(); }

int redux_host_set_devices(const int32_t *device_ids, uint32_t n) { return host::set_devices(device_ids, n); }

int redux_host_set_chunk_bytes(uint64_t min_bytes, uint64_t max_bytes)
{
    if (max_bytes && max_bytes < min_bytes)
        return REDUX_INVALID_INPUT;
    host::g_chunk_min.store(min_bytes);
    host::g_chunk_max.store(max_bytes);
    return REDUX_OK;
}

int redux_host_chunk_plan(uint64_t nblocks, uint32_t block_size, uint32_t ncontexts, int decode, uint64_t *chunk_blocks,
                          uint64_t *nchunks)
{
    if (nblocks == 0 || block_size == 0 || ncontexts == 0 || ncontexts > 16 || !chunk_blocks || !nchunks)
        return REDUX_INVALID_INPUT;
    *chunk_blocks = host::chunk_blocks_for(nblocks, block_size, decode ? host::kDecChunkMax : host::kEncChunkMax, ncontexts);
    *nchunks      = (nblocks + *chunk_blocks - 1) / *chunk_blocks;
    return REDUX_OK;
}

uint64_t redux_host_allocations(void)
{
    uint64_t n = 0;
    for (host::Ctx &c : host::g_ctx) {
        std::lock_guard<std::mutex> l(c.mu);
        n += c.allocs;
    }
    return n;
}

uint64_t redux_host_resident_bytes(void)
{
    uint64_t n = 0;
    for (host::Ctx &c : host::g_ctx) {
        std::lock_guard<std::mutex> l(c.mu);
        if (!c.ready)
            continue;
        for (host::Slot &s : c.slot)
            for (const host::Buf *b : {&s.d_in, &s.d_ws, &s.d_out, &s.d_off, &s.d_sz, &s.d_st, &s.d_sum, &s.d_used, &s.d_tab})
                n += b->cap;
    }
    return n;
}

/* Diagnostic: timeline of the last host-pointer call on the current device (redux_host.hpp, Ctx::trace). */
uint64_t redux_host_trace(double *out, uint64_t cap)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16)
        return 0;
    host::Ctx &c = host::g_ctx[dev];
    std::lock_guard<std::mutex> l(c.mu);
    const uint64_t n = c.trace.size() < cap ? c.trace.size() : cap;
    for (uint64_t i = 0; i < n; i++)
        out[i] = c.trace[i];
    return c.trace.size();
}

int redux_gen_iid_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream)
{
    if (len == 0)
        return REDUX_OK;
    k_gen_iid<<<2048, 256, 0, (hipStream_t)stream>>>((uint8_t *)d_out, len, first_byte, seed);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_gen_zipf_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream)
{
    if (len == 0)
        return REDUX_OK;
    k_gen_zipf<<<2048, 256, 0, (hipStream_t)stream>>>((uint8_t *)d_out, len, first_byte, seed);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

const uint32_t *redux_zipf_thresholds(void) { return h_zipf; }

} // extern "C"
