// redux_encode.hpp -- encode kernels of the MI355X block coder (gfx950 only).
//
//   k_fill_rc            per-step reciprocal table 1/(257+i), biased up 4 ulp
//   k_encode<U16,FIXUP>  one wave per 64 blocks (general form: u32 trees, unaligned input, giant blocks)
//   k_encode_pair<..>    default: a model wave and a coder wave per 64 blocks, LDS ring between them
//
// Included by redux_hip.hip (one translation unit); the host side launches these from
// redux_encode_slots_dev.
#pragma once

#include "redux_coder.hpp"

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

// ======================================================================================
// reciprocal table
// ======================================================================================
__global__ void k_fill_rc(double *rc, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double r = 1.0 / (double)(257u + i); // correctly rounded IEEE division
        rc[i] = __longlong_as_double(__double_as_longlong(r) + 4);
    }
}

// ======================================================================================
// encode
// ======================================================================================
struct EncArgs {
    const uint8_t *in;
    uint64_t       in_len;
    uint64_t       nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    const double  *rc;
    uint32_t       block_size;
    uint32_t       slot_cap;  // usable bytes of a slot
    uint32_t       nfreeze;   // freq_max - 257: number of updates before the freeze
    uint32_t       code_bits;
    uint32_t       aligned16; // in and block_size are 16-byte multiples
    uint32_t       lanes;     // live lanes per wave: 64, or 1 when 64 slots overflow 32-bit offsets
    uint32_t      *claims;    // kClaimWords words, zero at launch: k_encode_pair's per-CU role book
    // Block table (redux_encode_blocks_v_dev; null: block b = in[b * block_size ...)).  Entry j is the block that slot j
    // codes: its first byte is in + offset (offset < 2^32), it is `length` <= block_size bytes long, and its size and
    // status belong to entry `index` of the output tables.
    const redux_block *table;
    uint32_t       pair_width; // small-grid kernels (redux_coop.hpp): lanes per row of the (low, high) pairs in the workspace
    // ... which code a launch's blocks in WINDOWS of winlen symbols, one pair of kernel launches per window (the pairs area
    // holds one window): symbols [win0, win0 + winlen) of every block, the EOF symbol of a block that ends inside them
    // included.  rc[i] is then the reciprocal of count 257 + win0 + i, rc_frozen the frozen model's; cstate / cbase carry
    // a block's coder state (8 words) and symbol counts (256 words) from window to window (null: a single window).
    uint32_t       win0, winlen;
    double         rc_frozen;
    uint32_t      *cstate, *cbase;
};

// A table entry with this index is an idle lane: redux_block_table_v pads the table with them so that blocks of very
// different lengths do not share a wave (a wave runs its fast path for as long as its shortest block lasts).  Its
// offset still names readable input bytes (the lane runs the instruction stream on them and stores into its own slot).
constexpr uint32_t kIdleEntry = 0xFFFFFFFFu;

// what a lane codes: source offset from the wave's base, length, and where its results go
struct EncLane {
    uint32_t soff, len;
    uint64_t ob;
};
__device__ __forceinline__ EncLane enc_lane(const EncArgs &a, uint64_t blk0, uint64_t blk, uint32_t lane, bool live, const uint8_t *&wsrc)
{
    EncLane L;
    if (a.table) { // (wave-uniform branch)
        const redux_block e = a.table[live ? blk : blk0]; // dead lanes of the last wave re-read block blk0's bytes
        wsrc   = a.in;
        L.soff = (uint32_t)e.offset;
        L.len  = live ? e.length : 0u;
        L.ob   = e.index; // (an idle entry, index kIdleEntry: never used -- the caller's `live` is false for it)
    } else {
        wsrc   = a.in + blk0 * a.block_size;
        L.soff = live ? lane * a.block_size : 0u;
        L.len  = 0;
        if (live) {
            const uint64_t rem = a.in_len - blk * a.block_size;
            L.len              = rem < a.block_size ? (uint32_t)rem : a.block_size;
        }
        L.ob = blk;
    }
    return L;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// 16 consecutive symbols of every lane's block, all lanes alive, no EOF: the hot loop body.
// UPD: the model is still adaptive for the whole chunk (updates so far = q = p + i);
// otherwise it is frozen (adaptive_tree.rs:84) and nup = nfreeze for every symbol.
// A pending run longer than 32 bits can add any number of bytes, so the chunk's byte budget
// is guarded by the caller only for the common path (4 bytes per symbol) plus slack; the
// careful path inside encode_symbol_fast is entered at most once per such run and the caller
// re-checks the budget every chunk.
template <bool U16, bool FIXUP, bool UPD>
__device__ __forceinline__ void encode_chunk(const Tree<U16> &T, EncState &S, const uint4 cur, uint32_t p,
                                             uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst)
{
    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
    double         r[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        r[i] = rc[UPD ? p + i : nfreeze];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s   = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const uint32_t nup = UPD ? p + i : nfreeze; // wave-uniform
        uint32_t       lo, hi;
        T.template get_frequency<UPD>(s, nup, true, lo, hi);
        encode_symbol_fast<FIXUP>(S, lo, hi, 257u + nup, r[i], sh, wdst);
    }
}

template <bool U16, bool FIXUP>
__global__ void __launch_bounds__(64) k_encode(EncArgs a)
{
    __shared__ uint32_t lds[Tree<U16>::kDwords];
    const uint32_t lane = threadIdx.x;
    const uint64_t blk0 = (uint64_t)blockIdx.x * a.lanes; // wave-uniform
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < a.lanes && blk < a.nblocks && !(a.table && a.table[blk].index == kIdleEntry);

    for (uint32_t i = lane; i < Tree<U16>::kDwords / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    Tree<U16> T;
    T.init(lds, lane);

    // wave-uniform bases (SGPR pairs) + 32-bit per-lane offsets
    const uint8_t *wsrc;
    const EncLane  EL    = enc_lane(a, blk0, blk, lane, live, wsrc);
    const uint32_t len   = EL.len, soff = EL.soff;
    uint8_t       *wdst  = a.slots + blk0 * a.slot_bytes;
    // Dead lanes of the last wave run the same instruction stream on block blk0's bytes and
    // store into the spare slot behind the last real one, so the hot loop needs no predicate.
    // One block per wave (giant blocks: a.lanes == 1): the 63 other lanes mirror lane 0 -- the same bytes, the same state --
    // and store what it stores where it stores it (a spare slot would lie beyond a 32-bit offset as soon as 64 slots do).
    const uint32_t off0  = live ? lane * (uint32_t)a.slot_bytes
                                : (a.lanes == 1 ? 0u : (uint32_t)(a.nblocks - blk0) * (uint32_t)a.slot_bytes);
    const uint32_t limit = off0 + a.slot_cap;

    // The lock-step loop covers [0, maxlen]; the unrolled path covers whole 16-byte chunks
    // strictly below the shortest live block's last symbol.
    const uint32_t minlen  = __builtin_amdgcn_readfirstlane(wave_min(live ? len : 0xFFFFFFFFu));
    // bytes the unrolled path may add per chunk without any per-store check: 16 x 4 + slack
    constexpr uint32_t kChunkBudget = 16 * 4 + 32;
    const uint32_t maxlen  = __builtin_amdgcn_readfirstlane(wave_max(live ? len : 0u));
    const uint32_t sh      = 32 - a.code_bits;
    const uint32_t nfreeze = a.nfreeze;
    const rc_ptr   rc      = (rc_ptr)a.rc;

    EncState S;
    enc_init(S, off0);

    uint32_t p        = 0;
    uint32_t main_end = 0;
    if (a.aligned16 && minlen != 0xFFFFFFFFu && minlen > 16)
        main_end = (minlen - 1) & ~15u;

    if (main_end) {
        // (A) adaptive chunks
        const uint32_t a_end = main_end < (nfreeze & ~15u) ? main_end : (nfreeze & ~15u);
        if (p < a_end) {
            uint4 cur = *reinterpret_cast<const uint4 *>(wsrc + soff);
            for (; p < a_end; p += 16) {
                if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit)) {
                    main_end = p; // a slot is nearly full: finish in the checked tail loop
                    break;
                }
                uint4 nxt = cur;
                if (p + 16 < a_end)
                    nxt = *reinterpret_cast<const uint4 *>(wsrc + soff + p + 16);
                encode_chunk<U16, FIXUP, true>(T, S, cur, p, nfreeze, rc, sh, wdst);
                cur = nxt;
            }
        }
        // (M) the one chunk that crosses the freeze point, symbol by symbol
        if (p < main_end && p < nfreeze) {
            const uint32_t m_end = p + 16;
            for (; p < m_end; p++) {
                const uint32_t nup = p < nfreeze ? p : nfreeze;
                uint32_t       lo, hi;
                T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze, lo, hi);
                encode_symbol<FIXUP>(S, lo, hi, 257u + nup, rc[nup], sh, false, wdst, limit);
            }
        }
        // (F) frozen chunks: static model, no LDS writes
        if (p < main_end) {
            uint4 cur = *reinterpret_cast<const uint4 *>(wsrc + soff + p);
            for (; p < main_end; p += 16) {
                if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit)) {
                    main_end = p;
                    break;
                }
                uint4 nxt = cur;
                if (p + 16 < main_end)
                    nxt = *reinterpret_cast<const uint4 *>(wsrc + soff + p + 16);
                encode_chunk<U16, FIXUP, false>(T, S, cur, p, nfreeze, rc, sh, wdst);
                cur = nxt;
            }
        }
    }

    // Tail: symbol by symbol with per-lane predicates (ragged lengths, the EOF symbol).
    for (; p <= maxlen; p++) {
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   r   = rc[nup];
        const uint32_t c   = 257u + nup;
        if (live && p < len) {
            uint32_t lo, hi;
            // The update of a block's last symbol is never observed (the EOF range is
            // derived), and skipping it keeps every u16 node below 65536.
            T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze && p + 1 != len, lo, hi);
            encode_symbol<FIXUP>(S, lo, hi, c, r, sh, false, wdst, limit);
        } else if (live && p == len) {
            // EOF symbol (codec.rs:108): cum(256) = count-1, cum(257) = count
            const uint32_t shifts = encode_symbol<FIXUP>(S, c - 1, c, c, r, sh, true, wdst, limit);
            const uint32_t size   = encode_finish(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[EL.ob]  = size;
            a.status[EL.ob] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
}

// ======================================================================================
// encode, two waves per 64 blocks (the production path for u16 trees)
//
// The tree (32 KiB per 64 blocks) caps a CU at four groups of 64 blocks -- one wave per SIMD
// if a group is one wave, and a lone wave issues one instruction per ~4 cycles.  The model
// (tree query + update) does not depend on the coder's interval state, so a group is split
// into a MODEL wave and a CODER wave that share the group's LDS:
//   wave 0: input bytes -> get_frequency -> (low, high) pairs into an LDS ring
//   wave 1: ring -> interval narrowing, renormalisation, bit output
// Eight waves per CU = two per SIMD, so each SIMD always has a second instruction stream to
// issue from.  (Measured, profiles/r01_ubench: a gfx950 SIMD retires the VOP3-type ops this
// code is made of at ~4.5 cycles per wave-instruction however many waves feed it, so the two
// streams together run at the SIMD's VALU rate.  Two three-wave variants were built, passed
// the whole parity suite and were removed because they were slower: the model split by tree
// level (+35 %: duplicated per-symbol work) and a three-stage pipeline nodes -> sums -> coder
// with no duplicated work (+83 %: the LDS only has room for 2-symbol ring halves, and three
// synchronised waves get LESS aggregate VALU throughput than two, 5.7 vs 4.9 cycles per
// instruction in tools/ubench "40 VALU + barrier").)  The ring holds 2 x 8 symbols x 64 lanes x 8 B = 8 KiB (40 KiB per workgroup,
// four workgroups = the CU's 160 KiB exactly); one s_barrier per 8 symbols hands a half over.
// Only LDS traffic must be complete at the hand-off, so the barrier waits on lgkmcnt alone:
// the coder's stores and the model's prefetch loads stay in flight across it.
// ======================================================================================
constexpr uint32_t kRingSlots = 8;                               // symbols per hand-off
constexpr uint32_t kRingBytes = 2 * kRingSlots * 64 * 8;         // two halves of uint2[8][64]

__device__ __forceinline__ void pair_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Input path of the model wave: every lane reads its own block one whole 128-byte line at a
// time (eight 16-byte loads issued back to back), so each line crosses the L2 <-> fabric
// boundary once: FETCH_SIZE 2.14e6 KiB per 4 GiB pass, identical to a clean streaming read,
// against 7.4e6-9.5e6 KiB with 16 bytes per visit (the line was evicted between visits), and
// WRITE_SIZE drops 38 % as well (less L2 pollution).  c[] is the current line, n[] the next
// one, already in flight; the chunk index is wave-uniform (the lanes advance in lock-step),
// so pop() is a scalar switch.  Costs 2.8 % of kernel time (profiles/r01_traffic_matrix.txt);
// (16 bytes per visit: 2.8 % faster, 2.4-3.2 x the traffic).
struct ChunkQueue {
    const uint8_t *base; // wave-uniform
    uint32_t       soff; // this lane's block offset
    uint32_t       last; // offset of the last 16-byte chunk the unrolled path reads
    uint32_t       nextp;
    uint32_t       idx;  // next chunk of c[] (wave-uniform)
    uint4          c[8], n[8];

    // chunks past the end re-read the last valid chunk (never used)
    __device__ __forceinline__ void prefetch()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t p = nextp + 16 * i;
            n[i] = *reinterpret_cast<const uint4 *>(base + soff + (p < last ? p : last));
        }
        nextp += 128;
    }
    // current line <- prefetched line, then put the line after it in flight
    __device__ __forceinline__ void swap()
    {
#pragma unroll
        for (int i = 0; i < 8; i++)
            c[i] = n[i];
        idx = 0;
        prefetch();
    }
    __device__ __forceinline__ void init(const uint8_t *b, uint32_t so, uint32_t e)
    {
        base = b; soff = so; last = e - 16; nextp = 0;
        prefetch();
        swap();
    }
    // the next 16 bytes of every lane's block
    __device__ __forceinline__ uint4 pop()
    {
        uint4 r;
        switch (idx) {
        case 0: r = c[0]; break;
        case 1: r = c[1]; break;
        case 2: r = c[2]; break;
        case 3: r = c[3]; break;
        case 4: r = c[4]; break;
        case 5: r = c[5]; break;
        case 6: r = c[6]; break;
        default: r = c[7]; break;
        }
        if (++idx == 8)
            swap();
        return r;
    }
};

// The pair kernel writes ROW-major group areas -- row r = dword r of the group's 64 lanes -- so that the lanes of a wave,
// whose dword counts stay within one or two of each other on iid bytes, fill whole 128-byte lines within a few symbols
// (L2 <-> fabric write traffic 1.0 x the stream instead of 4.1 x with one padded slot per block), and k_compact_rows
// gathers them.
constexpr int kPairStride = 256 | kSwapped; // row-major group areas, dwords left in accumulator order (redux_coder.hpp)
// The dot-product masks of finish() come from k_mask_table, loaded kMaskAhead symbols ahead (14 VALU instructions per
// symbol became two 16-byte loads).  The ring holds the (low, high) pairs of two consecutive symbols side by side (16 bytes
// per lane), written by one ds_write_b128 per two symbols and read by one ds_read_b128.
__device__ __forceinline__ uint32_t ring_at(uint32_t i, uint32_t lane) // index of symbol i's entry, in uint2 units
{
    return (i >> 1) * 128u + lane * 2u + (i & 1u);
}
// (Tried: the coder wave, which has the slack, pulling every input line into the L2 a line-time before the model wave asks
// for it: 1.3 % faster, but the lines do not stay in the L2 until they are used and FETCH_SIZE doubles.  Not kept: the read
// side stays at 1.00 x the input.)
constexpr int kMaskAhead = 8; // symbols between a mask row's load and its use (4: 11.25 ms, 8: 11.01 ms when it was introduced)
static_assert(kMaskAhead == 4 || kMaskAhead == 8, "prime() covers these");
static_assert(16 % kMaskAhead == 0, "slot i % AHEAD must mean the same in every chunk");
struct MaskPipe {
    uint4    s[kMaskAhead], m[kMaskAhead];
    uint32_t four; // the constant 4 in a VGPR: SDWA has no inline constants on gfx9
    __amdgpu_buffer_rsrc_t rsrc; // k_mask_table as a raw buffer (257 rows of 16 bytes)
    __device__ __forceinline__ void init()
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(k_mask_table.v), 0, sizeof k_mask_table.v, 0x00027000);
        four = 4;
        asm volatile("" : "+v"(four));
    }
    // byte K of w, times 16 = the row's offset: one SDWA shift (the byte select is part of the operand) instead of
    // extract + shift -- the model wave's instruction count is what the kernel follows (section 4.0)
    __device__ __forceinline__ uint32_t row_of(int K, uint32_t w) const // K: a constant once the caller's loop is unrolled
    {
        uint32_t off;
        if (K == 0)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(four), "v"(w));
        else if (K == 1)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(four), "v"(w));
        else if (K == 2)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(four), "v"(w));
        else
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(four), "v"(w));
        return off;
    }
    __device__ __forceinline__ void load(int slot, uint32_t row) // row = 16 * symbol
    {
        const char *e = reinterpret_cast<const char *>(k_mask_table.v) + row;
        // (as buffer loads: a plain 16-byte load whose dwords are used in two places -- addends and sums -- is split by
        // the compiler into a 12- and a 4-byte gather, and the texture-address path pays per gather; the intrinsic stays whole)
        (void)e;
        const auto r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row, 0, 0), r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row + 16u, 0, 0);
        s[slot] = make_uint4(r0[0], r0[1], r0[2], r0[3]);
        m[slot] = make_uint4(r1[0], r1[1], r1[2], r1[3]);
        const uint32_t sym = row >> 4; // (only the probe below uses it)
    }
    // the first AHEAD symbols of a chunk (entering a run of model_chunk calls)
    __device__ __forceinline__ void prime(const uint4 cur)
    {
        init();
        const uint32_t w[2] = {cur.x, cur.y};
        load(0, row_of(0, w[0])); load(1, row_of(1, w[0])); load(2, row_of(2, w[0])); load(3, row_of(3, w[0]));
        if (kMaskAhead == 8) {
            load(4 % kMaskAhead, row_of(0, w[1])); load(5 % kMaskAhead, row_of(1, w[1]));
            load(6 % kMaskAhead, row_of(2, w[1])); load(7 % kMaskAhead, row_of(3, w[1]));
        }
    }
};

template <bool UPD>
__device__ __forceinline__ void model_chunk(const Tree<true> &T, uint2 *ring, uint32_t lane, const uint4 cur, const uint4 nxt,
                                            MaskPipe &mp, uint32_t p, uint32_t nfreeze, uint32_t *top = nullptr)
{
    constexpr int kInFlight = UPD ? 7 : 8; // LDS ops of one symbol (node 128 of an adapting model lives in a register)
    const uint32_t w[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
    auto sym = [&](int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; };
    // (Keeping the symbols the mask loads extracted, 8 registers, instead of extracting them again here: one VALU
    // instruction fewer per symbol and 0.9 % slower, profiles/r02_masktable/ab.txt.)
    // software pipeline: symbol i+1's LDS ops are in flight while symbol i's sums are formed
    // (depths 2 and 3 measured no faster in round 1 and slower with the mask table, 11.37 / 11.72 against 11.02 ms: LDS latency is not what the model wave waits for)
    constexpr int D = 1;
    Tree<true>::Nodes q[D + 1];
    uint2             held = make_uint2(0, 0); // the even symbol of a pair, until the odd one is done
    // (the adapting model takes its addends from the symbol's masks, Tree::issue_masked)
    constexpr bool kMasked = UPD && D == 1;
#pragma unroll
    for (int d = 0; d < D; d++)
        q[d] = kMasked ? T.issue_masked(sym(d), mp.s[d % kMaskAhead], top) : T.template issue<UPD>(sym(d), true, top);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s   = sym(i);
        const uint32_t nup = UPD ? p + i : nfreeze;
        // At the hand-over in the middle of the chunk the next symbol's eight LDS ops are issued
        // AFTER this symbol's ring write and stay in flight across the barrier (LDS ops of a wave
        // complete in order, so lgkmcnt <= 8 means the ring half is written).
        const bool late = i == 7;
        if (i + D < 16 && !late) {
            q[D] = kMasked ? T.issue_masked(sym(i + D), mp.s[(i + D) % kMaskAhead], top) : T.template issue<UPD>(sym(i + D), true, top);
            __builtin_amdgcn_sched_barrier(0);
        }
        // One hand-placed s_waitcnt for all seven node values of symbol i -- the ring write of i - 1 and the seven
        // atomics of i + 1 are younger, so lgkmcnt(8) -- instead of the compiler's four, one in front of each
        // v_perm: with the masks no longer computed the model wave is the pair's critical path and every issue slot
        // of it counts (10.78 -> 10.61 ms; with a sched_barrier behind the wait 10.67; when the masks were still
        // computed the same change measured 1 % slower).  The wait is a real S_WAITCNT, so the compiler's pass sees
        // it and drops its own; where the count is different (chunk start, around the mid-chunk barrier) it is left
        // to the compiler.  finish_tab() uses the masks loaded second first, so one vmcnt wait covers both loads.
        uint32_t lo, hi;
        if (D == 1 && i + D < 16 && !late && i != 8 && i != 0 && kInFlight == 7)
        {
            if (i & 1) // (no ring write between the atomics of i and of i + 1)
                __builtin_amdgcn_s_waitcnt(0xC00F | 0x70 | (7 << 8));
            else
                __builtin_amdgcn_s_waitcnt(0xC00F | 0x70 | (8 << 8)); // vmcnt and expcnt fields all ones = no wait
        }
        T.finish_tab(s, nup, q[0], mp.s[i % kMaskAhead], mp.m[i % kMaskAhead], lo, hi);
        mp.load(i % kMaskAhead, mp.row_of((i + kMaskAhead) & 3, w[(i + kMaskAhead) >> 2])); // (the last ones are the next chunk's)
        // (lo and hi come out of v_dot2, and gfx950 wants three wait states between a dot result and an
        // LDS instruction reading it: the compiler puts an s_nop here.  Filling the gap instead -- the next symbol's
        // address preparation pinned there in round 1, the two mask loads ordered there with sched_group_barrier
        // in round 2 -- removes the s_nop and is no faster (10.64 against 10.53 ms for the latter).)
        if (i & 1) {
            *reinterpret_cast<uint4 *>(ring + ring_at(i - 1, lane)) = make_uint4(held.x, held.y, lo, hi);
        } else
            held = make_uint2(lo, hi);
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            q[D] = kMasked ? T.issue_masked(sym(i + D), mp.s[(i + D) % kMaskAhead], top) : T.template issue<UPD>(sym(i + D), true, top);
            if (kInFlight == 7)
                asm volatile("s_waitcnt lgkmcnt(7)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
        } else if ((i & 7) == 7)
            pair_barrier();
#pragma unroll
        for (int d = 0; d < D; d++)
            q[d] = q[d + 1];
    }
}

// MODE 0: adaptive chunk (reciprocals rc[p..p+15]); MODE 1: frozen chunk (rc[nfreeze]).
// MODE 0 takes the reciprocals of its first eight symbols in r[] and leaves those of the next
// chunk's first eight there: each half loads the following half's eight right after its own
// ring reads have arrived, at the start of an eight-symbol stretch with no lgkmcnt wait in it.
// (SMEM shares lgkmcnt with LDS and returns out of order, so any LDS wait or ring barrier also
// waits for every scalar load in flight; loaded at the top of the chunk, their miss latency sat
// in front of the first barrier.  Sixteen at a time would need 64 SGPRs: spills.)
template <bool FIXUP, int MODE, bool CB32>
__device__ __forceinline__ void coder_chunk(EncState &S, const uint2 *ring, uint32_t lane, uint32_t p,
                                            uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst, double (&r)[8])
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        pair_barrier();
        uint2 lh[8]; // the whole half at once: one LDS round trip per 8 symbols
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const uint4 two = *reinterpret_cast<const uint4 *>(ring + ring_at(h * 8 + i, lane));
            lh[i]     = make_uint2(two.x, two.y);
            lh[i + 1] = make_uint2(two.z, two.w);
        }
        double rn[8];
        if (MODE == 0) {
            uint32_t zero; // opaque 0 that "depends" on the ring data: pins the loads behind the LDS wait
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero) : "v"(lh[0].x));
            const rc_ptr nb = rc + (p + 8u * h + 8u + zero); // the table has 32 entries of slack (geometry())
#pragma unroll
            for (int i = 0; i < 8; i++)
                rn[i] = nb[i];
        }
        // Eight symbols straight-line; the rare symbol whose pending run needs more than one
        // 32-bit append only raises a flag, and the half is then redone from the saved state
        // with the general encode_symbol (no per-symbol branch, no merge of two state versions).
        const EncState S0  = S;
        SpecCarry      C   = spec_begin(S);
        uint32_t       mx  = 0;          // largest append of the half (pairs of symbols fold into one v_max3_u32)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t nup = MODE == 0 ? p + h * 8 + i : nfreeze;
            // keeps ISel from turning (u64 >> 32) -> f64 into a 64-bit conversion (+1 v_add_f64); converting
            // as signed avoids that too, but v_cvt_f64_i32 measured 9 % slower for the whole kernel.
            // Applied to the ring value in place: on a copy it costs a v_mov per symbol, because the redo below reads lh[i] again.
            asm volatile("" : "+v"(lh[i].y));
            const uint32_t hi = lh[i].y;
            const uint32_t m = encode_symbol_spec<FIXUP, CB32, kPairStride>(S, C, lh[i].x, hi, 257u + nup, MODE == 0 ? r[i] : rc[nfreeze], sh, wdst);
            mx               = m > mx ? m : mx;
        }
        spec_end(S, C);
        const uint64_t bad = __builtin_amdgcn_ballot_w64(mx > 32u);
        if (__builtin_expect(bad != 0, 0)) {
            S = S0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t nup = MODE == 0 ? p + h * 8 + i : nfreeze;
                encode_symbol<FIXUP, kPairStride>(S, lh[i].x, lh[i].y, 257u + nup, MODE == 0 ? r[i] : rc[nfreeze], sh, false, wdst, 0xFFFFFFFFu);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                r[i] = rn[i];
        }
    }
}

// any chunk, rolled, every store checked against the slot limit
template <bool FIXUP>
__device__ __forceinline__ void coder_chunk_checked(EncState &S, const uint2 *ring, uint32_t lane, uint32_t p,
                                                    uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst,
                                                    uint32_t limit)
{
    for (uint32_t i = 0; i < 16; i++) {
        if ((i & 7) == 0)
            pair_barrier();
        const uint2    lh  = ring[ring_at(i, lane)];
        const uint32_t q   = p + i;
        const uint32_t nup = q < nfreeze ? q : nfreeze;
        encode_symbol<FIXUP, kPairStride>(S, lh.x, lh.y, 257u + nup, rc[nup], sh, false, wdst, limit);
    }
}

// The pair kernel needs every SIMD to hold exactly ONE model wave and ONE coder wave.  Where the
// two waves of a 128-thread workgroup land is up to the dispatcher: launched on an idle chip it
// alternates them perfectly, launched right after another kernel it puts two first-waves on
// some SIMDs (profiles/r01_final/placement_census.txt) -- two model waves at half speed each,
// which the whole lock-step kernel then waits for (0.5-3 ms of 13).  So the roles are not tied
// to the wave index: they are booked per CU at run time (below).
// (Tried instead: whole-CU workgroups of eight waves = four pairs, waves w and w+4 sharing a
// SIMD.  Placement is then perfect by construction, but the eight-wave s_barrier couples the
// four pairs and the kernel takes 14.05 ms against 12.8 ms.)
constexpr uint32_t kClaimWords = 2048; // (xcc:3, se:3, sh:1, cu:4) -> one word per CU
constexpr uint32_t kPairDwords = Tree<true>::kDwords + kRingBytes / 4;

template <bool FIXUP, bool CB32>
__global__ void __launch_bounds__(128) k_encode_pair(EncArgs a)
{
    __shared__ uint32_t lds[kPairDwords];
    const uint32_t w8   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // first or second wave of the workgroup
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk0 = (uint64_t)blockIdx.x * a.lanes;
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < a.lanes && blk < a.nblocks && !(a.table && a.table[blk].index == kIdleEntry);

    for (uint32_t i = threadIdx.x; i < Tree<true>::kDwords / 4; i += 128)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    uint32_t role = w8;
    uint2   *ring = reinterpret_cast<uint2 *>(lds + Tree<true>::kDwords);
    // claims[cu] counts the model waves (bits 4s..4s+3) and coder waves (bits 16+4s..) booked on
    // SIMD s of that CU.  A workgroup whose waves sit on SIMDs (s0, s1) books (model, coder) =
    // (s0, s1), or (s1, s0) when that collides with fewer roles already booked, and returns its
    // booking when its coder wave ends.  Greedy, so not always perfect, but on an idle chip the
    // dispatcher's own choice is kept and after a compaction every SIMD still gets (1, 1).
    uint32_t claim_delta = 0; // (a VGPR on purpose: it is live across the whole kernel)
    {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // (the last 16 bytes of the ring: the model wave writes there only after the first hand-off)
        volatile uint32_t *book = reinterpret_cast<volatile uint32_t *>(ring) + kRingBytes / 4 - 4;
        if (lane == 0)
            book[w8] = (hwid >> 4) & 3u; // my SIMD
        __syncthreads();
        uint32_t *claim_word = a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu)); // (xcc, se, sh, cu)
        if (w8 == 0 && lane == 0) {
            const uint32_t s0 = book[0], s1 = book[1];
            const uint32_t straight = (1u << (4 * s0)) | (1u << (16 + 4 * s1));
            const uint32_t flipped  = (1u << (4 * s1)) | (1u << (16 + 4 * s0));
            uint32_t       old      = __hip_atomic_load(claim_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t       flip;
            do {
                const uint32_t cs = ((old >> (4 * s0)) & 15u) + ((old >> (16 + 4 * s1)) & 15u);
                const uint32_t cf = ((old >> (4 * s1)) & 15u) + ((old >> (16 + 4 * s0)) & 15u);
                flip              = cf < cs;
            } while (!__hip_atomic_compare_exchange_strong(claim_word, &old, old + (flip ? flipped : straight),
                                                           __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            book[2] = flip;
            book[3] = flip ? flipped : straight;
        }
        __syncthreads();
        role ^= __builtin_amdgcn_readfirstlane(book[2]);
        claim_delta = book[3];
    }
    const uint32_t wave = role; // 0 = model, 1 = coder
    Tree<true> T;
    T.init(lds, lane);

    const uint8_t *wsrc;
    const EncLane  EL    = enc_lane(a, blk0, blk, lane, live, wsrc);
    const uint32_t len   = EL.len, soff = EL.soff;
    // row-major group area: dword r of lane l at wdst + 256 r + 4 l (dead lanes own a column too).
    // The areas are an ODD number of 128-byte lines apart: all groups write row r at about the
    // same time, and an even stride folds those lines onto a fraction of the L2 sets.
    uint8_t       *wdst  = a.slots + (uint64_t)blockIdx.x * (64 * a.slot_bytes + 128);
    const uint32_t off0  = lane * 4u;
    const uint32_t limit = off0 + (a.slot_cap / 4u) * 256u;

    const uint32_t minlen  = __builtin_amdgcn_readfirstlane(wave_min(live ? len : 0xFFFFFFFFu));
    const uint32_t maxlen  = __builtin_amdgcn_readfirstlane(wave_max(live ? len : 0u));
    const uint32_t sh      = 32 - a.code_bits;
    const uint32_t nfreeze = a.nfreeze;
    const rc_ptr   rc      = (rc_ptr)a.rc;
    constexpr uint32_t kChunkBudget = (16 * 4 + 32) * (stride_of<kPairStride> / 4);

    // both waves derive the same chunk schedule from wave-uniform values
    uint32_t main_end = 0;
    if (a.aligned16 && minlen != 0xFFFFFFFFu && minlen > 16)
        main_end = (minlen - 1) & ~15u;
    const uint32_t a_end = main_end < (nfreeze & ~15u) ? main_end : (nfreeze & ~15u); // adaptive chunks
    const uint32_t m_end = (a_end < main_end && a_end < nfreeze) ? a_end + 16 : a_end; // freeze-crossing chunk

    EncState S;
    enc_init(S, off0);

    if (wave == 0) {
        // ---------------- model wave ----------------
        // The model wave is the pair's critical path (it works ~590 cycles per symbol, the coder wave
        // ~430 and then waits at the ring barrier), but the SIMD's arbiter serves the two waves
        // round-robin: raising the model wave's issue priority lets it run at nearly the lone-wave
        // rate while the coder wave fills the gaps.  15.96 -> 14.0 ms.
        __builtin_amdgcn_s_setprio(3);
        if (main_end) {
            uint32_t p = 0;
            ChunkQueue Q;
            Q.init(wsrc, soff, main_end);
#define NEXT_CHUNK() Q.pop()
            // cur = the chunk at p, nxt the one after it (model_chunk loads masks a few symbols ahead, across the chunk boundary)
            uint4    cur = NEXT_CHUNK(), nxt;
            MaskPipe mp;
            mp.prime(cur);
            uint32_t top = 0; // node 128 in this lane's half, as the LDS dword would hold it
            for (; p < a_end; p += 16, cur = nxt) {
                nxt = NEXT_CHUNK(); // (one past the end at the last turn: the queue re-reads its last chunk)
                model_chunk<true>(T, ring, lane, cur, nxt, mp, p, nfreeze, &top);
            }
            T.add(T.A[7], top); // from here on (freeze-crossing chunk, frozen chunks, the coder wave's tail) the tree is read from LDS
            for (; p < m_end; p += 16) { // rolled: the update stops in the middle of this chunk
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t q   = p + i;
                    const uint32_t nup = q < nfreeze ? q : nfreeze;
                    uint32_t       lo, hi;
                    T.template get_frequency<true>(wsrc[soff + q], nup, q < nfreeze, lo, hi);
                    ring[ring_at(i, lane)] = make_uint2(lo, hi);
                    if ((i & 7) == 7)
                        pair_barrier();
                }
                cur = NEXT_CHUNK();
                mp.prime(cur);
            }
            for (; p < main_end; p += 16, cur = nxt) {
                nxt = NEXT_CHUNK();
                model_chunk<false>(T, ring, lane, cur, nxt, mp, p, nfreeze);
            }
#undef NEXT_CHUNK
        }
    } else {
        // ---------------- coder wave ----------------
        uint32_t p = 0;
        double   r[8];      // reciprocals of the first eight symbols of chunk r_at
        uint32_t r_at = ~0u;
        for (; p < main_end; p += 16) {
            if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit) || (p >= a_end && p < m_end))
                coder_chunk_checked<FIXUP>(S, ring, lane, p, nfreeze, rc, sh, wdst, limit);
            else if (p < a_end) {
                if (r_at != p) { // first chunk, or the previous one took the checked path
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        r[i] = rc[p + i];
                }
                coder_chunk<FIXUP, 0, CB32>(S, ring, lane, p, nfreeze, rc, sh, wdst, r);
                r_at = p + 16;
            } else
                coder_chunk<FIXUP, 1, CB32>(S, ring, lane, p, nfreeze, rc, sh, wdst, r);
        }
    }
    __syncthreads(); // the model wave's last updates are in LDS before the tail reads the tree
    if (wave == 0)
        return;

    // Tail (coder wave only): symbol by symbol with per-lane predicates.
    for (uint32_t p = main_end; p <= maxlen; p++) {
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   r   = rc[nup];
        const uint32_t c   = 257u + nup;
        if (live && p < len) {
            uint32_t lo, hi;
            T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze && p + 1 != len, lo, hi);
            encode_symbol<FIXUP, kPairStride>(S, lo, hi, c, r, sh, false, wdst, limit);
        } else if (live && p == len) {
            const uint32_t shifts = encode_symbol<FIXUP, kPairStride>(S, c - 1, c, c, r, sh, true, wdst, limit);
            const uint32_t size   = encode_finish<kPairStride>(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[EL.ob]  = size;
            a.status[EL.ob] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
    if (lane == 0) { // waves do not migrate: the same CU as at the start
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_sub(a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu)), claim_delta, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace redux
