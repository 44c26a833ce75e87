// redux_gen.hpp -- the lock-step block coder for the OTHER symbol widths the reference tests
// (src/model/tests.rs:95-251: 4- and 12-bit symbols), code_bits <= 32 (gfx950 only).  SURVEY section 8(f).3.
//
// redux_any.hpp covers every Parameters triple as a one-lane-per-block rendering of the reference's loops.  For
// symbol_bits 4 and 12 this file is the MI355X form of the same functions, built like the 8-bit kernels:
//   * one LANE per block, 64 blocks per wave, all lanes on the same symbol index, so the model's total
//     (2^symbol_bits + 1 + symbols coded, until the freq_max freeze: adaptive_tree.rs:84) is WAVE-UNIFORM and the two
//     u64 divisions of codec.rs:59-60 are multiplications by a per-step reciprocal (scale_div<FIXUP = true>: a block
//     of 4-bit symbols passes count 2^17);
//   * the Fenwick tree as increments d[i] = tree[i] - lowbit(i), i = 1 .. 2^symbol_bits - 1, in LDS, in per-lane
//     columns.  4-bit symbols: u32, row e of lane l at e * 256 + 4 l: 16 rows = 4 KiB per wave of 64 blocks.  12-bit
//     symbols: 4096 nodes per block, so the LDS, not the wave width, sets how many blocks a CU holds: u16 nodes (blocks of
//     at most 65535 symbols = 98,302 bytes; longer ones run on the one-lane kernels), two lanes to a dword, SIXTEEN live
//     lanes per wave: 128 KiB, one workgroup per CU.  (Round 2 kept 64 trees of 1 MiB per wave in the workspace and walked them with global
//     fetch-adds: 6.6 GB/s on the full grid, the atomics going to HBM.)
//   * get_frequency = one fetch-add per level (addend 1 where update(s+1) increments the node, 0 where the prefix
//     sums only read it) + two masked sums (adaptive_tree.rs:63-92), closed-form renormalisation and bit output
//     exactly as encode_symbol (redux_coder.hpp); the decoder's descent probes the same nodes (adaptive_tree.rs:115-136);
//   * symbols are read_bits(symbol_bits) MSB-first (bitio/mod.rs:78-120): two per byte, or two per three bytes; a
//     trailing partial symbol is dropped and the block ends (Err(Eof) -> EOF symbol, codec.rs:108); the decoder writes
//     write_bits(symbol, symbol_bits) and never flushes a partial byte (lib.rs:113-120).
// Everything is predicated per lane (ragged blocks, errors): this path is about being a designed kernel instead of a
// port, not about the last instruction; the 8-bit kernels are the tuned ones.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"

#include <type_traits>
#include "redux_decode.hpp" // BitIn
#include "redux_encode.hpp" // wave_max

#include "../../include/redux_hip.h"

namespace redux {

// INLDS (always for 4-bit symbols): the trees of a wave's blocks in LDS.  !INLDS (12-bit symbols, large grids, decode):
// 64 trees of u32 per wave in the workspace, 1 MiB, walked with global loads and fetch-adds -- slower per step, but
// 64 blocks per wave and several waves per SIMD instead of 16 blocks per CU: on the full grid the decoder, whose
// descent is twelve DEPENDENT probes, is faster this way (6.5 against 4.2 GB/s), the encoder, whose twelve fetch-adds
// are independent, in LDS (8.6 against 6.6 GB/s); below ~16,384 blocks both are 3 x faster in LDS.
template <int SB, bool INLDS = true>
struct GenTree {
    static constexpr bool     kU16    = INLDS && SB > 8;    // u16 nodes, lanes l and l + kBlocks / 2 in one dword
    // live lanes (= blocks) per wave: what 128 KiB of LDS hold as u16 nodes (12-bit symbols: 16, 11: 32, 9 and 10: 64)
    static constexpr uint32_t kBlocks = kU16 ? (SB >= 12 ? 16u : SB == 11 ? 32u : 64u) : 64u;
    static constexpr uint32_t kRows   = 1u << SB;
    static constexpr uint32_t kMask   = kRows - 1u;
    static constexpr uint32_t kPitch  = kU16 ? kBlocks / 2u : 64u; // dwords per row
    static constexpr uint32_t kDwords = kRows * kPitch;     // 4-bit: 4 KiB; 12-bit: 128 KiB of LDS, or 1 MiB of workspace per wave
    static constexpr uint32_t kLdsDwords = INLDS ? kDwords : 64u;
    // u16 nodes hold increments: a node of a block of n symbols receives at most n of them
    static constexpr uint32_t kMaxSymbols = SB > 8 ? 65535u : 0xFFFFFFFFu;

    uint32_t *base; // row e of this lane: base[e * kPitch]  (base already points at the lane's dword column)
    uint32_t  sh;   // kU16: bit position of this lane's half

    __device__ __forceinline__ void init(uint32_t *mem, uint32_t lane) // mem: this wave's LDS array / workspace tree
    {
        base = mem + (kU16 ? (lane & (kPitch - 1u)) : lane);
        sh   = kU16 ? ((lane / kPitch) & 1u) * 16u : 0u;
    }
    __device__ __forceinline__ uint32_t fetch_add(uint32_t e, uint32_t v) const
    {
        const uint32_t w = INLDS ? __hip_atomic_fetch_add(base + e * kPitch, v << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                 : __hip_atomic_fetch_add(base + e * kPitch, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return kU16 ? (w >> sh) & 0xFFFFu : w;
    }
    __device__ __forceinline__ uint32_t load(uint32_t e) const
    {
        const uint32_t w = base[e * kPitch];
        return kU16 ? (w >> sh) & 0xFFFFu : w;
    }

    // get_frequency(s) for a data symbol (adaptive_tree.rs:105-113); nup = updates so far, upd = the model is not frozen
    __device__ __forceinline__ void get_frequency(uint32_t s, uint32_t nup, bool upd, uint32_t &lo, uint32_t &hi) const
    {
        uint32_t       x[SB];
        const uint32_t m = s + 1;
#pragma unroll
        for (int b = 0; b < SB; b++) {
            const uint32_t e = (s | (1u << b)) & (kMask << b); // the one node of level b on s's root path
            x[b]             = fetch_add(e, (upd && !((s >> b) & 1u)) ? 1u : 0u);
        }
        uint32_t ls = s, hs = m;
#pragma unroll
        for (int b = 0; b < SB; b++) {
            ls += ((s >> b) & 1u) ? x[b] : 0u;
            hs += ((m >> b) & 1u) ? x[b] : 0u;
        }
        lo = ls;
        hi = hs + (m >> SB) * nup; // s + 1 == 2^SB selects the derived node 2^SB = 2^SB + #updates
    }
    // the share of levels [B0, B1) in get_frequency(s): fetch-adds and masked sums of those levels only; the level-
    // independent terms (s, s + 1, the derived node) ride with the share that starts at level 0.  The shares of a
    // partition of the levels add up to get_frequency's (low, high).  In two halves, so that a loop can put the NEXT
    // symbol's fetch-adds into the LDS queue before it consumes this symbol's values (the queue is in order: the next
    // symbol's fetch-adds see this symbol's increments).
    struct Nodes {
        uint32_t x[SB];
    };
    template <int B0, int B1>
    __device__ __forceinline__ Nodes issue_part(uint32_t s, bool upd) const
    {
        Nodes n;
#pragma unroll
        for (int b = B0; b < B1; b++) {
            const uint32_t e = (s | (1u << b)) & (kMask << b);
            n.x[b]           = fetch_add(e, (upd && !((s >> b) & 1u)) ? 1u : 0u);
        }
        return n;
    }
    template <int B0, int B1>
    __device__ __forceinline__ void finish_part(uint32_t s, uint32_t nup, const Nodes &n, uint32_t &lo, uint32_t &hi) const
    {
        const uint32_t m  = s + 1;
        uint32_t       ls = B0 == 0 ? s : 0u, hs = B0 == 0 ? m + (m >> SB) * nup : 0u;
#pragma unroll
        for (int b = B0; b < B1; b++) {
            ls += ((s >> b) & 1u) ? n.x[b] : 0u;
            hs += ((m >> b) & 1u) ? n.x[b] : 0u;
        }
        lo = ls;
        hi = hs;
    }
};

// symbol k of a block at src: read_bits(SB) MSB-first (bitio/mod.rs:78-120), in two halves, so that a loop can LOAD the bytes
// of symbol k one step before it DECODES them (the value that crosses the iteration is the raw load, which nothing waits for
// until it is decoded).  4-bit symbols: one byte; 12-bit: symbols 2j, 2j+1 share bytes 3j .. 3j+2; any other width: the
// three bytes from the symbol's first one (a symbol of at most 12 bits starting at bit r <= 7 of a byte ends inside them),
// the indices clamped to lastb, the last byte the lane may read -- bytes a symbol does not reach do not matter.
template <int SB>
__device__ __forceinline__ uint32_t gen_symbol_load(const uint8_t *src, uint32_t k, uint32_t lastb)
{
    if (SB == 4)
        return src[k >> 1];
    if (SB == 12) {
        const uint32_t i = k + (k >> 1);
        return (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8);
    }
    const uint32_t i  = (k * SB) >> 3;
    const uint32_t i1 = i + 1 < lastb ? i + 1 : lastb, i2 = i + 2 < lastb ? i + 2 : lastb;
    return ((uint32_t)src[i] << 16) | ((uint32_t)src[i1] << 8) | (uint32_t)src[i2];
}
template <int SB>
__device__ __forceinline__ uint32_t gen_symbol_decode(uint32_t raw, uint32_t k)
{
    if (SB == 4)
        return (k & 1u) ? (raw & 15u) : (raw >> 4);
    if (SB == 12) {
        const uint32_t b0 = raw & 0xFFu, b1 = raw >> 8;
        return (k & 1u) ? (((b0 & 15u) << 8) | b1) : ((b0 << 4) | (b1 >> 4));
    }
    const uint32_t r = (k * SB) & 7u;
    return (raw >> (24u - SB - r)) & ((1u << SB) - 1u);
}

struct GenEncArgs {
    const uint8_t *in;
    uint64_t       in_len;
    uint64_t       nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    const double  *rc;       // rc[i] = 1 / (2^SB + 1 + i), bumped (k_fill_rc_from)
    uint32_t       block_size;
    uint32_t       slot_cap;
    uint32_t       nfreeze;  // freq_max - (2^SB + 1): updates before the freeze
    uint32_t       code_bits;
};

__global__ void k_fill_rc_from(double *rc, uint32_t n, uint32_t first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double r = 1.0 / (double)(first + i); // correctly rounded, then biased up 4 ulp as k_fill_rc (scale_div)
        rc[i]          = __longlong_as_double(__double_as_longlong(r) + 4);
    }
}

template <int SB>
__global__ void __launch_bounds__(64) k_encode_gen(GenEncArgs a)
{
    typedef GenTree<SB> Tree;
    __shared__ uint32_t lds[Tree::kDwords];
    const uint32_t lane = threadIdx.x;
    const uint64_t blk0 = (uint64_t)blockIdx.x * Tree::kBlocks;
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < Tree::kBlocks && blk < a.nblocks;
    for (uint32_t i = lane; i < Tree::kDwords / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    Tree T;
    T.init(lds, lane);

    uint32_t len = 0;
    if (live) {
        const uint64_t rem = a.in_len - blk * a.block_size;
        len                = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    const uint32_t nsym   = (uint32_t)(((uint64_t)len * 8) / SB); // whole symbols; trailing bits are dropped (codec.rs:108)
    const uint8_t *src    = a.in + (live ? blk : blk0) * (uint64_t)a.block_size;
    uint8_t       *wdst   = a.slots + blk0 * a.slot_bytes;
    const uint32_t off0   = live ? lane * (uint32_t)a.slot_bytes : 0u; // (dead lanes store nothing: everything below is predicated)
    const uint32_t limit  = off0 + a.slot_cap;
    const uint32_t maxsym = __builtin_amdgcn_readfirstlane(wave_max(live ? nsym : 0u));
    const uint32_t sh     = 32 - a.code_bits;
    const uint32_t nfreeze = a.nfreeze;
    const rc_ptr   rc     = (rc_ptr)a.rc;
    constexpr uint32_t kCount0 = (1u << SB) + 1u;

    EncState S;
    enc_init(S, off0);

    // ---- whole waves: turns of 32 symbols (= SB dwords of input per lane), every lane alive, no predicate ----------------
    // As k_encode's chunks (redux_encode.hpp): the turn's input dwords are loaded a turn ahead, its symbols sit at static bit
    // positions (read_bits(SB) MSB-first, bitio/mod.rs:78-120), the reciprocals come eight at a time, and the coder is
    // encode_symbol_fast -- one store site, the long-pending-run case found by one wave-level ballot -- instead of the
    // per-lane predicated encode_symbol, whose store sites under branches make every vector-memory wait a full one.  Runs while
    // every lane has the symbols and the count stays below 2^17 (no quotient fix-up, scale_div; a model that freezes below
    // that never leaves the loop for it); 4-byte aligned blocks only.
    uint32_t p = 0;
    if (SB < 8) {
        constexpr uint32_t U = 32;
        const bool     whole  = __builtin_amdgcn_ballot_w64(live) == ~0ull;
        const uint32_t minsym = __builtin_amdgcn_readfirstlane(wave_min(live ? nsym : 0u));
        constexpr uint32_t kNoFix = (1u << 17) - kCount0;
        const uint32_t e      = (nfreeze < kNoFix || minsym < kNoFix) ? minsym : kNoFix; // (a model that freezes below 2^17 never needs the fix-up)
        const uint32_t fast_end = (whole && (((uintptr_t)a.in | a.block_size) & 3u) == 0) ? (e & ~(U - 1u)) : 0u;
        if (fast_end) {
            constexpr uint32_t kBudget = U * 4 + 32; // bytes a turn may add on the common path (one dword per symbol at most) + slack
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(src);
            uint32_t        cur[SB], nxt[SB];
#pragma unroll
            for (int d = 0; d < SB; d++)
                cur[d] = wp[d];
            // (the loop exists twice: while every symbol of a turn updates the model -- count and reciprocal by induction --, and
            // with both selected per symbol for the turns from the freeze point on)
            auto turns = [&](auto frozen_tag, const uint32_t pend) {
            constexpr bool FRZ = decltype(frozen_tag)::value;
            for (; p < pend; p += U) {
                if (__builtin_amdgcn_ballot_w64(S.off + kBudget > limit))
                    break; // a slot is nearly full: the checked loop below finishes the block
                {
                    const uint32_t t = p + U < fast_end ? (p + U) / U : p / U; // (the last turn re-reads its own dwords)
#pragma unroll
                    for (int d = 0; d < SB; d++)
                        nxt[d] = wp[t * SB + d];
                }
                uint32_t be[SB];
#pragma unroll
                for (int d = 0; d < SB; d++)
                    be[d] = __builtin_bswap32(cur[d]);
#pragma unroll
                for (uint32_t h = 0; h < U / 8; h++) {
                    double r[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t q = p + 8 * h + i;
                        r[i]             = rc[FRZ ? (q < nfreeze ? q : nfreeze) : q];
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 8; i++) {
                        const uint32_t k = 8 * h + i, o = k * SB, d = o >> 5, rr = o & 31u;
                        uint32_t       sym;
                        if (rr + SB <= 32)
                            sym = (be[d] >> (32 - rr - SB)) & Tree::kMask;
                        else
                            sym = ((be[d] << (rr + SB - 32)) | (be[d + 1 < SB ? d + 1 : d] >> (64 - rr - SB))) & Tree::kMask;
                        const uint32_t nup = FRZ ? (p + k < nfreeze ? p + k : nfreeze) : p + k; // wave-uniform; past the freeze point nothing updates
                        uint32_t       lo, hi;
                        T.get_frequency(sym, nup, FRZ ? p + k < nfreeze : true, lo, hi);
                        encode_symbol_fast<false>(S, lo, hi, kCount0 + nup, r[i], sh, wdst);
                    }
                }
#pragma unroll
                for (int d = 0; d < SB; d++)
                    cur[d] = nxt[d];
            }
            };
            const uint32_t a_end = fast_end < (nfreeze & ~(U - 1u)) ? fast_end : (nfreeze & ~(U - 1u));
            turns(std::false_type(), a_end);
            if (p == a_end && a_end < fast_end) // (not left early for a full slot)
                turns(std::true_type(), fast_end);
        }
    }

    // ---- the rest, symbol by symbol with per-lane predicates (ragged blocks, frozen models, the EOF symbol) ----------------
    // The symbol and the reciprocal of step p + 1 are loaded during step p: one wave per CU has nothing else to hide a
    // load behind.  The symbol's load is unconditional (index clamped into the block; a lane without symbols reads its
    // block's first bytes, or the buffer's for an empty input... never past what a.in holds), so that it is not followed
    // by the wait a branch join would put behind it.
    const uint32_t last_sym = nsym ? nsym - 1u : 0u;
    const bool     can_load = live && nsym != 0;
    const uint8_t *psrc     = can_load ? src : reinterpret_cast<const uint8_t *>(a.rc); // (always mapped, at least 33 doubles)
    const uint32_t lastb    = can_load ? len - 1u : 2u;
    uint32_t       k_next   = can_load ? (p < last_sym ? p : last_sym) : 0u;
    uint32_t       raw_next = gen_symbol_load<SB>(psrc, k_next, lastb);
    double         r_next   = rc[p < nfreeze ? p : nfreeze];
    for (; p <= maxsym; p++) {
        const uint32_t nup = p < nfreeze ? p : nfreeze; // wave-uniform
        const double   r   = r_next;
        const uint32_t c   = kCount0 + nup;
        const uint32_t sym = gen_symbol_decode<SB>(raw_next, k_next);
        {
            const uint32_t q = p + 1 < last_sym ? p + 1 : last_sym;
            k_next           = can_load ? q : 0u;
            raw_next         = gen_symbol_load<SB>(psrc, k_next, lastb);
            r_next           = rc[p + 1 < nfreeze ? p + 1 : nfreeze]; // (the table has 32 entries of slack)
        }
        if (live && p < nsym) {
            uint32_t lo, hi;
            T.get_frequency(sym, nup, p < nfreeze, lo, hi);
            encode_symbol<true>(S, lo, hi, c, r, sh, false, wdst, limit);
        } else if (live && p == nsym) {
            // EOF symbol (codec.rs:108): cum(2^SB) = count - 1, cum(2^SB + 1) = count
            const uint32_t shifts = encode_symbol<true>(S, c - 1, c, c, r, sh, true, wdst, limit);
            const uint32_t size   = encode_finish(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[blk]  = size;
            a.status[blk] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
}

// 12-bit symbols: the LDS holds 16 trees per CU, so a CU runs ONE workgroup and three of its SIMDs would idle.  The
// same kernel as four waves on four SIMDs: waves 0-2 are the model, each for four of the tree's twelve levels (symbol
// load a step ahead, its fetch-adds -- the twelve are independent -- and its share of the two masked sums -> a partial
// (low, high) into an LDS ring of its own), wave 3 the coder (adds the shares; interval narrowing, renormalisation, bit
// output, EOF tail); a half of eight symbols is handed over per s_barrier, as in k_encode_pair.  A step costs the
// longest of the four instead of their sum: 8.6 GB/s as one wave, 14.0 with one model wave, 20.5 with two, 23.3 with
// three (the coder wave is the longest then).  All waves derive the same schedule from wave-uniform values.
template <int SB>
__global__ void __launch_bounds__(256) k_encode_gen_pair(GenEncArgs a)
{
    typedef GenTree<SB> Tree;
    constexpr uint32_t kHalf = 8, kModelWaves = 3;
    constexpr int      kSplit1 = SB / 3, kSplit2 = 2 * SB / 3;
    __shared__ uint32_t lds[Tree::kDwords];
    __shared__ uint2    ring[kModelWaves][2 * kHalf * 64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk0 = (uint64_t)blockIdx.x * Tree::kBlocks;
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < Tree::kBlocks && blk < a.nblocks;
    for (uint32_t i = threadIdx.x; i < Tree::kDwords / 4; i += 256)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    uint32_t len = 0;
    if (live) {
        const uint64_t rem = a.in_len - blk * a.block_size;
        len                = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    const uint32_t nsym    = (uint32_t)(((uint64_t)len * 8) / SB); // whole symbols; trailing bits are dropped (codec.rs:108)
    const uint32_t maxsym  = __builtin_amdgcn_readfirstlane(wave_max(live ? nsym : 0u));
    const uint32_t minsym  = __builtin_amdgcn_readfirstlane(wave_min(live ? nsym : 0xFFFFFFFFu));
    const bool     whole   = __builtin_amdgcn_ballot_w64(live) == (Tree::kBlocks == 64 ? ~0ull : (1ull << Tree::kBlocks) - 1ull);
    const uint32_t nfreeze = a.nfreeze;
    const uint32_t nper    = (maxsym + 1 + kHalf - 1) / kHalf; // symbols 0 .. maxsym (the longest block's EOF)
    constexpr uint32_t kCount0 = (1u << SB) + 1u;
    // Lanes without a block leave here (a wave of 12-bit symbols has 16 blocks): what is left of each wave runs with a
    // shorter exec mask, so the coder wave's unpredicated path below stores nothing for them.  (The barriers count waves.)
    if (lane >= Tree::kBlocks)
        return;

    if (wave < kModelWaves) {
        // ---------------- model waves ----------------
        Tree T;
        T.init(lds, lane);
        const uint8_t *src      = a.in + (live ? blk : blk0) * (uint64_t)a.block_size;
        const uint32_t last_sym = nsym ? nsym - 1u : 0u;
        const bool     can_load = live && nsym != 0;
        const uint8_t *psrc     = can_load ? src : reinterpret_cast<const uint8_t *>(a.rc); // (always mapped, at least 33 doubles)
        const uint32_t lastb    = can_load ? len - 1u : 2u;
        uint32_t       raw_next = gen_symbol_load<SB>(psrc, 0, lastb), k_next = 0;
        uint2         *my       = ring[wave];
        auto part = [&](uint32_t sym, uint32_t nup, bool upd, uint32_t &lo, uint32_t &hi) {
            if (wave == 0)
                T.template finish_part<0, kSplit1>(sym, nup, T.template issue_part<0, kSplit1>(sym, upd), lo, hi);
            else if (wave == 1)
                T.template finish_part<kSplit1, kSplit2>(sym, nup, T.template issue_part<kSplit1, kSplit2>(sym, upd), lo, hi);
            else
                T.template finish_part<kSplit2, SB>(sym, nup, T.template issue_part<kSplit2, SB>(sym, upd), lo, hi);
        };
        // The eight symbols of a half are exactly SB bytes of the block, from byte t * SB on: a half whose bytes -- and the next
        // half's -- lie inside EVERY lane's block takes them from (up to three, unaligned) dwords loaded a half earlier, at
        // static bit positions.  (Loading a symbol's two bytes one step ahead, as the other halves do, leaves the load's latency
        // exposed once per symbol: that, not the fetch-adds, was what a step of this kernel waited for.)
        constexpr int kW = (SB + 3) / 4; // dwords that hold a half's SB bytes
        uint32_t      w[kW], wn[kW];
        bool          have_w = false;
        for (uint32_t t = 0; t < nper; t++) {
            // (whole: every lane of the workgroup has a block; the bytes read reach at most 3 past the next half's: inside the block)
            if (whole && ((uint64_t)(t + 2) * SB + 4 <= (uint64_t)minsym * SB / 8)) {
                const uint8_t *hp = src + (uint64_t)t * SB;
                if (!have_w) {
#pragma unroll
                    for (int d = 0; d < kW; d++)
                        w[d] = *reinterpret_cast<const uint32_t *>(hp + 4 * d);
                }
#pragma unroll
                for (int d = 0; d < kW; d++)
                    wn[d] = *reinterpret_cast<const uint32_t *>(hp + SB + 4 * d);
                uint32_t be[kW];
#pragma unroll
                for (int d = 0; d < kW; d++)
                    be[d] = __builtin_bswap32(w[d]);
#pragma unroll
                for (uint32_t i = 0; i < kHalf; i++) {
                    const uint32_t p   = t * kHalf + i;
                    const uint32_t nup = p < nfreeze ? p : nfreeze; // wave-uniform
                    const uint32_t o = i * SB, d = o >> 5, rr = o & 31u;
                    uint32_t       sym;
                    if (rr + SB <= 32)
                        sym = (be[d] >> (32 - rr - SB)) & Tree::kMask;
                    else
                        sym = ((be[d] << (rr + SB - 32)) | (be[d + 1 < kW ? d + 1 : d] >> (64 - rr - SB))) & Tree::kMask;
                    uint32_t lo, hi;
                    part(sym, nup, p < nfreeze, lo, hi);
                    my[((t & 1u) * kHalf + i) * 64 + lane] = make_uint2(lo, hi);
                }
#pragma unroll
                for (int d = 0; d < kW; d++)
                    w[d] = wn[d];
                have_w = true;
                // (the per-symbol loader below continues behind this half)
                k_next   = (t + 1) * kHalf < last_sym ? (t + 1) * kHalf : last_sym;
                raw_next = gen_symbol_load<SB>(psrc, k_next, lastb);
            } else {
                have_w = false;
#pragma unroll
                for (uint32_t i = 0; i < kHalf; i++) {
                    const uint32_t p   = t * kHalf + i;
                    const uint32_t nup = p < nfreeze ? p : nfreeze; // wave-uniform
                    const uint32_t sym = gen_symbol_decode<SB>(raw_next, k_next);
                    {
                        const uint32_t q = p + 1 < last_sym ? p + 1 : last_sym;
                        k_next           = can_load ? q : 0u;
                        raw_next         = gen_symbol_load<SB>(psrc, k_next, lastb);
                    }
                    uint32_t lo = 0, hi = 0;
                    if (live && p < nsym)
                        part(sym, nup, p < nfreeze, lo, hi);
                    my[((t & 1u) * kHalf + i) * 64 + lane] = make_uint2(lo, hi);
                }
            }
            pair_barrier();
        }
        return;
    }
    // ---------------- coder wave ----------------
    uint8_t       *wdst  = a.slots + blk0 * a.slot_bytes;
    const uint32_t off0  = live ? lane * (uint32_t)a.slot_bytes : 0u; // (dead lanes store nothing: everything below is predicated)
    const uint32_t limit = off0 + a.slot_cap;
    const uint32_t sh    = 32 - a.code_bits;
    const rc_ptr   rc    = (rc_ptr)a.rc;
    EncState S;
    enc_init(S, off0);
    for (uint32_t t = 0; t < nper; t++) {
        pair_barrier();
        uint2 lh[kHalf];
#pragma unroll
        for (uint32_t i = 0; i < kHalf; i++) {
            const uint32_t at = ((t & 1u) * kHalf + i) * 64 + lane;
            const uint2    p0 = ring[0][at], p1 = ring[1][at], p2 = ring[2][at];
            lh[i] = make_uint2(p0.x + p1.x + p2.x, p0.y + p1.y + p2.y);
        }
        // (count <= 2^SB + 1 + 65535 < 2^17 for the u16 trees: the quotients need no fix-up, scale_div)
        constexpr bool kFix = (1u << SB) + 1u + Tree::kMaxSymbols >= (1u << 17);
        // a half in which every lane codes a data symbol, with room for a dword per symbol: straight-line, one store site,
        // the long-pending-run case found by a wave-level ballot (encode_symbol_fast) instead of the predicated encode_symbol
        if (whole && (t + 1) * kHalf <= minsym && __builtin_amdgcn_ballot_w64(S.off + kHalf * 4 + 32 > limit) == 0) {
#pragma unroll
            for (uint32_t i = 0; i < kHalf; i++) {
                const uint32_t p   = t * kHalf + i;
                const uint32_t nup = p < nfreeze ? p : nfreeze;
                encode_symbol_fast<kFix>(S, lh[i].x, lh[i].y, kCount0 + nup, rc[nup], sh, wdst);
            }
            continue;
        }
#pragma unroll
        for (uint32_t i = 0; i < kHalf; i++) {
            const uint32_t p   = t * kHalf + i;
            const uint32_t nup = p < nfreeze ? p : nfreeze;
            const double   r   = rc[nup];
            const uint32_t c   = kCount0 + nup;
            const uint2    e   = lh[i];
            if (live && p < nsym) {
                encode_symbol<kFix>(S, e.x, e.y, c, r, sh, false, wdst, limit);
            } else if (live && p == nsym) {
                // EOF symbol (codec.rs:108): cum(2^SB) = count - 1, cum(2^SB + 1) = count
                const uint32_t shifts = encode_symbol<kFix>(S, c - 1, c, c, r, sh, true, wdst, limit);
                const uint32_t size   = encode_finish(S, shifts, a.code_bits, off0, wdst, limit);
                a.sizes[blk]  = size;
                a.status[blk] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
            }
        }
    }
}

struct GenDecArgs {
    const uint8_t  *in;
    const uint64_t *in_offsets; // nblocks + 1
    uint64_t        nblocks;
    uint8_t        *out;        // block b at out + b*block_size
    uint32_t       *out_sizes;
    int32_t        *status;
    const double   *rc;
    uint32_t       *trees;      // !INLDS: kRows * 64 u32 per wave, zero at launch
    uint64_t       *in_used;    // optional
    uint32_t        block_size;
    uint32_t        nfreeze;
    uint32_t        code_bits;
};

template <int SB, bool INLDS = true>
__global__ void __launch_bounds__(64) k_decode_gen(GenDecArgs a)
{
    typedef GenTree<SB, INLDS> Tree;
    __shared__ uint32_t lds[Tree::kLdsDwords];
    const uint32_t lane = threadIdx.x;
    const uint64_t blk  = (uint64_t)blockIdx.x * Tree::kBlocks + lane;
    const bool     live = lane < Tree::kBlocks && blk < a.nblocks;
    if (INLDS) {
        for (uint32_t i = lane; i < Tree::kLdsDwords / 4; i += 64)
            reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    Tree T;
    T.init(INLDS ? lds : a.trees + (uint64_t)blockIdx.x * Tree::kDwords, lane);

    const uint32_t cb = a.code_bits, sh = 32 - cb;
    uint64_t       size = 0;
    const uint8_t *sp   = a.in;
    if (live) {
        const uint64_t o0 = a.in_offsets[blk];
        size              = a.in_offsets[blk + 1] - o0;
        sp                = a.in + o0;
    }
    const uint64_t stream_bits = size * 8;
    uint8_t       *dst         = a.out + (live ? blk : 0) * (uint64_t)a.block_size;
    const uint32_t capn        = a.block_size; // bytes
    const rc_ptr   rcp         = (rc_ptr)a.rc;
    constexpr uint32_t kCount0 = (1u << SB) + 1u;

    BitIn B;
    B.init(sp, live ? size : 0);
    uint32_t W        = B.take(cb) << sh; // codec.rs:124-127
    uint64_t consumed = cb;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    int32_t  st   = REDUX_OK;
    bool     done = !live;
    if (live && consumed > stream_bits) {
        st   = REDUX_EOF;
        done = true;
    }
    uint64_t obits = 0; // bits handed to write_bits so far (bitio/mod.rs:148-181): bytes [0, obits / 8) are in dst
    uint32_t oacc  = 0; // the incomplete byte's bits, right-aligned
    for (uint32_t p = 0;; p++) {
        if (__builtin_amdgcn_readfirstlane(__ballot(!done) == 0))
            break;
        const uint32_t nup = p < a.nfreeze ? p : a.nfreeze;
        const double   rc  = rcp[nup];
        const uint32_t c   = kCount0 + nup;
        if (!done) {
            // value = ((pending - low + 1) * count - 1) / range      (codec.rs:129-131)
            const uint32_t R1  = (high - low) >> sh;
            const uint32_t Vd  = (W - low) >> sh;
            const uint64_t num = ((uint64_t)Vd + 1) * c - 1;
            const double   xd  = (double)R1 + 1.0;
            uint32_t       v   = (uint32_t)((double)num / xd);
            {
                const int64_t r = (int64_t)(num - ((uint64_t)v * R1 + v));
                if (r < 0)
                    v--;
                else if ((uint64_t)r > (uint64_t)R1)
                    v++;
            }
            uint32_t lo, hi, s = 0;
            bool     is_eof = false;
            if (v >= c - 1) { // first probe of get_symbol: tree[2^SB] = 2^SB + #updates = count - 1 (adaptive_tree.rs:116)
                is_eof = true;
                lo     = c - 1;
                hi     = c;
            } else {
                uint32_t x[SB], ea[SB];
                uint32_t i = 0, rem = v;
#pragma unroll
                for (int b = SB - 1; b >= 0; b--) {
                    ea[b] = i | (1u << b);
                    x[b]  = T.load(ea[b]);
                    const uint32_t tv = (1u << b) + x[b];
                    if (rem >= tv) {
                        i |= 1u << b;
                        rem -= tv;
                    }
                }
                s  = i;
                lo = v - rem;
                const uint32_t m  = s + 1;
                uint32_t       hs = m;
#pragma unroll
                for (int b = 0; b < SB; b++)
                    hs += ((m >> b) & 1u) ? x[b] : 0u;
                hi = hs + (s == Tree::kMask ? nup : 0u);
                if (p < a.nfreeze) {
#pragma unroll
                    for (int b = 0; b < SB; b++)
                        if (!((s >> b) & 1u))
                            T.fetch_add(ea[b], 1u);
                }
            }
            if (is_eof) { // codec.rs:136-138: returns before any renormalisation
                done = true;
            } else {
                const double   Y     = __builtin_fma((double)R1, rc, rc);
                const uint32_t nlow  = low + (scale_div<true>(R1, Y, lo, c) << sh);
                const uint32_t nhigh = low + (scale_div<true, true>(R1, Y, hi, c) << sh) - 1u;
                const uint32_t xx    = nlow ^ nhigh;
                const uint32_t k     = xx ? (uint32_t)__builtin_clz(xx) : 32u;
                const uint32_t low2  = (uint32_t)((uint64_t)nlow << k);
                const uint32_t ih2   = (uint32_t)((uint64_t)(~nhigh) << k);
                const uint32_t t     = (low2 & ih2) << 1;
                const uint32_t j     = (uint32_t)__builtin_clz(~t);
                low                  = (low2 << j) & 0x7FFFFFFFu;
                high                 = ~((ih2 << j) & 0x7FFFFFFFu);
                const uint32_t n     = k + j; // bits pulled by get_bit (codec.rs:157)
                consumed += n;
                if (consumed > stream_bits) { // read_bits would hit Err(Eof) (bitio/mod.rs:107)
                    st   = REDUX_EOF;
                    done = true;
                } else {
                    const uint32_t nb   = B.take(n);
                    const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nb << (32 + sh - n));
                    const uint64_t c1   = comb << k;
                    const uint64_t c2   = c1 << j;
                    W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) &
                        (0xFFFFFFFFu << sh);
                    // write_bits(symbol, SB) (codec.rs:171): bytes leave as they complete; the first one past the
                    // block's capacity is where the writer fails
                    uint32_t acc  = (oacc << SB) | s;
                    uint32_t have = (uint32_t)(obits & 7u) + SB;
                    uint64_t pos  = obits >> 3;
                    while (have >= 8 && !done) {
                        if (pos >= capn) {
                            st   = REDUX_OUTPUT_TOO_SMALL;
                            done = true;
                        } else {
                            have -= 8;
                            dst[pos++] = (uint8_t)(acc >> have);
                        }
                    }
                    if (!done) {
                        oacc = acc & ((1u << have) - 1u);
                        obits += SB;
                    } else
                        obits = pos * 8; // the bytes before the failing one are written
                }
            }
        }
    }
    if (live) {
        a.out_sizes[blk] = (uint32_t)(obits >> 3); // a partial byte is never flushed (lib.rs:113-120)
        a.status[blk]    = st;
        if (a.in_used) { // the reader fetches whole bytes, and never past the end of the stream
            const uint64_t used = (consumed + 7) / 8;
            a.in_used[blk]      = used < size ? used : size;
        }
    }
}

} // namespace redux
