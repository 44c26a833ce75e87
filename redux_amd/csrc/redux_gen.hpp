// redux_gen.hpp -- the lock-step ENCODERS for the symbol widths other than 8 (symbol_bits 1 .. 12, code_bits <= 32; gfx950
// only): the widths src/model/tests.rs:95-251 exercises (4 and 12) and everything between.  SURVEY section 8(f).3.  Their
// decoder is k_decode_cells (redux_decode_cells.hpp).
//
// redux_any.hpp covers every Parameters triple as a one-lane-per-block rendering of the reference's loops.  This file is
// the MI355X form of compress_stream for these widths, built like the 8-bit kernels:
//   * one LANE per block, all lanes on the same symbol index, so the model's total (2^symbol_bits + 1 + symbols coded,
//     until the freq_max freeze: adaptive_tree.rs:84) is WAVE-UNIFORM and the two u64 divisions of codec.rs:59-60 are
//     multiplications by a per-step reciprocal (scale_div; with its quotient fix-up once the count can pass 2^17);
//   * the Fenwick tree as increments d[i] = tree[i] - lowbit(i), i = 1 .. 2^symbol_bits - 1, in LDS, in per-lane
//     columns.  symbol_bits <= 7: u32, row e of lane l at e * 256 + 4 l (4-bit: 4 KiB per wave of 64 blocks).  symbol_bits
//     >= 9: the LDS, not the wave width, sets how many blocks a CU holds: u16 nodes (at most 65535 updates of the model),
//     two lanes to a dword, 64 / 64 / 32 / 16 blocks per workgroup for 9 / 10 / 11 / 12 bits (128 KiB for 12: one
//     workgroup per CU), coded by FOUR waves (k_encode_gen_pair);
//   * get_frequency = one fetch-add per level (addend 1 where update(s+1) increments the node, 0 where the prefix
//     sums only read it) + two masked sums (adaptive_tree.rs:63-92), closed-form renormalisation and bit output
//     exactly as encode_symbol (redux_coder.hpp);
//   * symbols are read_bits(symbol_bits) MSB-first (bitio/mod.rs:78-120); a trailing partial symbol is dropped and the
//     block ends (Err(Eof) -> EOF symbol, codec.rs:108);
//   * whole waves run unpredicated: k_encode_gen in turns of 32 symbols (= symbol_bits dwords of input, loaded a turn
//     ahead), k_encode_gen_pair's coder wave in halves of eight, its model waves on a half's symbol_bits bytes loaded as
//     dwords a half ahead; ragged waves, the last symbols of a block and the EOF symbol take per-lane predicated loops.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"

#include <type_traits>
#include "redux_decode.hpp" // BitIn
#include "redux_encode.hpp" // wave_max

#include "../../include/redux_hip.h"

namespace redux {

// The trees of a wave's blocks in LDS (INLDS; the other form -- u32 trees in the workspace, walked with global fetch-adds --
// served round 3's per-level decoder and is kept for experiments only).
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
        constexpr uint32_t kNoFix = (1u << 17) - kCount0; // count < 2^17 while p < kNoFix, or for good if the model freezes below
        const uint32_t e      = minsym;
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
            auto turns = [&](auto frozen_tag, auto fix_tag, const uint32_t pend) {
            constexpr bool FRZ = decltype(frozen_tag)::value, FIX = decltype(fix_tag)::value; // FIX: the count may be 2^17 or more (scale_div's fix-up)
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
                        encode_symbol_fast<FIX>(S, lo, hi, kCount0 + nup, r[i], sh, wdst);
                    }
                }
#pragma unroll
                for (int d = 0; d < SB; d++)
                    cur[d] = nxt[d];
            }
            };
            // adaptive turns below 2^17, adaptive turns beyond (a block of more than 2^17 symbols whose model has not frozen),
            // then the turns from the freeze point on
            const uint32_t a_end = fast_end < (nfreeze & ~(U - 1u)) ? fast_end : (nfreeze & ~(U - 1u));
            const uint32_t n_end = a_end < (kNoFix & ~(U - 1u)) ? a_end : (kNoFix & ~(U - 1u));
            turns(std::false_type(), std::false_type(), n_end);
            if (p == n_end && n_end < a_end) // (not left early for a full slot)
                turns(std::false_type(), std::true_type(), a_end);
            if (p == a_end && a_end < fast_end && nfreeze < kNoFix) // (a model frozen at 2^17 or more: the loop below)
                turns(std::true_type(), std::false_type(), fast_end);
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
    uint32_t       *trees;      // k_decode_cells<.., GLOBAL0>: the blocks' bottom cells (k_fill_cells16)
    uint64_t       *in_used;    // optional
    uint32_t        block_size;
    uint32_t        nfreeze;
    uint32_t        code_bits;
};

// (the decoder of these widths is k_decode_cells, redux_decode_cells.hpp; rounds 2 and 3 had a per-level walk here, k_decode_gen)

} // namespace redux
