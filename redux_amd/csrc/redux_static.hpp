// redux_static.hpp -- the coder core under a STATIC frequency table (gfx950 only).
//
// the reference's Codec is generic over its Model trait
// (src/model/mod.rs: parameters / total_frequency / get_frequency / get_symbol, lib.rs:14-15), and
// the cheapest second model is a fixed table: cum[0..=257] with cum[0] = 0, cum strictly
// increasing, cum[257] = total <= freq_max.  Symbol s (0..255 data, 256 = EOF, model/mod.rs
// symbol_eof) owns [cum[s], cum[s+1]); nothing is ever updated.  Block b's stream is what
// Codec::compress_stream (codec.rs:104-120) writes for that block with such a model, and
// k_decode_static is Codec::decompress_stream (codec.rs:164-176) with get_symbol as a binary
// search of the table.  oracle/ has the same model (OX_MODEL_STATIC, StaticModel).
//
//   k_encode_static<FIXUP>   one lane per block, 64 blocks per wave, table in LDS (1 KiB)
//   k_decode_static<FIXUP>   the inverse, per-lane control flow (totals >= 2^17)
//   k_decode_static_lock     the inverse in lock-step form (totals < 2^17: the default)
//
// These share every building block with the adaptive kernels (EncState, encode_symbol,
// encode_finish, BitIn, scale_div); what they do not have is a tree, so a workgroup needs 1 KiB
// of LDS instead of 32 and a SIMD holds eight waves instead of one or two.  FIXUP as in
// scale_div: total >= 2^17.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"
#include "redux_decode.hpp" // BitIn
#include "redux_encode.hpp" // wave_min / wave_max

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

constexpr uint32_t kStaticEntries = 258; // cum[0..=257]

struct StaticTable { // passed by value in the kernel arguments (1032 bytes)
    uint32_t cum[kStaticEntries];
};

struct StaticEncArgs {
    const uint8_t *in;
    uint64_t       in_len;
    uint64_t       nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    double         rc;        // 1/total rounded, then bumped 4 ulp (as k_fill_rc)
    uint32_t       block_size;
    uint32_t       slot_cap;
    uint32_t       code_bits;
    uint32_t       aligned16; // in and block_size are 16-byte multiples
    StaticTable    tab;
};

// Sixteen data symbols straight-line, all 64 lanes active, stores unchecked (the caller has
// checked the chunk's budget): encode_symbol_spec as in the adaptive coder wave, with the same
// "redo the stretch from the saved state with the general encode_symbol if any lane needed more
// than one 32-bit append" rule.  The two table reads of a symbol are one ds_read2_b32.
template <bool FIXUP, bool CB32>
__device__ __forceinline__ void static_chunk(EncState &S, const uint32_t *tab, const uint4 cur, uint32_t c, double rc,
                                             uint32_t sh, uint8_t *wdst)
{
    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
    const EncState S0   = S;
    SpecCarry      C    = spec_begin(S);
    uint32_t       mx   = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const uint32_t m = encode_symbol_spec<FIXUP, CB32>(S, C, tab[s], tab[s + 1], c, rc, sh, wdst);
        mx               = m > mx ? m : mx;
    }
    spec_end(S, C);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > 32u) != 0, 0)) {
        S = S0;
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            const uint32_t s = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            encode_symbol<FIXUP>(S, tab[s], tab[s + 1], c, rc, sh, false, wdst, 0xFFFFFFFFu);
        }
    }
}

template <bool SOLO>
__device__ __forceinline__ void claim_the_simd();

template <bool FIXUP, bool CB32, bool SOLO = false>
__global__ void __launch_bounds__(64) k_encode_static(StaticEncArgs a)
{
    __shared__ uint32_t tab[kStaticEntries + 2];
    claim_the_simd<SOLO>();
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < kStaticEntries; i += 64)
        tab[i] = a.tab.cum[i];
    __syncthreads();
    const uint64_t blk0 = (uint64_t)blockIdx.x * 64;
    const uint64_t blk  = blk0 + lane;
    const bool     live = blk < a.nblocks;
    uint32_t       len  = 0;
    if (live) {
        const uint64_t rem = a.in_len - blk * a.block_size;
        len                = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    const uint8_t *src   = a.in + (live ? blk : blk0) * (uint64_t)a.block_size;
    uint8_t       *wdst  = a.slots + blk0 * a.slot_bytes;
    // dead lanes of the last wave own the spare slot behind the last real one (they store nothing)
    const uint32_t off0  = live ? lane * (uint32_t)a.slot_bytes : (uint32_t)(a.nblocks - blk0) * (uint32_t)a.slot_bytes;
    const uint32_t limit = off0 + a.slot_cap;
    const uint32_t maxlen = __builtin_amdgcn_readfirstlane(wave_max(live ? len : 0u));
    const uint32_t sh     = 32 - a.code_bits;
    const uint32_t c      = tab[kStaticEntries - 1]; // total_frequency()
    const double   rc     = a.rc;

    EncState S;
    enc_init(S, off0);
    uint32_t p = 0;
    // whole 16-byte chunks below the shortest live block's end: one 16-byte load per lane and
    // chunk, the next one in flight while this one is coded; all 64 lanes active
    const uint32_t minlen = __builtin_amdgcn_readfirstlane(wave_min(live ? len : 0xFFFFFFFFu));
    if (a.aligned16 && minlen != 0xFFFFFFFFu && minlen >= 32) {
        const uint32_t main_end = minlen & ~15u;
        // every lane reads its block one whole 128-byte line at a time (ChunkQueue, redux_encode.hpp):
        // with 16 bytes per visit the line is evicted between visits and fetched eight times
        ChunkQueue Q;
        Q.init(a.in + blk0 * (uint64_t)a.block_size, live ? lane * a.block_size : 0u, main_end);
        constexpr uint32_t kChunkBudget = 16 * 4 + 32; // bytes a chunk may add without a per-store check
        for (; p < main_end; p += 16) {
            const uint4 cur = Q.pop();
            if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit)) { // a slot is nearly full: every store checked
                const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    const uint32_t s = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                    encode_symbol<FIXUP>(S, tab[s], tab[s + 1], c, rc, sh, false, wdst, limit);
                }
            } else
                static_chunk<FIXUP, CB32>(S, tab, cur, c, rc, sh, wdst);
        }
    }
    for (; p <= maxlen; p++) {
        if (live && p < len) {
            const uint32_t s = src[p];
            encode_symbol<FIXUP>(S, tab[s], tab[s + 1], c, rc, sh, false, wdst, limit); // get_frequency(s)
        } else if (live && p == len) {
            // EOF symbol (codec.rs:108): [cum[256], total)
            const uint32_t shifts = encode_symbol<FIXUP>(S, tab[256], c, c, rc, sh, true, wdst, limit);
            const uint32_t size   = encode_finish(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[blk]  = size;
            a.status[blk] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
}

struct StaticDecArgs {
    const uint8_t  *in;
    const uint64_t *in_offsets; // nblocks + 1
    uint64_t        nblocks;
    uint8_t        *out;        // block b at out + b*block_size
    uint32_t       *out_sizes;
    int32_t        *status;
    double          rc;
    uint32_t        block_size;
    uint32_t        code_bits;
    uint32_t        aligned4; // out and block_size are 4-byte multiples
    StaticTable     tab;
};

template <bool FIXUP>
__global__ void __launch_bounds__(64) k_decode_static(StaticDecArgs a)
{
    __shared__ uint32_t tab[kStaticEntries + 2];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < kStaticEntries; i += 64)
        tab[i] = a.tab.cum[i];
    __syncthreads();
    const uint64_t blk  = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = blk < a.nblocks;
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
    const uint32_t capn        = a.block_size;
    const uint32_t c           = tab[kStaticEntries - 1];
    const double   rc          = a.rc;

    BitIn B;
    B.init(sp, live ? size : 0);
    // decompress_symbol's first call pulls code_bits bits (codec.rs:124-127)
    uint32_t W        = B.take(cb) << sh;
    uint64_t consumed = cb;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    int32_t  st   = REDUX_OK;
    bool     done = !live;
    if (live && consumed > stream_bits) { // stream shorter than code_bits: Err(Eof) at once
        st   = REDUX_EOF;
        done = true;
    }
    uint32_t n_out = 0, obuf = 0;
    for (uint32_t p = 0;; p++) {
        if (__builtin_amdgcn_readfirstlane(__ballot(!done) == 0))
            break;
        if (done)
            continue;
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
        // get_symbol: the s in 0..256 with cum[s] <= v < cum[s+1] (v < total)
        uint32_t s = 0;
#pragma unroll
        for (int b = 8; b >= 0; b--) {
            const uint32_t t = s | (1u << b);
            if (t <= 256u && tab[t] <= v)
                s = t;
        }
        if (s == 256u) { // codec.rs:136-138: EOF returns before any renormalisation
            done = true;
            continue;
        }
        const uint32_t lo = tab[s], hi = tab[s + 1];
        const double   Y     = __builtin_fma((double)R1, rc, rc);
        const uint32_t nlow  = low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
        const uint32_t nhigh = low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh) - 1u;
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
            continue;
        }
        if (p >= capn) { // the symbol is decoded; writing it is what fails (codec.rs:171)
            st   = REDUX_OUTPUT_TOO_SMALL;
            done = true;
            continue;
        }
        // E1/E2 shift the value by k; each of the j E3 steps drops the bit below the top one
        const uint32_t nb   = B.take(n);
        const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nb << (32 + sh - n));
        const uint64_t c1   = comb << k;
        const uint64_t c2   = c1 << j;
        W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) & (0xFFFFFFFFu << sh);
        // write_bits(symbol, 8), codec.rs:171: four symbols per store where the layout allows
        if (a.aligned4) {
            obuf |= s << (8 * (p & 3));
            if ((p & 3) == 3) {
                *reinterpret_cast<uint32_t *>(dst + (p & ~3u)) = obuf;
                obuf = 0;
            }
        } else
            dst[p] = (uint8_t)s;
        n_out = p + 1;
    }
    if (live) {
        if (a.aligned4)
            for (uint32_t i = n_out & ~3u; i < n_out; i++)
                dst[i] = (uint8_t)(obuf >> (8 * (i & 3)));
        a.out_sizes[blk] = n_out;
        a.status[blk]    = st;
    }
}

// The static decoder in k_decode_lock's form (redux_decode.hpp, decode_lock_body<CB32, true>): all 64 lanes in
// lock-step, the table as its Fenwick form in LDS (1 KiB per workgroup, shared by the lanes), get_symbol as the
// same carry-driven descent with speculative loads (two LDS round trips instead of nine dependent ones), stream
// ring, staged 16-byte output stores, one ballot per step.  For total < 2^17 (no quotient fix-up, as the adaptive
// lock-step decoder); larger totals keep k_decode_static<true>.
struct StaticLockArgs {
    DecArgs     d;
    double      rc;
    StaticTable tab;
};

// SOLO: the kernel claims more than half of a SIMD's 512 registers, so that no two of its waves share a SIMD.
// A lock-step wave that has a SIMD to itself finishes a 64 KiB block in ~20 ms; the dispatcher, free to pack
// (81 registers, 9 KiB of LDS), puts two waves on some SIMDs and none on others when the grid is only four waves
// per CU, and the kernel then lasts as long as the doubled-up ones (32.3 ms; SQ_WAVE_CYCLES says the average wave
// lived 0.74 of that).  k_decode_lock is safe from this by accident of its register allocation (257).  Used for
// grids of at most one wave per SIMD; larger grids want several waves per SIMD (4: 237 GB/s at 16 KiB blocks).
template <bool SOLO>
__device__ __forceinline__ void claim_the_simd()
{
    if (SOLO)
        asm volatile("" ::: "a255");
}

template <bool CB32, bool SOLO>
__global__ void __launch_bounds__(64) k_decode_static_lock(StaticLockArgs a)
{
    __shared__ uint32_t lds[kStaticTreeDwords + 32 * 64]; // Fenwick form of the table (~1 KiB, padded) + stream ring (8 KiB)
    claim_the_simd<SOLO>();
    decode_lock_body<CB32, 1>(a.d, lds, a.tab.cum, a.rc);
}

// Totals up to 2^16: get_symbol by direct lookup (dec_search_lut).  WAVES waves share one 64 KiB byte table lut[v] =
// symbol and the plain cumulative table; each has its own 8 KiB stream ring: 4 waves = 97 KiB, one workgroup and one wave
// per SIMD on a CU (the headline shape: 1024 groups of 64 blocks), 8 waves = 129 KiB for grids beyond that.
template <bool CB32, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_decode_static_lut(StaticLockArgs a)
{
    constexpr uint32_t kLutDwords = 65536 / 4, kTabDwords = 260;
    __shared__ uint32_t lds[kLutDwords + kTabDwords + WAVES * 32 * 64];
    uint8_t  *lut  = reinterpret_cast<uint8_t *>(lds);
    uint32_t *ctab = lds + kLutDwords;
    const uint32_t t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (uint32_t i = t; i < kStaticEntries; i += 64 * WAVES)
        ctab[i] = a.tab.cum[i];
    // symbol s fills [cum[s], cum[s+1]); the EOF symbol's range (and nothing beyond the total is ever looked up) gets 255
    for (uint32_t s = t; s < 257; s += 64 * WAVES) {
        const uint32_t b = a.tab.cum[s], e = a.tab.cum[s + 1] < 65536u ? a.tab.cum[s + 1] : 65536u;
        for (uint32_t i = b; i < e; i++)
            lut[i] = (uint8_t)(s < 256 ? s : 255);
    }
    __syncthreads();
    decode_lock_body<CB32, 2>(a.d, ctab + kTabDwords + wave * (32 * 64), a.tab.cum, a.rc, t & 63u,
                              (uint64_t)blockIdx.x * WAVES + wave, lut, ctab);
}

} // namespace redux
