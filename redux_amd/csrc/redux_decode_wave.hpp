// redux_decode_wave.hpp -- k_decode_wave: one block per WAVE, the model in registers across the lanes (gfx950 only).
//
// The lock-step decoder (redux_decode_adaptive.hpp) runs 64 blocks per wave, each lane descending its own Fenwick tree of
// u16 nodes; a block it cannot take -- u32 counts, count >= 2^17: above all ONE block of any length, redux_decompress of
// a whole stream, the literal redux::decompress -- used to fall to k_decode: eight dependent LDS probes and a true
// division per symbol on one lane, ~1 MB/s.  A launch that leaves most SIMDs idle can afford a wave per block, and then
// get_symbol / update (adaptive_tree.rs:115-136, :83-92) are wave operations on the PLAIN cumulative table:
//   * lane l holds cum(4l + 1) .. cum(4l + 4) -- the frequencies' inclusive prefix sums of FOUR CONSECUTIVE symbols -- as
//     u32 (any block length);
//   * get_symbol(v): the symbol is the number of table entries that do not exceed v: four v_cmp + four scalar popcounts,
//     no plane to select and no plane tops to maintain (round 3 kept the table as four planes of 64 and picked the plane
//     with scalar compares: ~25 scalar instructions a step); cum(s) and cum(s + 1) are picked per lane by the same four
//     compare masks and fetched with three v_readlane;
//   * update(s): every table entry above s grows by one: a compare + add-with-carry per register, no memory at all;
//   * the stream lives in a register across the lanes too (WaveBits): a refill is a v_readlane.
// Everything else is k_decode's (the same order of the Eof / capacity checks, codec.rs:123-176), computed by the 64 lanes
// on identical values; the three ways a block ends leave the loop by one branch.  Most of the step is scalar -- a lone wave
// pays an issue slot for those instructions too -- so for blocks the lock-step decoder takes this one is not the faster
// one, and it is chosen only for the others, in launches of at most kWaveDecMaxBlocks.  A serial adaptive stream has no
// other parallelism: one CPU thread decodes ~19 MB/s, this kernel a tenth of that (INTEGRATION.md says so next to the
// numbers).
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_decode.hpp"

namespace redux {

#ifndef REDUX_WAVE_DEC_MAX_BLOCKS // (A/B builds set it)
#define REDUX_WAVE_DEC_MAX_BLOCKS 1024
#endif
// ... except launches of more than kWaveDecManyBlocks blocks of kWaveDecLargeBlock bytes or more: 1024 waves over streams of
// 1 MiB and more run at 547 ns per symbol (768: 415; 1024 x 128 KiB: 401), k_decode_cells<8> at 479 (profiles/r04_wave_decoder.txt)
constexpr uint64_t kWaveDecManyBlocks = 768;
constexpr uint32_t kWaveDecLargeBlock = 1u << 20;
constexpr uint64_t kWaveDecMaxBlocks = REDUX_WAVE_DEC_MAX_BLOCKS; // one wave per SIMD: beyond that the waves share SIMDs and the lock-step decoder wins

// The bit reader of a wave that decodes ONE stream (bitio/mod.rs:78-120): the stream comes in rows of 64 dwords -- lane l
// loads dword l of the row, one coalesced request -- and lives in a register ACROSS the lanes; the next dword for the
// 64-bit buffer is a v_readlane with a wave-uniform index.  The row after the current one is already loaded, so a refill
// never waits for memory.  (BitIn's one-dword-ahead load is copied at a branch join,
// which makes the compiler wait for it at once: ~1 us per refill on a lone wave.)  Dwords past the stream's end read as
// zero; running past the end is detected by the caller's consumed-bit count, as with BitIn.
struct WaveBits {
    uint64_t       bits;
    uint32_t       cnt;
    uint32_t       d, nd;   // next dword / number of dwords, counted from base16
    uint32_t       lane;
    const uint8_t *base16;
    uint32_t       P;       // the 64 dwords d & ~63 .. : lane l holds dword (d & ~63) + l
    uint32_t       N;       // the 64 dwords after those, already loaded

    __device__ __forceinline__ uint32_t load_row(uint32_t first) const // dword first + lane of the stream, zero past its end
    {
        const uint32_t i = first + lane;
        return i < nd ? reinterpret_cast<const uint32_t *>(base16)[i] : 0u;
    }
    __device__ __forceinline__ uint32_t fetch()
    {
        const uint32_t w = __builtin_amdgcn_readlane(P, d & 63u);
        d++;
        if ((d & 63u) == 0) { // (wave-uniform, once per 64 dwords) the row is used up: the loaded one takes its place, the one after it is requested
            P = N;
            N = load_row(d + 64u);
        }
        return w;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            bits |= (uint64_t)__builtin_bswap32(fetch()) << (32 - cnt);
            cnt += 32;
        }
    }
    __device__ __forceinline__ void init(const uint8_t *sp, uint64_t size, uint32_t lane_)
    {
        lane   = lane_;
        // rows of 64 dwords = 256 bytes, from the 256-byte boundary below the stream (the bytes in front of it belong to
        // the same buffer -- the previous block's stream -- or, for the first block of a buffer that is not 256-byte
        // aligned itself, to the same page)
        base16 = reinterpret_cast<const uint8_t *>((uintptr_t)sp & ~(uintptr_t)255);
        const uint32_t lead = (uint32_t)((uintptr_t)sp & 255);
        nd     = (uint32_t)((lead + size + 3) >> 2);
        d      = lead >> 2;
        P      = load_row(0);
        N      = load_row(64);
        bits   = 0;
        cnt    = 0;
        const uint32_t skip = (lead & 3u) * 8;
        refill();
        bits <<= skip;
        cnt -= skip;
        refill();
    }
    // drop the next n (<= 32) bits; at least 33 valid bits are left in front afterwards
    __device__ __forceinline__ void skip(uint32_t n)
    {
        bits <<= n;
        cnt -= n;
        refill();
    }
    // next n (<= 32) bits, MSB first
    __device__ __forceinline__ uint32_t take(uint32_t n)
    {
        const uint32_t v = (uint32_t)((bits >> 1) >> (63 - n));
        bits <<= n;
        cnt -= n;
        refill();
        return v;
    }
};

template <bool FIXUP>
__global__ void __launch_bounds__(64) k_decode_wave(DecArgs a)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t slot = blockIdx.x;
    if (slot >= a.nblocks)
        return;
    uint64_t blk = slot, dst_off = slot * (uint64_t)a.block_size;
    uint32_t capn = a.block_size;
    if (a.table) { // block table: see DecArgs
        const redux_block e = a.table[slot];
        if (e.index == 0xFFFFFFFFu) // idle entry
            return;
        blk     = e.index;
        dst_off = e.offset;
        capn    = e.length;
    }
    const uint32_t cb = a.code_bits, sh = 32 - cb;
    const uint64_t o0          = a.in_offsets[blk];
    const uint64_t size        = a.in_offsets[blk + 1] - o0;
    const uint8_t *sp          = a.in + o0;
    const uint64_t stream_bits = size * 8;
    uint8_t       *dst         = a.out + dst_off;

    // the model: inclusive prefix sums of the 256 data symbols' frequencies (all 1 at the start), four consecutive
    // symbols to a lane; EOF sits above them
    const uint32_t L4 = 4u * lane, L4b = L4 + 1u, L4c = L4 + 2u, L4d = L4 + 3u; // the entries' numbers
    uint32_t C0 = L4 + 1, C1 = L4 + 2, C2 = L4 + 3, C3 = L4 + 4;

    WaveBits B;
    B.init(sp, size, lane);
    uint32_t W        = B.take(cb) << sh; // codec.rs:124-127
    uint64_t consumed = cb;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    int32_t  st   = REDUX_OK;
    bool     done = false;
    if (consumed > stream_bits) { // stream shorter than code_bits: Err(Eof) at once
        st   = REDUX_EOF;
        done = true;
    }
    uint32_t n_out = 0;

    // One way through the step: everything is computed and the three ways a block ends -- the EOF symbol (codec.rs:136-138,
    // decided first: decompress_symbol returns before renormalising), a stream that runs dry in the renormalisation
    // (bitio/mod.rs:107), no room for the decoded symbol (codec.rs:171) -- leave the loop by ONE rarely taken branch; a taken
    // branch costs a lone wave ~35 cycles and every scalar instruction an issue slot, so the loop has no other exits.  (An
    // EOF step's garbage is harmless: its "symbol" is 256, which the update ignores, and nothing of it is committed.)
    double   rcv     = 0.0;         // lane l holds the reciprocal of count 257 + rbase + l
    uint32_t rbase   = 0xFFFFFF00u; // (no such base: the first step computes its 64)
    uint32_t p       = 0;
    bool     eof = false, dry = false;
    // stream bits not pulled yet, as a 32-bit count-down (a block is below 4 GiB, so stream_bits < 2^35: the count-down is
    // clamped to 2^31 - 1 and topped up from the rest whenever it has dropped below 2^30; a step pulls at most 64 bits)
    uint64_t rest = stream_bits - (done ? stream_bits : consumed);
    uint32_t left = rest > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)rest;
    rest -= left;
    uint32_t n = 0;
    if (!done) {
        for (;; p++) {
            const uint32_t nup = p < a.nfreeze ? p : a.nfreeze;
            // This step's reciprocal (what k_fill_rc would hold for it: the correctly rounded 1 / count, biased up 4 ulp --
            // scale_div's proof needs a value in that range).  No table: lane l of rcv holds the reciprocal of count
            // 257 + rbase + l, all 64 computed by ONE division sequence every 64 steps, and a step picks its own with two
            // v_readlane -- no load in the step, nothing to wait for, the same cost at any block length (378 ns per symbol; a table of 2^20
            // entries read by a scalar load per step, with a division per step behind it: 408 inside the table, 431 over a 4 MiB block).
            if (__builtin_expect(nup - rbase >= 64u, 0)) { // (wave-uniform)
                rbase          = nup & ~63u;
                const double r = 1.0 / (double)(257u + rbase + lane);
                rcv            = __longlong_as_double(__double_as_longlong(r) + 4);
            }
            double rc;
            {
                const uint32_t           li  = nup - rbase;
                const unsigned long long rb  = (unsigned long long)__double_as_longlong(rcv);
                const uint32_t           lo_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rb, (int)li);
                const uint32_t           hi_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rb >> 32), (int)li);
                rc = __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));
            }
            const uint32_t c   = 257u + nup;
            // value = ((pending - low + 1) * count - 1) / range      (codec.rs:129-131)
            const uint32_t R1 = (high - low) >> sh;
            const uint32_t Vd = (W - low) >> sh;
            uint32_t       v;
            if (FIXUP) {
                // count up to 2^30: the numerator has up to 62 bits.  The quotient (< count) is estimated with a reciprocal
                // refined by one Newton step (relative error ~2^-48: the estimate is within 1) and settled by the exact
                // 64-bit remainder -- where a true f64 division would spend a dozen instructions to round a quotient that
                // is only an estimate anyway.
                const uint64_t num = ((uint64_t)Vd + 1) * c - 1;
                const double   xd  = (double)R1 + 1.0;
                double         ri  = __builtin_amdgcn_rcp(xd);
                ri                 = __builtin_fma(__builtin_fma(-xd, ri, 1.0), ri, ri);
                v                  = (uint32_t)((double)num * ri);
                const int64_t r    = (int64_t)(num - ((uint64_t)v * R1 + v));
                v += r < 0 ? 0xFFFFFFFFu : ((uint64_t)r > (uint64_t)R1 ? 1u : 0u);
            } else {
                v = dec_value((double)R1, Vd, (double)c, (double)(c - 1u)); // (count < 2^17: the reciprocal form, redux_decode.hpp)
            }
            const uint32_t vu = __builtin_amdgcn_readfirstlane(v); // (the same in every lane: scalar from here on)
            eof               = vu >= c - 1;                          // tree[256] = count - 1: the EOF symbol
            // get_symbol (adaptive_tree.rs:115-136) on the plain table: s = how many of the 256 prefix sums v has reached
            const bool     g0 = C0 <= vu, g1 = C1 <= vu, g2 = C2 <= vu, g3 = C3 <= vu;
            const uint32_t bl = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g3)); // lanes wholly at or below v
            const uint32_t s  = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g0)) +
                               (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g1)) +
                               (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g2)) + bl;
            // cum(s + 1) and cum(s) sit in lane bl (s = 4 bl + w): picked per lane by the same masks, fetched by readlane;
            // for w == 0 the low end is the lane before's last entry (0 in front of the table)
            const uint32_t hi_l = g0 ? (g1 ? (g2 ? C3 : C2) : C1) : C0;
            const uint32_t lo_l = g1 ? (g2 ? C2 : C1) : C0;
            const uint32_t hi   = __builtin_amdgcn_readlane(hi_l, bl & 63u);
            const uint32_t loin = __builtin_amdgcn_readlane(lo_l, bl & 63u);
            const uint32_t prev = __builtin_amdgcn_readlane(C3, (bl - 1u) & 63u);
            const uint32_t lo   = (s & 3u) ? loin : (bl ? prev : 0u);
            { // update(s + 1), adaptive_tree.rs:83-92: every prefix sum above s grows by one (not once the model is frozen).
              // The four compares first, then the four adds-with-carry: written as compare + add pairs the compiler puts a
              // wait state between each compare and the add that reads its mask.
                const uint32_t su = p < a.nfreeze ? s : 0xFFFFFFFFu;
                uint64_t       m0, m1, m2, m3;
                asm("v_cmp_le_u32_e64 %4, %8, %9\n\t"
                    "v_cmp_le_u32_e64 %5, %8, %10\n\t"
                    "v_cmp_le_u32_e64 %6, %8, %11\n\t"
                    "v_cmp_le_u32_e64 %7, %8, %12\n\t"
                    "v_addc_co_u32_e64 %0, %4, 0, %0, %4\n\t"
                    "v_addc_co_u32_e64 %1, %5, 0, %1, %5\n\t"
                    "v_addc_co_u32_e64 %2, %6, 0, %2, %6\n\t"
                    "v_addc_co_u32_e64 %3, %7, 0, %3, %7"
                    : "+v"(C0), "+v"(C1), "+v"(C2), "+v"(C3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                    : "s"(su), "v"(L4), "v"(L4b), "v"(L4c), "v"(L4d));
            }
            const double   Y     = __builtin_fma((double)R1, rc, rc);
            const uint32_t nlow  = low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
            const uint32_t nhigh = low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh) - 1u;
            const uint32_t xx    = nlow ^ nhigh;
            const uint32_t k     = xx ? (uint32_t)__builtin_clz(xx) : 32u;
            const uint32_t low2  = (uint32_t)((uint64_t)nlow << k);
            const uint32_t ih2   = (uint32_t)((uint64_t)(~nhigh) << k);
            const uint32_t t     = (low2 & ih2) << 1;
            const uint32_t j     = (uint32_t)__builtin_clz(~t);
            n                    = k + j; // bits pulled by get_bit (codec.rs:157)
            dry                  = n > left; // read_bits would hit Err(Eof) (bitio/mod.rs:107)
            if (__builtin_expect(eof || dry || p >= capn, 0))
                break;
            left -= n;
            if (__builtin_expect(rest != 0, 0)) { // (streams beyond 256 MiB only)
                if (left < 0x40000000u) {
                    const uint64_t add = rest > 0x3FFFFFFFull ? 0x3FFFFFFFull : rest;
                    left += (uint32_t)add;
                    rest -= add;
                }
            }
            low  = (low2 << j) & 0x7FFFFFFFu;
            high = ~((ih2 << j) & 0x7FFFFFFFu);
            // [value | next 32 stream bits] << k, keep the top bit, << j, put it back: k E1/E2 steps shift the value left, each
            // of the j E3 steps drops the bit below the top one (codec.rs:143-157; k_decode_lock's form: the bits are taken
            // from the reader's look-ahead in place, then the reader moves on)
            const uint32_t nxt  = (uint32_t)(B.bits >> 32);
            const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nxt << sh);
            const uint32_t h2   = (uint32_t)((comb << n) >> 32);
            const uint32_t h1   = (uint32_t)((comb << k) >> 32);
            W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
            B.skip(n);
            // emit the symbol (write_bits(symbol, 8), codec.rs:171): every lane stores the same byte to the same address (no
            // exec mask, no branch around it: the requests merge; a byte per ~700 cycles is nothing to the memory system.
            // Staging 64 symbols across the lanes and storing them as one run was built and measured: +1-2 % for one
            // stream and no gain for many, profiles/r04_wave_decoder.txt)
            dst[p] = (uint8_t)s;
        }
        // bits pulled when the loop was left: the steps before this one, plus -- unless it ended on the EOF symbol, which
        // returns before renormalising -- this step's own (its renormalisation was entered whichever way it failed)
        consumed = stream_bits - rest - left;
        if (!eof) {
            consumed += n;
            st = dry ? REDUX_EOF : REDUX_OUTPUT_TOO_SMALL;
        }
    }
    n_out = p; // every committed step emitted one symbol
    if (lane == 0) {
        a.out_sizes[blk] = n_out;
        a.status[blk]    = st;
        if (a.in_used) { // the reader fetches whole bytes, and never past the end of the stream
            const uint64_t used = ((uint64_t)consumed + 7) / 8;
            a.in_used[blk]      = used < size ? used : size;
        }
    }
}

} // namespace redux
