// redux_decode_wave.hpp -- k_decode_wave: one block per WAVE, the model in registers across the lanes (gfx950 only).
//
// The lock-step decoder (redux_decode_adaptive.hpp) runs 64 blocks per wave, each lane descending its own Fenwick tree of
// u16 nodes; a block it cannot take -- u32 counts, count >= 2^17: above all ONE block of any length, redux_decompress of
// a whole stream, the literal redux::decompress -- used to fall to k_decode: eight dependent LDS probes and a true
// division per symbol on one lane, ~1 MB/s.  A launch that leaves most SIMDs idle can afford a wave per block, and then
// get_symbol / update (adaptive_tree.rs:115-136, :83-92) are wave operations on the PLAIN cumulative table:
//   * lane l holds cum(l + 1), cum(l + 65), cum(l + 129), cum(l + 193) -- the frequencies' inclusive prefix sums -- as
//     u32 (any block length); the plane tops cum(64), cum(128), cum(192) are kept wave-uniform as well;
//   * get_symbol(v): three scalar compares pick the plane, ONE v_cmp + ballot + popcount finds the lane where the table
//     passes v; cum(s) and cum(s + 1) are two v_readlane;
//   * update(s): every table entry above s grows by one: a compare + add-with-carry per plane, no memory at all;
//   * the stream lives in a register across the lanes too (WaveBits): a refill is a v_readlane.
// Everything else is k_decode's (the same order of the Eof / capacity checks, codec.rs:123-176), computed by the 64 lanes
// on identical values; the three ways a block ends leave the loop by one branch.  1 MiB as one stream: 545 ms = 1.9 MB/s
// (k_decode: ~1.1).  The step is ~200 instructions, most of them scalar -- a lone wave pays an issue slot for those too --
// and has not been worked down as k_decode_lock's has: for blocks the lock-step decoder takes it is the slower one
// (28.9 against 24 ms per 64 KiB block), so it is chosen only for the others, in launches of at most kWaveDecMaxBlocks.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_decode.hpp"

namespace redux {

constexpr uint64_t kWaveDecMaxBlocks = 1024; // one wave per SIMD: beyond that the waves share SIMDs and the lock-step decoder wins

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
    const rc_ptr   rcp         = (rc_ptr)a.rc;
    const bool     aligned4    = a.aligned4 != 0 || ((((uintptr_t)dst) & 3) == 0);

    // the model: inclusive prefix sums of the 256 data symbols' frequencies (all 1 at the start); EOF sits above them
    uint32_t C0 = lane + 1, C1 = lane + 65, C2 = lane + 129, C3 = lane + 193;
    uint32_t T0 = 64, T1 = 128, T2 = 192; // cum(64), cum(128), cum(192): wave-uniform

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
    uint32_t obuf  = 0;

    // One way through the step: everything is computed and the three ways a block ends -- the EOF symbol (codec.rs:136-138,
    // decided first: decompress_symbol returns before renormalising), a stream that runs dry in the renormalisation
    // (bitio/mod.rs:107), no room for the decoded symbol (codec.rs:171) -- leave the loop by ONE rarely taken branch; a taken
    // branch costs a lone wave ~35 cycles and every scalar instruction an issue slot, so the loop has no other exits.  (An
    // EOF step's garbage is harmless: its "symbol" is 256, which the update ignores, and nothing of it is committed.)
    double   rc_next = rcp[0]; // the reciprocal of step p + 1 is loaded during step p (a lone wave hides no latency by itself)
    uint32_t p       = 0;
    bool     eof = false, dry = false;
    uint64_t cons2 = consumed;
    if (!done) {
        for (;; p++) {
            const uint32_t nup = p < a.nfreeze ? p : a.nfreeze;
            const double   rc  = rc_next;
            rc_next            = rc_lookup(rcp, a.rc_n, p + 1 < a.nfreeze ? p + 1 : a.nfreeze, 257u); // (computed past the table's window)
            const uint32_t c   = 257u + nup;
            // value = ((pending - low + 1) * count - 1) / range      (codec.rs:129-131)
            const uint32_t R1 = (high - low) >> sh;
            const uint32_t Vd = (W - low) >> sh;
            uint32_t       v;
            if (FIXUP) {
                const uint64_t num = ((uint64_t)Vd + 1) * c - 1;
                const double   xd  = (double)R1 + 1.0;
                v                  = (uint32_t)((double)num / xd);
                const int64_t r    = (int64_t)(num - ((uint64_t)v * R1 + v));
                v += r < 0 ? 0xFFFFFFFFu : ((uint64_t)r > (uint64_t)R1 ? 1u : 0u);
            } else {
                v = dec_value((double)R1, Vd, (double)c, (double)(c - 1u)); // (count < 2^17: the reciprocal form, redux_decode.hpp)
            }
            const uint32_t vu = __builtin_amdgcn_readfirstlane(v); // (the same in every lane: scalar from here on)
            eof               = vu >= c - 1;                          // tree[256] = count - 1: the EOF symbol
            // get_symbol (adaptive_tree.rs:115-136) on the plain table
            const uint32_t pq   = (vu >= T0 ? 1u : 0u) + (vu >= T1 ? 1u : 0u) + (vu >= T2 ? 1u : 0u);
            const uint32_t X    = pq == 0 ? C0 : pq == 1 ? C1 : pq == 2 ? C2 : C3;
            const uint32_t b    = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(X <= vu)); // < 64 unless EOF: the plane's top is above v
            const uint32_t s    = 64u * pq + b;
            const uint32_t hi   = __builtin_amdgcn_readlane(X, b & 63u);
            const uint32_t base = pq == 0 ? 0u : pq == 1 ? T0 : pq == 2 ? T1 : T2;
            const uint32_t lo   = b ? __builtin_amdgcn_readlane(X, (b - 1u) & 63u) : base;
            if (p < a.nfreeze) { // update(s + 1), adaptive_tree.rs:83-92: every prefix sum above s grows by one
                C0 += lane >= s ? 1u : 0u;
                C1 += lane + 64u >= s ? 1u : 0u;
                C2 += lane + 128u >= s ? 1u : 0u;
                C3 += lane + 192u >= s ? 1u : 0u;
                T0 += s < 64u ? 1u : 0u;
                T1 += s < 128u ? 1u : 0u;
                T2 += s < 192u ? 1u : 0u;
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
            const uint32_t n     = k + j; // bits pulled by get_bit (codec.rs:157)
            cons2                = consumed + n;
            dry                  = cons2 > stream_bits; // read_bits would hit Err(Eof) (bitio/mod.rs:107)
            if (__builtin_expect(eof || dry || p >= capn, 0))
                break;
            consumed = cons2;
            low      = (low2 << j) & 0x7FFFFFFFu;
            high     = ~((ih2 << j) & 0x7FFFFFFFu);
            // k E1/E2 steps shift the value left, each of the j E3 steps drops the bit below the top one (k_decode)
            const uint32_t nb   = B.take(n);
            const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nb << (32 + sh - n));
            const uint64_t c1   = comb << k;
            const uint64_t c2   = c1 << j;
            W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) & (0xFFFFFFFFu << sh);
            // emit the symbol (write_bits(symbol, 8), codec.rs:171); every lane stores the same dword to the same address
            // (no exec mask, no branch around it: the requests merge)
            if (aligned4) {
                obuf |= s << (8 * (p & 3));
                if ((p & 3) == 3) {
                    *reinterpret_cast<uint32_t *>(dst + (p & ~3u)) = obuf;
                    obuf = 0;
                }
            } else {
                dst[p] = (uint8_t)s;
            }
        }
        if (!eof) { // the renormalisation was entered: its bits count as consumed whichever way the step failed
            consumed = cons2;
            st       = dry ? REDUX_EOF : REDUX_OUTPUT_TOO_SMALL;
        }
    }
    n_out = p; // every committed step emitted one symbol
    if (lane == 0) {
        if (aligned4)
            for (uint32_t i = n_out & ~3u; i < n_out; i++)
                dst[i] = (uint8_t)(obuf >> (8 * (i & 3)));
        a.out_sizes[blk] = n_out;
        a.status[blk]    = st;
        if (a.in_used) { // the reader fetches whole bytes, and never past the end of the stream
            const uint64_t used = ((uint64_t)consumed + 7) / 8;
            a.in_used[blk]      = used < size ? used : size;
        }
    }
}

} // namespace redux
