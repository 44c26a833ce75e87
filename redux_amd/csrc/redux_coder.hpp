// redux_coder.hpp -- device-side building blocks of the MI355X block coder (gfx950 only).
//
// One LANE of a 64-wide wavefront owns one block: its interval state lives in registers, its
// 255-node cumulative-frequency tree lives in LDS, and all 64 lanes advance one symbol per
// step in lock-step.  Because every lane has coded the same number of symbols, the model's
// total frequency (257 + symbols coded, until the freq_max freeze) is WAVE-UNIFORM, which
// turns the two u64 divisions of codec.rs:59-60 into multiplications by one per-step
// reciprocal.
//
// What must equal the reference bit for bit (file:line under the reference checkout):
//   model   src/model/adaptive_tree.rs:36-136   (cumulative frequencies + freeze rule)
//   coder   src/codec.rs:55-101, :123-161        (narrowing, E1/E2/E3 renormalisation, EOF tail)
//   bit I/O src/bitio/mod.rs:148-198, :78-120    (MSB-first packing, zero padding)
// HOW it is computed is different everywhere; each routine below states the identity it
// relies on.
//
// Issue economics that shape the code: the tree (512 B per block) caps residency at 4-5
// waves per CU, i.e. ONE wave per SIMD.  A lone wave issues one instruction -- VALU or SALU
// -- per ~4 cycles with nothing to hide behind, so the hot path minimises instruction COUNT
// and avoids branches.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) double *rc_ptr; // forces scalar (SMEM) loads

// --------------------------------------------------------------------------------------
// Frequency tree in LDS.
//
// The reference keeps a Fenwick array tree[0..257] of u64 with tree[i] initialised to
// lowbit(i) (adaptive_tree.rs:43-45).  Here node i stores only d[i] = tree[i] - lowbit(i),
// the number of increments it has received, for i = 1..255:
//   * tree[0] is never touched; tree[256] = 256 + (#updates) and tree[257] = 1 while data
//     symbols are coded (EOF is coded once, last), so both are derived, not stored;
//   * sum of lowbit over the Fenwick path of s is s itself, so cum(s) = s + sum of d[].
// For a byte s, level b (0..7) has exactly ONE node on s's root path,
//       e_b(s) = (s | 1<<b) & (0xFF << b),
// which is READ by the prefix sum of s when bit b of s is set and INCREMENTED by the update
// of s when bit b is clear (adaptive_tree.rs:51-59 and :83-92 walk exactly these nodes).
// And cum(s+1) uses the same eight nodes with the bits of s+1 as the mask, so one batch of
// eight independent LDS reads serves both ends of the range (adaptive_tree.rs:63-80).
//
// Two layouts:
//   U16: d[] as u16, two LANES per dword: byte address = e*128 + 4*(lane&31), half = lane>>5.
//        32 KiB per 64 blocks -> four groups of 64 blocks per CU.  Valid while every d[] stays
//        < 65536: blocks of <= 65536 symbols with the (unobservable) update of a block's last
//        symbol skipped.
//   U32: d[] as u32, byte address = e*256 + 4*lane.  64 KiB per wave; any block length.
// --------------------------------------------------------------------------------------
// The masks finish() multiplies the node pairs with, one 16-byte row per value 0..256: (v * 0x8001 >> 2j) & 0x10001,
// j = 0..3.  A byte s uses row s and row s + 1 (row 256 is all zero: the derived node 256 is added separately).
// 4 KiB, read through the vector L1 by the pair kernel's model wave.
struct MaskTable {
    uint32_t v[257 * 4];
    constexpr MaskTable() : v{}
    {
        for (uint32_t s = 0; s < 257; s++)
            for (uint32_t j = 0; j < 4; j++)
                v[4 * s + j] = ((s * 0x8001u) >> (2 * j)) & 0x10001u;
    }
};
__device__ const MaskTable k_mask_table __attribute__((aligned(128))) = MaskTable();

template <bool U16>
struct Tree {
    static constexpr uint32_t kDwords = U16 ? 256 * 32 : 256 * 64;
    static constexpr int      kShift  = U16 ? 7 : 8; // log2(row bytes)

    uint32_t *lds;
    uint32_t  A[8]; // per-level address constant: (1 << (b + kShift)) | column
    uint32_t  L;    // this lane's column (byte offset inside a row)
    uint32_t  inc;  // +1 in this lane's slot of the dword
    uint32_t  sel;  // U16: v_perm selector picking this lane's halves of two dwords
    uint32_t  hsh;  // U16: bit position of this lane's slot (0 or 16)
    uint32_t  nmask; // 0xFF << hsh
    uint32_t  psel_even, psel_odd; // U16: v_perm selectors moving byte 0 / byte 2 of a mask dword to this lane's half, zeros elsewhere

    // the eight node values of one symbol, possibly still in flight from LDS
    struct Nodes {
        uint32_t x[8];
    };

    __device__ __forceinline__ void init(uint32_t *p, uint32_t lane)
    {
        lds = p;
        // U16: lanes l and l+32 share a dword.  They belong to different LDS lane groups
        // ({0-31} and {32-63} are serviced in separate LDS cycles), and inside a group every
        // lane has its own bank: conflict-free for any per-lane row.
        L   = U16 ? (lane & 31) * 4u : lane * 4u;
        inc = (U16 && (lane >> 5)) ? 0x10000u : 1u;
        sel = (lane >> 5) ? 0x07060302u : 0x05040100u;
        hsh = (U16 && (lane >> 5)) ? 16u : 0u;
        nmask = 0xFFu << hsh;
        asm volatile("" : "+v"(nmask));
        psel_even = (lane >> 5) ? 0x0C000C0Cu : 0x0C0C0C00u; // (selector byte 0x0C = constant 0x00)
        psel_odd  = (lane >> 5) ? 0x0C020C0Cu : 0x0C0C0C02u;
        asm volatile("" : "+v"(psel_even), "+v"(psel_odd));
#pragma unroll
        for (int b = 0; b < 8; b++) {
            A[b] = (1u << (b + kShift)) | L;
            // keep A[b] an opaque VGPR: otherwise the compiler peels the constant into the
            // ds offset field and spends two more VALU ops per level re-attaching the column
            asm volatile("" : "+v"(A[b]));
        }
    }
    __device__ __forceinline__ uint32_t ld(uint32_t byte_addr) const
    {
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + byte_addr);
    }
    // returns the value BEFORE the add: one LDS op serves the query and the update
    __device__ __forceinline__ uint32_t add(uint32_t byte_addr, uint32_t v) const
    {
        return __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + byte_addr), v,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // this lane's value of the node at byte_addr (decode descent)
    __device__ __forceinline__ uint32_t node(uint32_t byte_addr) const
    {
        const uint32_t w = ld(byte_addr);
        return U16 ? ((inc == 1u) ? (w & 0xFFFFu) : (w >> 16)) : w;
    }

    // First half of get_frequency(s) (adaptive_tree.rs:105-113): touch the eight nodes.
    // UPD: each level is ONE ds_add_rtn_u32 whose addend is this lane's +1 where update(s+1)
    // increments the node (bit b of s clear) and 0 where the prefix sums only read it; the
    // returned pre-add value is the query's.  The addend is ((~s) << slot) >> b & inc: plain
    // shift + and, which a gfx950 SIMD retires at twice the rate of a compare/select pair
    // (and without the 2 wait states between a VALU VCC write and its VALU read).
    // Splitting issue() from finish() lets the caller put the next symbol's LDS traffic in
    // flight before it consumes this symbol's values.
    // issue() = fire(prep()).  prep() is the part that needs nothing but the symbol -- its row | this
    // lane's column, and its inverted bits in this lane's slot -- so a caller can compute it early.
    struct Prep {
        uint32_t t, nsl;
    };
    __device__ __forceinline__ Prep prep(uint32_t s) const
    {
        Prep r;
        r.t = (s << kShift) | L;
        asm volatile("" : "+v"(r.t)); // opaque: otherwise (ss | L) & keep is distributed back into one VOP3 per level
        r.nsl = (s << hsh) ^ nmask;   // (~s & 0xFF) << hsh with s already extracted: shift + xor
        return r;
    }
    // Level b keeps the bits of s above b and the column with ONE v_and, and the level's own bit
    // -- a constant, known clear in the masked value -- rides in the DS instruction's offset field.
    // top != nullptr: node 128 (level 7, the same node for every symbol) is kept in the caller's
    // register instead of LDS -- one atomic fewer per symbol; the caller adds it back into LDS
    // (add(A[7], *top)) before anything reads the tree from there.
    template <bool UPD>
    __device__ __forceinline__ Nodes fire(const Prep &pr, bool upd, uint32_t *top = nullptr) const
    {
        const uint32_t iv = upd ? inc : 0u;
        Nodes          n;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t keep = (((0xFFu << (b + 1)) & 0xFFu) << kShift) | ((1u << kShift) - 1u);
            const uint32_t a    = b == 7 ? A[7] : ((pr.t & keep) | (1u << (b + kShift)));
            if (UPD && b == 7 && top) {
                n.x[7] = *top;
                *top += (pr.nsl >> 7) & iv;
            } else
                n.x[b] = UPD ? add(a, (pr.nsl >> b) & iv) : ld(a);
        }
        return n;
    }
    template <bool UPD>
    __device__ __forceinline__ Nodes issue(uint32_t s, bool upd, uint32_t *top = nullptr) const
    {
        return fire<UPD>(prep(s), upd, top);
    }
    // issue() for a model that adapts, with the symbol's dot-product masks at hand (k_mask_table row s: dword j =
    // bit 2j | bit 2j+1 << 16): the addend of level b -- this lane's +1 where bit b of s is CLEAR -- is one v_perm_b32
    // of the inverted mask dword, byte 0 (even levels) or byte 2 (odd levels) moved to this lane's half, instead of
    // shift + and of the inverted symbol; four v_xor invert the masks.  13 instructions for the eight addends where
    // prep() + fire() spend 18.
    __device__ __forceinline__ Nodes issue_masked(uint32_t s, const uint4 &ms, uint32_t *top) const
    {
        static_assert(U16, "u16 trees only");
        Nodes          n;
        uint32_t       t = (s << kShift) | L;
        asm volatile("" : "+v"(t));
        const uint32_t nm[4] = {ms.x ^ 0x10001u, ms.y ^ 0x10001u, ms.z ^ 0x10001u, ms.w ^ 0x10001u};
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t keep = (((0xFFu << (b + 1)) & 0xFFu) << kShift) | ((1u << kShift) - 1u);
            const uint32_t a    = b == 7 ? A[7] : ((t & keep) | (1u << (b + kShift)));
            const uint32_t add1 = __builtin_amdgcn_perm(nm[b >> 1], nm[b >> 1], (b & 1) ? psel_odd : psel_even);
            if (b == 7 && top) {
                n.x[7] = *top;
                *top += add1;
            } else
                n.x[b] = add(a, add1);
        }
        return n;
    }
    // Second half: (low, high) of get_frequency_range(s).  d256 = number of updates so far.
    // u = s * 0x8001 puts bit i of s at bits i and 15+i, so (u >> 2j) & 0x10001 is the pair
    // (bit 2j, bit 2j+1) in the two 16-bit lanes of a v_dot2_u32_u16 -- one 32-bit shift
    // instead of a packed shift; for s+1 the same with u + 0x8001.
    __device__ __forceinline__ void finish(uint32_t s, uint32_t d256, const Nodes &n, uint32_t &lo,
                                           uint32_t &hi) const
    {
        const uint32_t m = s + 1;
        if (U16) {
            const uint32_t *xs = n.x;
            const uint32_t us = s * 0x8001u;
            const uint32_t um = us + 0x8001u;
            uint32_t       ls = s;
            uint32_t       hs = __umul24(m >> 8, d256) + m; // bit 8 of s+1 selects the derived node 256 (d256 < 2^24: one v_mad_u32_u24)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // [lo16 = this lane's d[e_2j], hi16 = this lane's d[e_2j+1]]
                const u16x2 pv = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(xs[2 * j + 1], xs[2 * j], sel));
                ls = __builtin_amdgcn_udot2(pv, __builtin_bit_cast(u16x2, (us >> (2 * j)) & 0x10001u), ls, false);
                hs = __builtin_amdgcn_udot2(pv, __builtin_bit_cast(u16x2, (um >> (2 * j)) & 0x10001u), hs, false);
            }
            lo = ls;
            hi = hs;
        } else {
            uint32_t ls = s, hs = m;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                ls += ((s >> b) & 1u) ? n.x[b] : 0u;
                hs += ((m >> b) & 1u) ? n.x[b] : 0u;
            }
            lo = ls;
            hi = hs + (m >> 8) * d256;
        }
    }
    // finish() with the eight dot-product masks of (s, s+1) taken from k_mask_table instead of computed: they depend
    // on the symbol alone, so the caller loads them symbols ahead through the vector memory path, which this kernel
    // otherwise uses once per 16 symbols.  ms / mm = masks of s / of s+1.
    __device__ __forceinline__ void finish_tab(uint32_t s, uint32_t d256, const Nodes &n, const uint4 &ms, const uint4 &mm,
                                               uint32_t &lo, uint32_t &hi) const
    {
        static_assert(U16, "u16 trees only");
        const uint32_t  m     = s + 1;
        const uint32_t *xs    = n.x;
        const uint32_t  a[4]  = {ms.x, ms.y, ms.z, ms.w};
        const uint32_t  b[4]  = {mm.x, mm.y, mm.z, mm.w};
        uint32_t        ls    = s;
        uint32_t        hs    = __umul24(m >> 8, d256) + m;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u16x2 pv = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(xs[2 * j + 1], xs[2 * j], sel));
            // (the masks of s + 1 were loaded second: used first, ONE vmcnt wait covers both loads)
            hs = __builtin_amdgcn_udot2(pv, __builtin_bit_cast(u16x2, b[j]), hs, false);
            ls = __builtin_amdgcn_udot2(pv, __builtin_bit_cast(u16x2, a[j]), ls, false);
        }
        lo = ls;
        hi = hs;
    }
    template <bool UPD>
    __device__ __forceinline__ void get_frequency(uint32_t s, uint32_t d256, bool upd, uint32_t &lo,
                                                  uint32_t &hi) const
    {
        const Nodes n = issue<UPD>(s, upd);
        finish(s, d256, n, lo, hi);
    }
};

// --------------------------------------------------------------------------------------
// floor((R1+1) * f / c) mod 2^32 for the wave-uniform divisor c (codec.rs:59-60, :133-134).
//
// Y = fma(R1, rc, rc) = (R1+1)*rc with rc = (1/c) biased UP by 4 ulp (k_fill_rc).  The
// computed Y*f equals (x/c)(1+t), x = (R1+1)*f, with 2^-53 <= t <= 2^-49.6:
//   * t > 0 : never below the true quotient, exact multiples of c included;
//   * FIXUP == false is used only when c < 2^17, so x < 2^49 and (k+1-1/c)(1+t) < k+1:
//     the truncation is the exact floor with no correction step;
//   * FIXUP == true (c up to 2^30-1): the estimate is q or q+1; the remainder mod 2^32
//     (in [0,c) or in [2^32-c, 2^32)) tells which.
// f < c always (data symbols: h <= c-1; EOF low: c-1), so the quotient is < 2^32.
//
// NONZERO (the caller knows f >= 1: every high end of a range, cum(s + 1) >= s + 1): the floor is taken by an fma,
// trunc(Y*f) = low dword of fma(Y, f, 2^52 - 0.5).  The exact sum lies in [2^52, 2^53) -- Y*f >= 2^13 here, the
// interval is at least a quarter of the code space wide and c < 2^30 -- where one ulp is 1, so the fma's single
// rounding is "Y*f - 0.5 to the nearest integer" = floor(Y*f) whenever Y*f is not an integer; and it never is: an
// exact multiple x = n*c gives Y*f = n(1 + t) with 0 < n*t < 1, anything else lies strictly between floor(x/c) and
// floor(x/c) + 1 by the bound above.  One instruction instead of v_mul_f64 + v_cvt_u32_f64, and the product is not
// rounded before the floor.  (f = 0 is why the low end keeps the two-instruction form: 0 + 2^52 - 0.5 is exactly
// representable one binade down, and its low dword is 0xFFFFFFFF.)  tests/test_fma_floor_cpu.py replays both forms
// with exact rationals.
// --------------------------------------------------------------------------------------
template <bool FIXUP, bool NONZERO = false>
__device__ __forceinline__ uint32_t scale_div(uint32_t R1, double Y, uint32_t f, uint32_t c)
{
    uint32_t q = NONZERO ? (uint32_t)__double_as_longlong(__builtin_fma(Y, (double)f, 0x1p52 - 0.5)) : (uint32_t)(Y * (double)f);
    if (FIXUP) {
        const uint32_t r = R1 * f + f - q * c;
        q -= (r >= c) ? 1u : 0u;
    }
    return q;
}

// --------------------------------------------------------------------------------------
// Encoder lane state.  low and ~high are kept LEFT-ALIGNED in 32 bits (value << sh, high
// padded with ones, sh = 32 - code_bits), which makes renormalisation independent of
// code_bits.  Output goes through a 64-bit accumulator; completed 32-bit groups are big-endian
// (MSB-first stream, bitio/mod.rs:148-181) and stored at wave-uniform base + 32-bit offset.  Stores that would pass `limit` are dropped but still counted, so an
// overflowing block ends with off > limit and is reported, never written out of bounds.
// --------------------------------------------------------------------------------------
struct EncState {
    uint32_t low, ihigh;
    uint32_t pend; // pending (E3) bit count, codec.rs:18
    uint32_t nb;   // bits waiting in acc (< 32 between symbols)
    uint32_t off;  // byte offset of the next dword from the wave's uniform base
    uint64_t acc;  // newest bit at bit 0
};

__device__ __forceinline__ void enc_init(EncState &S, uint32_t off0) // codec.rs:28-36
{
    S.low = 0; S.ihigh = 0; S.pend = 0; S.nb = 0; S.off = off0; S.acc = 0;
}

// A completed 32-bit group is stored at once, 4 bytes per lane.  (Staging four of them for one 16-byte store was built in
// round 1: a quarter of the L2 write requests, WRITE_SIZE 7.5e6 -> 5.9e6 KiB per 4 GiB pass with linear slots, at 6 % more
// kernel time -- the kernel is bound by instruction issue.  Row-major group areas took the write traffic to 1.00 x.)
// `off` is always the byte offset of the NEXT dword; slots start 16-byte aligned.
// ST: distance between a lane's consecutive dwords: 4 in a linear slot; 256 in a ROW-major group
// area, where row r holds dword r of the group's 64 lanes (see Geometry in redux_hip.hip).
// kSwapped, OR-ed into ST: the slot holds each dword as it leaves the accumulator (first stream
// bit = bit 31 of a little-endian dword), i.e. byte-reversed inside every aligned dword; the
// compaction kernel, which moves every byte anyway and has VALU to spare, restores the stream
// order.  Saves the coder wave one v_perm per symbol.
constexpr int kSwapped = 1 << 16;
template <int ST>
constexpr uint32_t stride_of = (uint32_t)(ST & 0xFFFF);
template <int ST>
__device__ __forceinline__ uint32_t stream_dword(uint32_t w) // w: the next 32 stream bits, first bit in bit 31
{
    return (ST & kSwapped) ? w : __builtin_bswap32(w);
}

template <bool CHECKED, int ST = 4>
__device__ __forceinline__ void emit_dword(EncState &S, uint32_t w, uint8_t *wbase, uint32_t limit)
{
    if (!CHECKED || S.off + stride_of<ST> <= limit)
        *reinterpret_cast<uint32_t *>(wbase + S.off) = w;
    S.off += stride_of<ST>;
}

template <int ST = 4>
__device__ __forceinline__ void put_bits(EncState &S, uint32_t val, uint32_t m, uint8_t *wbase, uint32_t limit)
{
    S.acc = (S.acc << m) | val; // m <= 32
    const uint32_t nb = S.nb + m;
    if (nb >= 32)
        emit_dword<true, ST>(S, stream_dword<ST>((uint32_t)(S.acc >> (nb - 32))), wbase, limit);
    S.nb = nb & 31u;
}

template <int ST = 4>
__device__ __forceinline__ void put_run(EncState &S, uint32_t bit, uint32_t n, uint8_t *wbase, uint32_t limit)
{
    while (n > 0) {
        const uint32_t m = n < 32 ? n : 32;
        put_bits<ST>(S, bit ? (0xFFFFFFFFu >> (32 - m)) : 0u, m, wbase, limit);
        n -= m;
    }
}

// compress_symbol (codec.rs:55-89) for one lane, given the model's (lo, hi, count).
// Returns the number of renormalisation shifts (needed by the EOF tail only).
//
// Renormalisation in closed form.  The reference loop (codec.rs:62-89) does, per iteration,
// E1/E2 (emit the common top bit) or E3 (low in the 2nd quarter, high in the 3rd: count a
// pending bit and drop bit 30).  Once an E3 step has happened low < half <= high holds for
// good, so the loop is exactly: k E1/E2 steps, k = number of leading bits low and high
// share, then j E3 steps, j = length of the run below the top bit where low has 1 and high
// has 0, then stop.  put_bit (codec.rs:39-46) makes the emitted string, for k > 0,
//     b, !b x pending, next k-1 bits of low      (b = top bit of low)
// and "b followed by P copies of !b" is the number (2^P - 1) + b, so the whole string is
//     top_k_bits(low) + ((2^P - 1) << (k-1))     in k + P bits.
template <bool FIXUP, int ST = 4>
__device__ __forceinline__ uint32_t encode_symbol(EncState &S, uint32_t lo, uint32_t hi, uint32_t c, double rc,
                                                  uint32_t sh, bool is_eof, uint8_t *wbase, uint32_t limit)
{
    const uint32_t R1 = (~(S.ihigh + S.low)) >> sh; // high - low = ~ihigh - low = ~(ihigh + low)
    const double   Y  = __builtin_fma((double)R1, rc, rc);
    const uint32_t nlow = S.low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
    // EOF has hi == c: floor(range*c/c) = range, high is unchanged (and 2^32 would not fit).
    const uint32_t nihigh = is_eof ? S.ihigh : ~(S.low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh) - 1u);

    const uint32_t x    = ~(nlow ^ nihigh); // low ^ high
    const uint32_t k    = x ? (uint32_t)__builtin_clz(x) : 32u;
    const uint64_t sl   = (uint64_t)nlow << k;
    const uint32_t topk = (uint32_t)(sl >> 32); // the k shared leading bits
    const uint32_t low2 = (uint32_t)sl;
    const uint32_t ih2  = (uint32_t)((uint64_t)nihigh << k); // ~high2: ones shifted into high
    const uint32_t t    = (low2 & ih2) << 1;
    const uint32_t j    = (uint32_t)__builtin_clz(~t); // ~t has bit 0 set: never zero
    S.low   = (low2 << j) & 0x7FFFFFFFu;
    S.ihigh = (ih2 << j) & 0x7FFFFFFFu;

    const uint32_t P  = S.pend;
    const uint32_t Pz = k ? P : 0u; // pending bits flushed by this symbol
    S.pend            = P - Pz + j;
    if (k + Pz <= 32) {
        put_bits<ST>(S, topk + (((1u << Pz) - 1u) << ((k - 1u) & 31u)), k + Pz, wbase, limit);
    } else { // long pending run: rare, bit-serial in spirit
        const uint32_t b = topk >> (k - 1);
        put_bits<ST>(S, b, 1, wbase, limit);
        put_run<ST>(S, b ^ 1u, P, wbase, limit);
        put_bits<ST>(S, topk & ((1u << (k - 1)) - 1u), k - 1, wbase, limit);
    }
    return k + j;
}

// Hot-loop version of encode_symbol for data symbols (never EOF), all 64 lanes active; the
// caller guarantees off + 4 <= limit for the whole chunk.  Differences from encode_symbol,
// none of them visible in the stream:
//   * the long-pending-run case is detected with one wave-level ballot;
//   * only the store of a completed group is predicated: `off` advances by arithmetic;
//   * CB32 (code_bits == 32) drops the alignment shifts.
template <bool FIXUP, bool CB32 = false>
__device__ __forceinline__ void encode_symbol_fast(EncState &S, uint32_t lo, uint32_t hi, uint32_t c, double rc,
                                                   uint32_t sh_, uint8_t *wbase)
{
    const uint32_t sh = CB32 ? 0u : sh_; // code_bits == 32: the alignment shifts vanish
    const uint32_t R1 = (~(S.ihigh + S.low)) >> sh;
    const double   Y  = __builtin_fma((double)R1, rc, rc);
    const uint32_t nlow   = S.low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
    uint32_t       nihigh = 0u - (S.low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh)); // ~(new high) = -(high + 1)
    asm volatile("" : "+v"(nihigh)); // opaque: low ^ high is then one v_xnor, not (high+1)-1 followed by v_xor

    // k = clz(low ^ high), 32 for low == high (v_ffbh + v_min).  The 64-bit shifts then shift
    // everything out, so the state update needs no special case; the emission takes the careful
    // path whenever k + P > 32 (for k == 32 and P == 0 the common path appends all 32 bits).
    const uint32_t x    = ~(nlow ^ nihigh);
    const uint32_t k    = x ? (uint32_t)__builtin_clz(x) : 32u;
    const uint64_t sl   = (uint64_t)nlow << k;
    const uint32_t topk = (uint32_t)(sl >> 32);
    const uint32_t low2 = (uint32_t)sl;
    const uint32_t ih2  = (uint32_t)((uint64_t)nihigh << k);
    const uint32_t t    = (low2 & ih2) << 1;
    const uint32_t j    = (uint32_t)__builtin_clz(~t);
    S.low   = (low2 << j) & 0x7FFFFFFFu;
    S.ihigh = (ih2 << j) & 0x7FFFFFFFu;

    const uint32_t P  = S.pend;
    const uint32_t Pz = k ? P : 0u;
    S.pend            = P - Pz + j;
    const uint32_t m  = k + Pz;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m > 32) == 0, 1)) {
        // "b, then Pz copies of !b" is (2^Pz - 1) + b in Pz+1 bits: v_bfm_b32 builds
        // ((1 << Pz) - 1) << (k - 1) in one instruction (both fields are taken mod 32)
        uint32_t run;
        asm("v_bfm_b32 %0, %1, %2" : "=v"(run) : "v"(Pz), "v"(k - 1u));
        S.acc = (S.acc << m) | (topk + run);
        const uint32_t nb = S.nb + m;
        if (nb >= 32) // the only predicated instruction group: shift, byte swap, store
            *reinterpret_cast<uint32_t *>(wbase + S.off) = __builtin_bswap32((uint32_t)(S.acc >> (nb - 32u)));
        S.off += (nb >> 3) & 4u; // nb < 64: +4 exactly when a group completed
        S.nb = nb & 31u;
    } else { // some lane has a pending run too long for one append: careful path for all
        const uint32_t kk = k;
        const uint32_t tk = topk;
        if (kk + Pz <= 32) {
            put_bits(S, tk + (((1u << Pz) - 1u) << ((kk - 1u) & 31u)), kk + Pz, wbase, 0xFFFFFFFFu);
        } else {
            const uint32_t b = tk >> (kk - 1);
            put_bits(S, b, 1, wbase, 0xFFFFFFFFu);
            put_run(S, b ^ 1u, P, wbase, 0xFFFFFFFFu);
            put_bits(S, tk & ((1u << (kk - 1)) - 1u), kk - 1, wbase, 0xFFFFFFFFu);
        }
    }
}

// encode_symbol_fast without its careful path: the caller keeps a copy of the state, ORs the
// returned ballot ("some lane needs more than one 32-bit append") over a run of symbols and,
// if it is non-zero, restores the copy and redoes the run with encode_symbol.  A symbol that
// needed the careful path leaves garbage in acc/nb, but `off` still advances by at most one
// dword per symbol and never beyond what the redo writes, so every stray store is overwritten.
// What encode_symbol_spec carries from symbol to symbol besides EncState (spec_begin / spec_end
// convert): nbm = bits in the accumulator - 32; r1 = high - low of the interval (what the next
// symbol scales); ih = ~high left-aligned with a stray bit 31.
struct SpecCarry {
    uint32_t nbm, r1, ih;
};
__device__ __forceinline__ SpecCarry spec_begin(const EncState &S)
{
    SpecCarry C;
    C.nbm = S.nb - 32u;
    C.r1  = ~(S.ihigh + S.low);
    C.ih  = S.ihigh;
    return C;
}
__device__ __forceinline__ void spec_end(EncState &S, const SpecCarry &C)
{
    S.nb    = C.nbm + 32u;
    S.ihigh = C.ih & 0x7FFFFFFFu;
}

template <bool FIXUP, bool CB32, int ST = 4>
__device__ __forceinline__ uint32_t encode_symbol_spec(EncState &S, SpecCarry &C, uint32_t lo, uint32_t hi, uint32_t c,
                                                       double rc, uint32_t sh_, uint8_t *wbase)
{
    const uint32_t sh = CB32 ? 0u : sh_;
    const uint32_t R1 = C.r1 >> sh;
    const double   Y  = __builtin_fma((double)R1, rc, rc);
    const uint32_t nlow   = S.low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
    uint32_t       nihigh = 0u - (S.low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh));
    const uint32_t x    = ~(nlow ^ nihigh);
    // 32-bit codes with count < 2^17 (!FIXUP): the interval is at least 2^30 wide before the
    // symbol and 2^30 / 2^17 after it, so low != high, x != 0 and k <= 31: no "x == 0 -> 32" select
    // and a 32-bit shift for ~high
    constexpr bool kNonZero = CB32 && !FIXUP;
    const uint32_t k    = kNonZero ? (uint32_t)__builtin_clz(x) : (x ? (uint32_t)__builtin_clz(x) : 32u);
    const uint32_t P = S.pend;
    // (Pz and k - 1 come out as one v_subrev_co -- borrow = "k == 0" -- and one v_cndmask on that
    // borrow.  gfx950 wants two wait states between them and the compiler spends an s_nop on one;
    // pinning both 64-bit shifts into the gap with one asm block costs a v_mov for the zero high
    // half plus a conservative s_nop after the block: no gain.)
    const uint64_t sl  = (uint64_t)nlow << k;
    const uint32_t ih2 = kNonZero ? nihigh << k : (uint32_t)((uint64_t)nihigh << k);
    const uint32_t Pz  = k ? P : 0u;
    const uint32_t km1 = k - 1u;
    const uint32_t topk = (uint32_t)(sl >> 32);
    const uint32_t low2 = (uint32_t)sl;
    // j = leading ones of (low2 & ih2) << 1 = clz of its complement, formed as ((~(low2 & ih2)) << 1) | 1:
    // a v_bitop3 (nand) and a v_lshl_or instead of and, shift, not
    const uint32_t nt   = ((~(low2 & ih2)) << 1) | 1u;
    const uint32_t j    = (uint32_t)__builtin_clz(nt);
    // The E3 steps drop bit 30 j times: shift by j, clear bit 31.  Bit 31 of both shifted values is
    // the same (j >= 1: both had ones there; j == 0: both have 0 after the k shared bits), so the
    // next interval width ~(low + ~high) is the same with or without the two clears (2 * 2^31 = 0
    // mod 2^32) and only low, which is added to twice, gets its clear.
    const uint32_t L = low2 << j;
    C.ih             = ih2 << j;
    C.r1             = ~(L + C.ih);
    S.low            = L & 0x7FFFFFFFu;
    S.pend            = P - Pz + j;
    const uint32_t m  = k + Pz;
    uint32_t run;
    asm("v_bfm_b32 %0, %1, %2" : "=v"(run) : "v"(Pz), "v"(km1));
    // acc = (acc << m) | (topk + run): the low m bits of the shifted accumulator are zero and
    // topk + run < 2^m, so the OR is an addition and the three terms are one v_add3_u32
    // (the & 63 is what v_lshlrev_b64 does anyway: no instruction)
    const uint64_t sa = S.acc << (m & 63u);
    S.acc = (sa & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)sa + topk + run);
    // nbm = (bits in the accumulator) - 32, in [-32, -1] between symbols: its sign is the "a dword
    // is complete" test, its value the shift that extracts the dword, and OR-ing -32 takes the 32
    // stored bits off again (a lane that raised the flag has garbage here; everything stays bounded)
    const uint32_t nb = C.nbm + m;
    if ((int32_t)nb >= 0) { // one exec-masked region: shift, byte swap, store, advance
        *reinterpret_cast<uint32_t *>(wbase + S.off) = stream_dword<ST>((uint32_t)(S.acc >> (nb & 63u)));
        // in place: as plain C++ the sum lands in a new register and a v_mov merges it after the region
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(S.off) : "i"(stride_of<ST>) : "memory");
    }
    C.nbm = nb | 0xFFFFFFE0u;
    return m; // the caller raises the flag if any m of the half exceeds 32
}

// The EOF tail (codec.rs:91-99) + flush_bits (bitio/mod.rs:183-198).  `shifts` is what
// encode_symbol returned for the EOF symbol.  Returns the block's stream length in bytes.
template <int ST = 4>
__device__ __forceinline__ uint32_t encode_finish(EncState &S, uint32_t shifts, uint32_t cb, uint32_t off0,
                                                  uint8_t *wbase, uint32_t limit)
{
    if (shifts < cb) {
        const uint32_t extra = cb - shifts;
        const uint32_t b     = S.low >> 31;
        put_bits<ST>(S, b, 1, wbase, limit);
        put_run<ST>(S, b ^ 1u, S.pend, wbase, limit);
        S.pend = 0;
        const uint32_t rest = extra - 1;
        if (rest > 0)
            put_bits<ST>(S, (S.low << 1) >> (32 - rest), rest, wbase, limit);
    }
    const uint32_t nbytes = (S.nb + 7) >> 3;
    const uint64_t tail   = S.nb ? (S.acc << (64 - S.nb)) : 0; // left-align, zero padding
    for (uint32_t i = 0; i < nbytes; i++) // nbytes <= 4: inside one dword in either layout
        if (S.off + (stride_of<ST> != 4 ? stride_of<ST> - 1u : (ST & kSwapped) ? 3u : i) < limit) // (limit is a dword boundary in the swapped layout)
            wbase[S.off + ((ST & kSwapped) ? 3u - i : i)] = (uint8_t)(tail >> (56 - 8 * i));
    const uint32_t size = (S.off - off0) / stride_of<ST> * 4 + nbytes; // dropped stores are still counted
    S.off += nbytes;
    S.nb = 0;
    return size;
}

} // namespace redux
