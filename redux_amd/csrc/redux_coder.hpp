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
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

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
//   U16: d[] as u16, two LANES per dword: dword index = e*32 + (lane>>1), half = lane&1.
//        32 KiB per wave -> 4 waves (one per SIMD) + staging fit the CU's 160 KiB.  Valid while
//        every d[] stays < 65536: blocks of <= 65536 symbols with the (unobservable) update
//        of a block's last symbol skipped.
//   U32: d[] as u32, dword index = e*64 + lane.  64 KiB per wave; any block length.
// --------------------------------------------------------------------------------------
template <bool U16>
struct Tree;

template <>
struct Tree<true> {
    static constexpr uint32_t kDwords = 256 * 32;
    uint32_t *t;
    uint32_t  col;   // lane >> 1
    uint32_t  inc;   // 1 or 0x10000: +1 in this lane's half
    uint32_t  sel;   // v_perm selector picking this lane's half of two dwords

    __device__ __forceinline__ void init(uint32_t *lds, uint32_t lane)
    {
        t   = lds;
        col = lane >> 1;
        inc = (lane & 1) ? 0x10000u : 1u;
        sel = (lane & 1) ? 0x07060302u : 0x05040100u;
    }
    __device__ __forceinline__ uint32_t idx(uint32_t e) const { return e * 32u + col; }
    __device__ __forceinline__ void bump(uint32_t e) const
    {
        __hip_atomic_fetch_add(&t[idx(e)], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (low, high) of get_frequency_range(s) given d256 = number of updates so far.
    __device__ __forceinline__ void range(uint32_t s, uint32_t d256, uint32_t e[8], uint32_t &lo, uint32_t &hi) const
    {
        uint32_t x[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            e[b] = (s & ((0xFFu << b) & 0xFFu)) | (1u << b);
            x[b] = t[idx(e[b])];
        }
        const uint32_t m  = s + 1;
        const u16x2    sv = __builtin_bit_cast(u16x2, s | (s << 16));
        const u16x2    mv = __builtin_bit_cast(u16x2, m | (m << 16));
        uint32_t       ls = s, hs = m;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // [lo16 = this lane's d[e_2j], hi16 = this lane's d[e_2j+1]]
            const u16x2 pv = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(x[2 * j + 1], x[2 * j], sel));
            const u16x2 sh = {(uint16_t)(2 * j), (uint16_t)(2 * j + 1)};
            const u16x2 one = {1, 1};
            ls = __builtin_amdgcn_udot2(pv, (sv >> sh) & one, ls, false);
            hs = __builtin_amdgcn_udot2(pv, (mv >> sh) & one, hs, false);
        }
        lo = ls;
        hi = hs + (s == 255u ? d256 : 0u); // bit 8 of s+1: the derived node 256
    }
    // value of node e for this lane (decode descent)
    __device__ __forceinline__ uint32_t node(uint32_t e) const
    {
        const uint32_t w = t[idx(e)];
        return (inc == 1u) ? (w & 0xFFFFu) : (w >> 16);
    }
};

template <>
struct Tree<false> {
    static constexpr uint32_t kDwords = 256 * 64;
    uint32_t *t;
    uint32_t  lane;

    __device__ __forceinline__ void init(uint32_t *lds, uint32_t l)
    {
        t    = lds;
        lane = l;
    }
    __device__ __forceinline__ uint32_t idx(uint32_t e) const { return e * 64u + lane; }
    __device__ __forceinline__ void bump(uint32_t e) const
    {
        __hip_atomic_fetch_add(&t[idx(e)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void range(uint32_t s, uint32_t d256, uint32_t e[8], uint32_t &lo, uint32_t &hi) const
    {
        uint32_t x[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            e[b] = (s & ((0xFFu << b) & 0xFFu)) | (1u << b);
            x[b] = t[idx(e[b])];
        }
        const uint32_t m  = s + 1;
        uint32_t       ls = s, hs = m;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            ls += ((s >> b) & 1u) ? x[b] : 0u;
            hs += ((m >> b) & 1u) ? x[b] : 0u;
        }
        lo = ls;
        hi = hs + (s == 255u ? d256 : 0u);
    }
    __device__ __forceinline__ uint32_t node(uint32_t e) const { return t[idx(e)]; }
};

// update(s+1) of adaptive_tree.rs:83-92 restricted to the stored nodes: +1 on every level
// whose bit of s is clear (e_b > s  <=>  bit b of s is clear).
template <bool U16>
__device__ __forceinline__ void tree_update(const Tree<U16> &T, uint32_t s, const uint32_t e[8])
{
#pragma unroll
    for (int b = 0; b < 8; b++)
        if (e[b] > s)
            T.bump(e[b]);
}

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
// --------------------------------------------------------------------------------------
template <bool FIXUP>
__device__ __forceinline__ uint32_t scale_div(uint32_t R1, double Y, uint32_t f, uint32_t c)
{
    uint32_t q = (uint32_t)(Y * (double)f);
    if (FIXUP) {
        const uint32_t r = R1 * f + f - q * c;
        q -= (r >= c) ? 1u : 0u;
    }
    return q;
}

// --------------------------------------------------------------------------------------
// Encoder lane state.  low/high are kept LEFT-ALIGNED in 32 bits (value << sh, high padded
// with ones, sh = 32 - code_bits), which makes the renormalisation independent of code_bits.
// --------------------------------------------------------------------------------------
struct EncState {
    uint32_t low, high;
    uint32_t pend; // pending (E3) bit count, codec.rs:18
    uint32_t nb;   // valid bits in acc (< 32 between calls)
    uint32_t pos;  // bytes emitted so far (multiple of 4 until the final flush)
    uint64_t acc;  // bit accumulator, newest bit at bit 0
};

// BitWriter::write_bits for m <= 32 bits (bitio/mod.rs:148-181): MSB-first, so a completed
// 32-bit group is stored big-endian.  Stores beyond cap are dropped but still counted, so an
// overflowing block ends with pos > cap and is reported, never written out of bounds.
__device__ __forceinline__ void put_bits(EncState &S, uint32_t val, uint32_t m, uint8_t *out, uint32_t cap)
{
    S.acc = (S.acc << m) | val;
    S.nb += m;
    if (S.nb >= 32) {
        const uint32_t w = (uint32_t)(S.acc >> (S.nb - 32));
        if (S.pos + 4 <= cap)
            *reinterpret_cast<uint32_t *>(out + S.pos) = __builtin_bswap32(w);
        S.pos += 4;
        S.nb -= 32;
    }
}

__device__ __forceinline__ void put_run(EncState &S, uint32_t bit, uint32_t n, uint8_t *out, uint32_t cap)
{
    while (n > 0) {
        const uint32_t m = n < 32 ? n : 32;
        put_bits(S, bit ? (0xFFFFFFFFu >> (32 - m)) : 0u, m, out, cap);
        n -= m;
    }
}

// compress_symbol (codec.rs:55-101) for one lane, given the model's (lo, hi, count).
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
template <bool FIXUP>
__device__ __forceinline__ void encode_symbol(EncState &S, uint32_t lo, uint32_t hi, uint32_t c, double rc,
                                              uint32_t sh, bool is_eof, uint8_t *out, uint32_t cap)
{
    const uint32_t R1 = (S.high - S.low) >> sh; // range - 1
    const double   Y  = __builtin_fma((double)R1, rc, rc);
    const uint32_t ql = scale_div<FIXUP>(R1, Y, lo, c);
    // EOF has hi == c: floor(range*c/c) = range, high is unchanged (and 2^32 would not fit).
    const uint32_t nhigh = is_eof ? S.high : S.low + (scale_div<FIXUP>(R1, Y, hi, c) << sh) - 1u;
    const uint32_t nlow  = S.low + (ql << sh);

    const uint32_t x  = nlow ^ nhigh;
    const uint32_t k  = x ? (uint32_t)__builtin_clz(x) : 32u;
    const uint64_t sl = (uint64_t)nlow << k;
    const uint32_t topk  = (uint32_t)(sl >> 32);           // the k shared leading bits
    const uint32_t low2  = (uint32_t)sl;
    const uint32_t ihigh2 = (uint32_t)((uint64_t)(~nhigh) << k); // ~high2 (ones shifted in)
    const uint32_t t  = (low2 & ihigh2) << 1;
    const uint32_t j  = (uint32_t)__builtin_clz(~t);       // ~t has bit 0 set: never zero
    S.low  = (low2 << j) & 0x7FFFFFFFu;
    S.high = ~((ihigh2 << j) & 0x7FFFFFFFu);

    const uint32_t P = S.pend;
    if (k > 0) {
        if (k + P <= 32) {
            put_bits(S, topk + (((1u << P) - 1u) << (k - 1)), k + P, out, cap);
        } else { // long pending run: rare, bit-serial in spirit
            put_bits(S, topk >> (k - 1), 1, out, cap);
            put_run(S, (topk >> (k - 1)) ^ 1u, P, out, cap);
            put_bits(S, topk & ((1u << (k - 1)) - 1u), k - 1, out, cap);
        }
        S.pend = j;
    } else {
        S.pend = P + j;
    }

    if (is_eof) { // codec.rs:91-99
        const uint32_t cb    = 32 - sh;
        const uint32_t shifts = k + j;
        if (shifts < cb) {
            const uint32_t extra = cb - shifts;
            const uint32_t b     = S.low >> 31;
            put_bits(S, b, 1, out, cap);
            put_run(S, b ^ 1u, S.pend, out, cap);
            S.pend = 0;
            const uint32_t rest = extra - 1;
            if (rest > 0)
                put_bits(S, (S.low << 1) >> (32 - rest), rest, out, cap);
        }
        // flush_bits (bitio/mod.rs:183-198): left-align the tail and pad with zeros
        const uint32_t nbytes = (S.nb + 7) >> 3;
        const uint64_t tail   = S.nb ? (S.acc << (64 - S.nb)) : 0;
        for (uint32_t i = 0; i < nbytes; i++)
            if (S.pos + i < cap)
                out[S.pos + i] = (uint8_t)(tail >> (56 - 8 * i));
        S.pos += nbytes;
        S.nb = 0;
    }
}

} // namespace redux
