// redux_decode.hpp -- decode kernels of the MI355X block coder (gfx950 only).
//
//   k_decode<U16,FIXUP>  one wave per 64 blocks, per-lane control flow (general form: u32 trees, count >= 2^17)
//   decode_lock_body     the lock-step decoder body of the STATIC-table kernels (redux_static.hpp); shared pieces of the
//                        adaptive lock-step decoder k_decode_lock (redux_decode_adaptive.hpp): DecArgs, DecLane, DecTop,
//                        dec_value, dec_commit_careful
//   k_rcp_check          device-side exhaustive check of the reciprocal bound dec_value relies on
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

// ======================================================================================
// decode
// ======================================================================================
struct DecArgs {
    const uint8_t  *in;
    const uint64_t *in_offsets; // nblocks + 1
    uint64_t        nblocks;
    uint8_t        *out;        // block b at out + b*block_size
    uint32_t       *out_sizes;
    int32_t        *status;
    const double   *rc;
    uint32_t        block_size;
    uint32_t        nfreeze;
    uint32_t        code_bits;
    uint32_t        aligned4;   // 1: out and block_size are 4-byte multiples; 2: 16-byte multiples
    uint64_t       *in_used;    // optional: bytes of each stream the reader fetched (ByteCount, bitio/mod.rs:71)
    // Block table (redux_decode_blocks_v_dev; null: slot j decodes block j to out + j * block_size, capacity block_size).
    // Entry j: the block numbered `index` (its stream is in_offsets[index] .., its size / status go to entry `index`) is
    // written at out + offset and may hold `length` <= block_size bytes; index 0xFFFFFFFF: an idle lane.  nblocks counts
    // table entries then.
    const redux_block *table;
    // Entries of the reciprocal table at rc (k_fill_rc: rc[i] for 257 + i).  It covers min(block capacity, freeze point)
    // updates, but at most kDecRcWindow: a decoder that may run past that (k_decode, k_decode_wave: blocks of any
    // length, redux_decompress with a generous capacity) computes the reciprocals beyond it itself (rc_lookup), so the
    // workspace is bounded whatever the capacity.
    uint32_t rc_n;
};
constexpr uint32_t kDecRcWindow = 1u << 20;

// The reciprocal the coder multiplies with at update number `nup` (wave-uniform), count = count0 + nup: from the table
// while it lasts, otherwise computed as k_fill_rc computes an entry -- the correctly rounded 1 / count, biased up 4 ulp
// (scale_div's proof needs exactly that value) -- ~15 instructions of a step that has ~165.
__device__ __forceinline__ double rc_lookup(rc_ptr rc, uint32_t rc_n, uint32_t nup, uint32_t count0)
{
    if (__builtin_expect(nup < rc_n, 1)) // (scalar branch)
        return rc[nup];
    const double r = 1.0 / (double)(count0 + nup);
    return __longlong_as_double(__double_as_longlong(r) + 4);
}

// BitReader (bitio/mod.rs:78-120) as a 64-bit look-ahead register: the `cnt` not yet
// consumed bits sit in the TOP of `bits`; refills are whole aligned dwords, big-endian
// (MSB-first stream).  Reads past the stream's last dword yield zeros; running past the end
// is detected by the consumed-bit count, exactly where read_bits would return Err(Eof).
struct BitIn {
    uint64_t        bits;
    uint32_t        cnt;
    uint32_t        nextw; // the following dword, already loaded: a refill never waits on memory
    const uint32_t *rp, *end;

    __device__ __forceinline__ uint32_t fetch()
    {
        const uint32_t w = rp < end ? *rp : 0u;
        rp++;
        return w;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            bits |= (uint64_t)__builtin_bswap32(nextw) << (32 - cnt);
            cnt += 32;
            nextw = fetch(); // consumed by the NEXT refill of this lane, several symbols from now
        }
    }
    __device__ __forceinline__ void init(const uint8_t *sp, uint64_t size)
    {
        const uintptr_t a = (uintptr_t)sp & ~(uintptr_t)3;
        const uint32_t  skip = (uint32_t)((uintptr_t)sp & 3) * 8;
        rp    = reinterpret_cast<const uint32_t *>(a);
        end   = reinterpret_cast<const uint32_t *>(((uintptr_t)sp + size + 3) & ~(uintptr_t)3);
        bits  = 0;
        cnt   = 0;
        nextw = fetch();
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

template <bool U16, bool FIXUP>
__global__ void __launch_bounds__(64) k_decode(DecArgs a)
{
    __shared__ uint32_t lds[Tree<U16>::kDwords];
    constexpr int  KS   = Tree<U16>::kShift;
    const uint32_t lane = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = slot < a.nblocks && !(a.table && a.table[slot].index == 0xFFFFFFFFu /* idle entry */);
    uint64_t       blk = slot, dst_off = slot * (uint64_t)a.block_size;
    uint32_t       capn = a.block_size;
    if (a.table && live) { // block table: see DecArgs
        const redux_block e = a.table[slot];
        blk     = e.index;
        dst_off = e.offset;
        capn    = e.length;
    }

    for (uint32_t i = lane; i < Tree<U16>::kDwords / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    Tree<U16> T;
    T.init(lds, lane);

    const uint32_t cb = a.code_bits, sh = 32 - cb;
    uint64_t       size = 0;
    const uint8_t *sp   = a.in;
    if (live) {
        const uint64_t o0 = a.in_offsets[blk];
        size              = a.in_offsets[blk + 1] - o0;
        sp                = a.in + o0;
    }
    const uint64_t stream_bits = size * 8;
    uint8_t       *dst         = a.out + (live ? dst_off : 0);
    const rc_ptr   rcp         = (rc_ptr)a.rc;

    BitIn B;
    B.init(sp, live ? size : 0);
    // decompress_symbol's first call pulls code_bits bits into `pending` (codec.rs:124-127).
    // W holds that value left-aligned (value << sh), like low/high.
    uint32_t W        = B.take(cb) << sh;
    uint64_t consumed = cb;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    int32_t  st   = REDUX_OK;
    bool     done = !live;
    if (live && consumed > stream_bits) { // stream shorter than code_bits: Err(Eof) at once
        st   = REDUX_EOF;
        done = true;
    }
    uint32_t n_out = 0;
    uint32_t obuf  = 0;

    for (uint32_t p = 0;; p++) {
        if (__builtin_amdgcn_readfirstlane(__ballot(!done) == 0))
            break;
        const uint32_t nup = p < a.nfreeze ? p : a.nfreeze;
        const double   rc  = rc_lookup(rcp, a.rc_n, nup, 257u);
        const uint32_t c   = 257u + nup;
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
            // get_symbol (adaptive_tree.rs:115-136): the descent probes exactly the nodes
            // e_b(s); the same eight values give cum(s+1), and the levels where the descent
            // went left (bit clear) are the ones update(s+1) increments.
            uint32_t lo, hi;
            bool     is_eof = false;
            uint32_t s      = 0;
            if (v >= c - 1) { // first probe: tree[256] = 256 + #updates = count - 1
                is_eof = true;
                lo     = c - 1;
                hi     = c;
            } else {
                uint32_t x[8], ea[8];
                uint32_t i = 0, rem = v;
#pragma unroll
                for (int b = 7; b >= 0; b--) {
                    ea[b] = (i << KS) | T.A[b];
                    x[b]  = T.node(ea[b]);
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
                for (int b = 0; b < 8; b++)
                    hs += ((m >> b) & 1u) ? x[b] : 0u;
                hi = hs + (s == 255u ? nup : 0u);
                if (p < a.nfreeze) {
#pragma unroll
                    for (int b = 0; b < 8; b++)
                        T.add(ea[b], ((s >> b) & 1u) ? 0u : T.inc);
                }
            }
            if (is_eof) { // codec.rs:136-138: returns before any renormalisation
                done = true;
            } else {
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
                } else if (p >= capn) { // the symbol is decoded; writing it is what fails (codec.rs:171)
                    st   = REDUX_OUTPUT_TOO_SMALL;
                    done = true;
                } else {
                    // k E1/E2 steps shift the value left (codec.rs:143-146 + :155-157); each of
                    // the j E3 steps then drops the bit just below the top one (:147-151).  On
                    // the 64-bit image [value | n new bits] that is: shift by k, remember the top
                    // bit, shift by j more, put the remembered top bit back.
                    const uint32_t nb   = B.take(n);
                    const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nb << (32 + sh - n));
                    const uint64_t c1   = comb << k;
                    const uint64_t c2   = c1 << j;
                    W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) &
                        (0xFFFFFFFFu << sh);
                    // emit the symbol (write_bits(symbol, 8), codec.rs:171)
                    if (a.aligned4) {
                        obuf |= s << (8 * (p & 3));
                        if ((p & 3) == 3) {
                            *reinterpret_cast<uint32_t *>(dst + (p & ~3u)) = obuf;
                            obuf = 0;
                        }
                    } else {
                        dst[p] = (uint8_t)s;
                    }
                    n_out = p + 1;
                }
            }
        }
    }
    if (live) {
        if (a.aligned4)
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

// --------------------------------------------------------------------------------------
// Pieces shared by the lock-step decoders (the adaptive one, redux_decode_adaptive.hpp, and the static-table ones below).
// The descent keeps q = ~rem.  For a node value t, q2 = q + t is ~(rem - t): its top bit is the "go right" flag, the new
// q is max_u32(q, q2) (q2 wraps to a small number when the probe fails), and the flags are shifted into the symbol by
// v_alignbit.  cum(s+1) falls out of the same probes: it is the upper boundary of the LAST level where the descent went
// left, i.e. v + 1 + min_u32 over the levels of q2 (failed probes give the small values and the boundary only shrinks on
// the way down; the virtual root probe against tree[256] = count - 1 seeds the minimum).
// --------------------------------------------------------------------------------------
struct DecFound {
    uint32_t s, lo, hi;
    uint32_t eofq; // top bit set: v >= count - 1, the first probe of get_symbol fails -> EOF (adaptive_tree.rs:116)
};

// The seven nodes of levels 7, 6, 5 (128; 64, 192; 32, 96, 160, 224) are at fixed positions, so a decoder lane keeps
// them in VGPRs: the first three probes of every descent need no LDS round trip.
struct DecTop {
    uint32_t n128, n64, n192, n32, n96, n160, n224; // full tree values (lowbit + increments): u32, no overflow to think about
};
__device__ __forceinline__ DecTop dec_top_new() { return {128u, 64u, 64u, 32u, 32u, 32u, 32u}; }

// The same descent under a STATIC model (redux_static.hpp): node i, i = 1..255, is the Fenwick form of the fixed
// cumulative table -- cum[i] - cum[i - lowbit(i)], the total frequency of the symbols in [i - lowbit(i), i) -- as
// u32, ONE table per workgroup shared by its lanes (nothing is ever updated).  The lanes of a wave read DIFFERENT
// rows of it at the same column (a probe is node prefix + 2^b), so node i lives at dword i + (i >> 5): one pad
// dword per 32 puts the 8 possible round-B rows and the 32 possible round-C rows in distinct banks instead of
// one.  (On Zipf bytes it measures the same either way -- most lanes probe the same rows, which is a broadcast --
// and SQ_LDS_BANK_CONFLICT reads 0 with the padding; uniform symbols are where the unpadded table would pay.)
// cum256 = cum[256], the total of the data symbols (what tree[256] = count - 1 is to the adaptive model).
__device__ __forceinline__ uint32_t dec_static_slot(uint32_t i) { return i + (i >> 5); }
constexpr uint32_t kStaticTreeDwords = 272; // 256 + 8 pad dwords, rounded up to a 64-byte multiple

__device__ __forceinline__ DecFound dec_search_static(const uint32_t *tab, const DecTop &T, uint32_t v, uint32_t cum256)
{
    uint32_t q = ~v, hq = q + cum256, bits = 0, q2;
    DecFound f;
    f.eofq = hq;
#define REDUX_DEC_LEVEL(t)                                                                                             \
    left = __builtin_uadd_overflow(q, (t), &q2);                                                                       \
    bits = __builtin_amdgcn_alignbit(bits, q2, 31);                                                                    \
    q    = q > q2 ? q : q2;                                                                                            \
    hq   = hq < q2 ? hq : q2;
    bool left;
    REDUX_DEC_LEVEL(T.n128)
    const uint32_t x6 = left ? T.n64 : T.n192, c5l = left ? T.n32 : T.n160, c5r = left ? T.n96 : T.n224;
    REDUX_DEC_LEVEL(x6)
    const uint32_t x5 = left ? c5l : c5r;
    REDUX_DEC_LEVEL(x5)
    const uint32_t *r = tab + (bits & 7u) * 33u; // prefix i = bits << 5 at dword i + (i >> 5)
    const uint32_t w16 = r[16], w8 = r[8], w24 = r[24];
    REDUX_DEC_LEVEL(w16)
    const uint32_t x3 = left ? w8 : w24;
    REDUX_DEC_LEVEL(x3)
    const uint32_t  p5 = bits & 31u; // prefix i = p5 << 3: nodes i .. i+7 sit inside one block of 32
    const uint32_t *r4 = tab + (p5 << 3) + (p5 >> 2);
    const uint32_t  n1 = r4[1], n2 = r4[2], n3 = r4[3], n4 = r4[4], n5 = r4[5], n6 = r4[6], n7 = r4[7];
    REDUX_DEC_LEVEL(n4)               // node i+4
    const uint32_t e1  = left ? n2 : n6; // level 1: node i+2 or i+6
    const uint32_t e0l = left ? n1 : n5, e0r = left ? n3 : n7; // level 0: i+1 / i+5 or i+3 / i+7
    REDUX_DEC_LEVEL(e1)
    const uint32_t x0 = left ? e0l : e0r;
    REDUX_DEC_LEVEL(x0)
#undef REDUX_DEC_LEVEL
    f.s  = bits & 0xFFu;
    f.lo = v + q + 1u;  // v - rem = cum[s]
    f.hi = v + hq + 1u; // cum[s + 1]
    return f;
}

// get_symbol under a static model whose total is at most 2^16, by direct lookup: lut[v] = the symbol whose range holds
// v (one byte per code value, 64 KiB of LDS shared by the four waves of a workgroup), then cum[s] and cum[s+1] from the
// plain table with one ds_read2_b32.  Two LDS round trips and six instructions where the descent above takes three
// round trips and some fifty.
__device__ __forceinline__ DecFound dec_search_lut(const uint8_t *lut, const uint32_t *ctab, uint32_t v, uint32_t cum256)
{
    DecFound f;
    f.eofq = cum256 - 1u - v; // top bit set: v >= cum[256], the EOF symbol (lut[] holds 255 there)
    f.s    = lut[v & 0xFFFFu]; // (a finished lane's v is garbage: stay inside the table)
    f.lo   = ctab[f.s];
    f.hi   = ctab[f.s + 1u];
    return f;
}

// update(s+1) (adaptive_tree.rs:83-92): +1 on the levels where bit b of s is clear.  Levels 7-5
// live in registers: node e of level b is incremented iff s lies in [e - 2^b, e), an unsigned
// range compare + add-with-carry; levels 4-0 are fire-and-forget ds_add_u32.
__device__ __forceinline__ void dec_update_regs(DecTop &T, uint32_t s)
{
    T.n128 += s < 128u ? 1u : 0u;
    T.n64 += s < 64u ? 1u : 0u;
    T.n192 += (s - 128u) < 64u ? 1u : 0u;
    T.n32 += s < 32u ? 1u : 0u;
    T.n96 += (s - 64u) < 32u ? 1u : 0u;
    T.n160 += (s - 128u) < 32u ? 1u : 0u;
    T.n224 += (s - 192u) < 32u ? 1u : 0u;
}
// value = floor(((V - low + 1) * count - 1) / range) (codec.rs:129-131) in f64.  Numerator nd (< 2^49) and range xd (an
// integer in [1, 2^32]) are exact.  The raw v_rcp_f64 of gfx950 is within 2^-24 (measured: 2^-24.4) of 1/xd for EVERY
// integer xd in [1, 2^32] (checked exhaustively on the device by redux_debug_rcp_check, tests/test_gpu_parity.py), and the
// quotient is below count < 2^17.1, so nd * rcp(xd) is within 0.0084 of it; ONE fma subtracts 2^-6 on the way, which puts
// the estimate strictly below the quotient and less than 0.024 away: truncated it is q or q - 1 (a negative estimate
// converts to 0 = q), and one exact f64 remainder (fma; v * xd < 2^50) adds the 1 back.
__device__ __forceinline__ uint32_t dec_value(double R1d, uint32_t Vd, double cd, double cdm1)
{
    const double xd = R1d + 1.0;
    const double nd = __builtin_fma((double)Vd, cd, cdm1); // (Vd+1)*c - 1, exact (< 2^49)
    uint32_t     v  = (uint32_t)__builtin_fma(nd, __builtin_amdgcn_rcp(xd), -0x1p-6);
    // (Left as it is on purpose: the compiler keeps the uncorrected quotient and the comparison's VCC alive through the
    // search and forms ~v, -v and v + 1 with carry-in instructions.  Pinning v, or only the correction bit, into a register
    // frees VCC for the probes' carries and measured 0.6 - 0.9 ms slower, profiles/r02_decode/pair_experiment.txt.)
    v += __builtin_fma(-(double)v, xd, nd) >= xd ? 1u : 0u;
    return v;
}

// exhaustive check behind dec_value: max over all integers x in [lo, hi] of |rcp(x) * x - 1|, as
// the f64 bit pattern of the maximum (positive doubles order like their bits)
__global__ void k_rcp_check(uint64_t lo, uint64_t hi, unsigned long long *max_bits)
{
    double m = 0.0;
    for (uint64_t x = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x <= hi; x += (uint64_t)gridDim.x * blockDim.x) {
        const double xd = (double)x;
        const double e  = __builtin_fabs(__builtin_fma(__builtin_amdgcn_rcp(xd), xd, -1.0));
        m               = e > m ? e : m;
    }
    atomicMax(max_bits, (unsigned long long)__double_as_longlong(m));
}

struct DecLane {
    uint32_t low, ihigh; // ihigh = ~high; both left-aligned as in EncState
    uint32_t W;          // code value (codec.rs `pending`), left-aligned
    uint64_t bbits;      // upcoming stream bits, left-aligned
    uint32_t bcnt;       // how many of them are valid
    uint32_t consumed;   // stream bits pulled so far
    uint32_t obuf;
    uint32_t n_out;      // symbols emitted: set when the block finishes (a live lane has emitted one per step)
    uint32_t dflag;      // 0x80000000 once the block is finished (EOF symbol or error)
    int32_t  st;
};

// Per-lane, predicated end of a step: decompress_symbol after the model answered
// (codec.rs:133-161) + decompress_stream's emission (:170-172).  `room`: p < block capacity.
template <bool CB32, typename UPD>
__device__ __forceinline__ void dec_commit_careful(DecLane &S, const DecFound &f, uint32_t R1, double R1d, double rc, uint32_t c,
                                                   uint32_t sh, uint32_t stream_bits, uint32_t p, bool room, bool aligned4,
                                                   uint8_t *dst, UPD &&update)
{
    if ((int32_t)S.dflag < 0)
        return;
    if ((int32_t)f.eofq < 0) { // codec.rs:136-138: returns before any renormalisation
        S.dflag = 0x80000000u;
        S.n_out = p;
        return;
    }
    update(f.s); // the model's update(s+1), adaptive_tree.rs:83-92 (nothing for a frozen or static model)
    const double   Y      = __builtin_fma(R1d, rc, rc);
    const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, f.lo, c) << sh);
    const uint32_t nihigh = 0u - (S.low + (scale_div<false, true>(R1, Y, f.hi, c) << sh));
    const uint32_t xx     = ~(nlow ^ nihigh);
    const uint32_t k      = xx ? (uint32_t)__builtin_clz(xx) : 32u;
    const uint32_t low2   = (uint32_t)((uint64_t)nlow << k);
    const uint32_t ih2    = (uint32_t)((uint64_t)nihigh << k);
    const uint32_t t2     = (low2 & ih2) << 1;
    const uint32_t j      = (uint32_t)__builtin_clz(~t2);
    S.low                 = (low2 << j) & 0x7FFFFFFFu;
    S.ihigh               = (ih2 << j) & 0x7FFFFFFFu;
    const uint32_t n      = k + j; // bits pulled by get_bit (codec.rs:157)
    S.consumed += n;
    if (S.consumed > stream_bits) { // read_bits would return Err(Eof) (bitio/mod.rs:107)
        S.st    = REDUX_EOF;
        S.dflag = 0x80000000u;
        S.n_out = p;
        return;
    }
    // decompress_symbol (codec.rs:123-161) has returned the symbol; only now does decompress_stream
    // try to write it (codec.rs:171), so a stream that runs dry in this symbol's renormalisation is
    // Err(Eof) even when the block is full as well
    if (!room) {
        S.st    = REDUX_OUTPUT_TOO_SMALL;
        S.dflag = 0x80000000u;
        S.n_out = p;
        return;
    }
    // [value | next 32 bits] << k, keep the top bit, << j, put it back (codec.rs:143-157)
    const uint32_t nxt  = (uint32_t)(S.bbits >> 32);
    const uint64_t comb = ((uint64_t)(S.W >> sh) << (32 + sh)) | ((uint64_t)nxt << sh);
    const uint64_t c1   = comb << k;
    const uint64_t c2   = c1 << j;
    S.W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) & (0xFFFFFFFFu << sh);
    S.bbits <<= n;
    S.bcnt -= n;
    if (aligned4)
        S.obuf |= f.s << (8 * (p & 3));
    else
        dst[p] = (uint8_t)f.s;
}

// STATIC: the model is a fixed table (redux_static.hpp): `lds` starts with its Fenwick form (256 dwords, shared
// by the lanes) instead of 64 per-lane trees, nothing is updated, total frequency and reciprocal are constants.
// Everything else -- code value, narrowing, renormalisation, bit reader, output staging, the careful per-lane
// commit -- is the same code.
// MODE 1: descent over the table's Fenwick form; 2: direct lookup (`lds` is then this wave's stream ring alone, `lut` /
// `ctab` the workgroup's shared tables, filled by the caller).  (The adaptive model's decoder, which this body was
// derived from, has its own tree layout and loop structure: redux_decode_adaptive.hpp.)
template <bool CB32, int MODE>
__device__ __forceinline__ void decode_lock_body(const DecArgs &a, uint32_t *lds, const uint32_t *cum, double rc_static,
                                                 uint32_t lane = threadIdx.x, uint64_t group = blockIdx.x,
                                                 const uint8_t *lut = nullptr, const uint32_t *ctab = nullptr)
{
    static_assert(MODE == 1 || MODE == 2, "static models only");
    constexpr uint32_t kModelBytes = MODE == 2 ? 0 : kStaticTreeDwords * 4;
    const uint64_t blk  = group * 64 + lane;
    const bool     live = blk < a.nblocks;

    if (MODE == 1) { // (MODE 2: tables filled by the caller, which also synchronises)
        for (uint32_t i = lane; i < 256; i += 64)
            lds[dec_static_slot(i)] = i ? cum[i] - cum[i - (i & (0u - i))] : 0u;
    }
    __syncthreads();
    const uint32_t L = lane * 4u;

    const uint32_t cb = CB32 ? 32u : a.code_bits, sh = CB32 ? 0u : 32 - cb;
    uint64_t       size = 0;
    const uint8_t *sp   = a.in;
    if (live) {
        const uint64_t o0 = a.in_offsets[blk];
        size              = a.in_offsets[blk + 1] - o0;
        sp                = a.in + o0;
    }
    const uint32_t stream_bits = (uint32_t)(size * 8);
    uint8_t       *dst         = a.out + (live ? blk : 0) * (uint64_t)a.block_size;
    const uint32_t capn        = a.block_size;
    const bool     aligned4    = a.aligned4 != 0;
    const bool     aligned16   = a.aligned4 == 2;

    // Bit reader (bitio/mod.rs:78-120).  The stream is read as aligned dwords from a per-lane base.
    // A lock-step wave waits for the SLOWEST of its 64 lanes on every vector-memory wait, and
    // with 64 independent streams some lane misses to HBM nearly every step, so a load that is
    // consumed one step later bounds the step at the memory latency (measured: 1700 cycles per
    // step whatever the step computes).  Hence a ring of 32 dwords per lane in LDS, filled by
    // the producer below (one 16-byte load per lane and group of four steps, retired into the
    // ring a whole group later) and drained by the reader with LDS reads: `fetched` is always
    // the dword at index rpo, read from the ring a step before it can be consumed.
    //   * dword d of lane l: ring byte RB + ((d & 31) << 8) + 4l (conflict-free per-lane rows);
    //   * chunk wr (dwords 4wr..4wr+3) is requested while 4wr - rpo <= 28, so its slot's old
    //     content (chunk wr-8) is consumed; a step consumes < 1 dword, a group < 4: the ring
    //     never runs dry (initial fill: 24 dwords);
    //   * indices are clamped to the stream's last dword: bits past the end of a stream are
    //     never USED (consuming them is the Eof error, detected by the bit count), so their
    //     value does not matter, but the loads must stay inside the buffer.  A lane without a
    //     stream reads the offsets table instead (always mapped) and is done from the start.
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4 *gptr4;
    const bool      has    = live && size > 0;
    const uintptr_t sp_abs = (uintptr_t)sp;
    const gptr      gin    = has ? (gptr)(sp_abs & ~(uintptr_t)3) : (gptr)(uintptr_t)a.in_offsets;
    const uint32_t  rpo_last = has ? (uint32_t)(((((sp_abs + size + 3) & ~(uintptr_t)3) - (sp_abs & ~(uintptr_t)3)) >> 2) - 1) : 0u;
    const uint32_t  skip   = has ? (uint32_t)(sp_abs & 3) * 8 : 0u;
    const gptr      gsafe  = (gptr)(uintptr_t)a.in_offsets;
    auto rd = [&](uint32_t o) { return gin[o < rpo_last ? o : rpo_last]; };
    constexpr uint32_t RB = kModelBytes;
    auto ring_write = [&](uint32_t chunk, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
        uint32_t *q = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + RB + ((chunk & 7u) << 10) + L);
        q[0] = x0; q[64] = x1; q[128] = x2; q[192] = x3;
    };
    auto ring_read = [&](uint32_t d) {
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + RB + ((d & 31u) << 8) + L);
    };
    uint32_t rpo = 2, wr = 0, pend_chunk = 0;
    u32x4    ldq = {0, 0, 0, 0};
    DecLane  S;
    {
        uint32_t d0 = 0, d1 = 0;
        for (; wr < 6; wr++) {
            const uint32_t x0 = rd(4 * wr), x1 = rd(4 * wr + 1), x2 = rd(4 * wr + 2), x3 = rd(4 * wr + 3);
            if (wr == 0) {
                d0 = has ? __builtin_bswap32(x0) : 0u;
                d1 = (has && rpo_last >= 1) ? __builtin_bswap32(x1) : 0u;
            }
            ring_write(wr, x0, x1, x2, x3);
            ldq        = u32x4{x0, x1, x2, x3};
            pend_chunk = wr; // (the first group "retires" chunk 5 once more)
        }
        S.bbits = (((uint64_t)d0 << 32) | d1) << skip;
        S.bcnt  = 64 - skip;
    }
    uint32_t fetched = ring_read(rpo);
    S.W = (uint32_t)((S.bbits >> 1) >> (63 - cb)) << sh; // codec.rs:124-127
    S.bbits <<= cb;
    S.bcnt -= cb;
    S.consumed = cb;
    S.low = 0; S.ihigh = 0;
    S.st = REDUX_OK;
    S.dflag = live ? 0u : 0x80000000u;
    if (live && S.consumed > stream_bits) { // stream shorter than code_bits: Err(Eof) at once
        S.st    = REDUX_EOF;
        S.dflag = 0x80000000u;
    }
    S.n_out = 0;
    S.obuf  = 0;
    uint32_t stored = 0; // bytes [0, stored) of the block are in memory
    uint32_t staged = 0; // bytes [stored, staged) are whole dwords waiting in oq (newest in .w)
    uint4    oq     = make_uint4(0, 0, 0, 0);
    uint32_t p      = 0;
    DecTop   T      = dec_top_new();
    if (MODE == 1)
        T = DecTop{lds[dec_static_slot(128)], lds[dec_static_slot(64)], lds[dec_static_slot(192)], lds[dec_static_slot(32)],
                   lds[dec_static_slot(96)], lds[dec_static_slot(160)], lds[dec_static_slot(224)]};
    const uint32_t cum256  = cum[256];
    const uint32_t c_const = cum[257]; // total_frequency() of the static model
#define REDUX_DEC_SEARCH(v_) (MODE == 2 ? dec_search_lut(lut, ctab, v_, cum256) : dec_search_static(lds, T, v_, cum256))

#define REDUX_DEC_READER                                                                                               \
    {                                                                                                                  \
        const bool     need = S.bcnt <= 32;                                                                            \
        const uint64_t add  = (uint64_t)(need ? __builtin_bswap32(fetched) : 0u) << ((32 - S.bcnt) & 63);              \
        S.bbits |= add;                                                                                                \
        S.bcnt += need ? 32u : 0u;                                                                                     \
        rpo += need ? 1u : 0u;                                                                                         \
        fetched = ring_read(rpo);                                                                                      \
    }
    // Once per group of four steps, in this order (vmcnt counts loads AND stores, in order, so
    // the one wait of a group must find nothing younger than a group in flight):
    //   RETIRE  wait for the chunk requested a group ago and move it into the ring;
    //   STORE   the four symbols the previous group produced;
    //   REQUEST the next chunk.
#define REDUX_DEC_RETIRE ring_write(pend_chunk, ldq.x, ldq.y, ldq.z, ldq.w);
    // Output: a finished group's dword is staged; 16-byte aligned blocks get one 16-byte store per
    // four groups (p is wave-uniform, so that is a scalar branch).  A 4-byte store every four steps
    // per lane is what the L2's background cleaning of resident dirty lines turns into ten times
    // the output in fabric writes (WRITE_SIZE 43e6 KiB for 4 GiB).
#define REDUX_DEC_STORE                                                                                                \
    if (aligned16) {                                                                                                   \
        if ((int32_t)S.dflag >= 0 && p > staged) { /* a live lane has emitted p symbols */                             \
            oq     = make_uint4(oq.y, oq.z, oq.w, S.obuf);                                                             \
            S.obuf = 0;                                                                                                \
            staged = p;                                                                                                \
            if ((p & 15u) == 0) {                                                                                      \
                *reinterpret_cast<uint4 *>(dst + (p - 16)) = oq;                                                       \
                stored = p;                                                                                            \
            }                                                                                                          \
        }                                                                                                              \
    } else if (aligned4 && (int32_t)S.dflag >= 0 && p > stored) {                                                      \
        *reinterpret_cast<uint32_t *>(dst + (p - 4)) = S.obuf;                                                         \
        S.obuf = 0;                                                                                                    \
        stored = p;                                                                                                    \
        staged = p;                                                                                                    \
    }
#define REDUX_DEC_REQUEST                                                                                              \
    {                                                                                                                  \
        /* chunk wr when its ring slot is free, otherwise chunk wr-1 once more (it lands on its own copy): the 16-byte \
           load is unconditional, so no join of an exec-masked region makes the compiler wait for a load it has just   \
           issued.  A chunk that crosses the end of its stream is read dword by dword with clamped indices in a rarely \
           entered block; the 16-byte load of such a lane reads the offsets table instead (always mapped). */          \
        const bool     room = (int32_t)(4u * wr - rpo) <= 28;                                                          \
        const uint32_t c_   = room ? wr : wr - 1u;                                                                     \
        const bool     tail = 4u * c_ + 3u > rpo_last;                                                                 \
        pend_chunk          = c_;                                                                                      \
        ldq = *reinterpret_cast<gptr4>(tail ? gsafe : gin + 4u * c_);                                                  \
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tail) != 0, 0)) {                                             \
            const uint32_t t0 = rd(4u * c_), t1 = rd(4u * c_ + 1u), t2 = rd(4u * c_ + 2u), t3 = rd(4u * c_ + 3u);      \
            ldq.x = tail ? t0 : ldq.x;                                                                                 \
            ldq.y = tail ? t1 : ldq.y;                                                                                 \
            ldq.z = tail ? t2 : ldq.z;                                                                                 \
            ldq.w = tail ? t3 : ldq.w;                                                                                 \
        }                                                                                                              \
        wr += room ? 1u : 0u;                                                                                          \
    }

    // ---------------- lock-step groups of four symbols ----------------
    // A step is computed AND committed for all 64 lanes without predication; the only things that can end a block here --
    // the EOF symbol (codec.rs:136-138) and a stream that runs dry in the renormalisation (bitio/mod.rs:107) -- are
    // terminal, so a lane they happen to records its result in a rarely entered block and from then on computes on
    // garbage that nobody reads (every address it forms stays inside the tables, its ring column and its stream buffer).
    // The phases of a step are fenced for the scheduler so that the bit reader's refill sits in the shadow of the
    // search's second LDS round trip.
    const uint32_t pfast = capn;
    uint32_t livemask = (int32_t)S.dflag < 0 ? 0x7FFFFFFFu : 0xFFFFFFFFu; // sign bit cleared once the block is finished
    uint32_t fin_obuf = 0, fin_cons = S.consumed;                         // what a lane that finishes in this loop ends with
    if (aligned4) {
        const double cdm1 = (double)(c_const - 1u), cd = (double)c_const;
#define REDUX_DEC_LEVEL(t)                                                                                             \
    left = __builtin_uadd_overflow(q, (t), &q2); /* carries exactly when the probe fails (rem < t: go left) */          \
    bits = __builtin_amdgcn_alignbit(bits, q2, 31);                                                                    \
    q    = q > q2 ? q : q2;                                                                                            \
    hq   = hq < q2 ? hq : q2;
        for (; p + 4 <= pfast; p += 4) {
            if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
                break;
            REDUX_DEC_RETIRE
            REDUX_DEC_STORE
            REDUX_DEC_REQUEST
#pragma unroll
            for (int K = 0; K < 4; K++) {
                const double   rc = rc_static;
                const uint32_t c  = c_const;
                // ---- A: code value
                const uint32_t R1  = (~(S.ihigh + S.low)) >> sh;
                const uint32_t Vd  = (S.W - S.low) >> sh;
                const double   R1d = (double)R1;
                const uint32_t v   = dec_value(R1d, Vd, cd, cdm1);
                // ---- B: get_symbol, first round
                uint32_t q = ~v, hq = q + cum256, bits = 0, q2;
                uint32_t eofq = hq; // top bit set: v >= count - 1 -> the EOF symbol (adaptive_tree.rs:116)
                bool     left;
                uint32_t sym = 0;
                uint32_t w16 = 0, w8 = 0, w24 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, n5 = 0, n6 = 0, n7 = 0;
                if constexpr (MODE == 2) {
                    eofq = cum256 - 1u - v;
                    sym  = lut[v & 0xFFFFu]; // (a finished lane's v is garbage: stay inside the table)
                } else {
                    REDUX_DEC_LEVEL(T.n128)
                    const uint32_t x6 = left ? T.n64 : T.n192, c5l = left ? T.n32 : T.n160, c5r = left ? T.n96 : T.n224;
                    REDUX_DEC_LEVEL(x6)
                    const uint32_t x5 = left ? c5l : c5r;
                    REDUX_DEC_LEVEL(x5)
                    const uint32_t *r = lds + bits * 33u; // prefix i = bits << 5 at dword i + (i >> 5)
                    w16 = r[16]; w8 = r[8]; w24 = r[24];
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- C: second round
                uint32_t lo, hi;
                if constexpr (MODE == 2) {
                    lo = ctab[sym];
                    hi = ctab[sym + 1u];
                } else {
                    REDUX_DEC_LEVEL(w16)
                    const uint32_t x3 = left ? w8 : w24;
                    REDUX_DEC_LEVEL(x3)
                    const uint32_t *r4 = lds + (bits << 3) + (bits >> 2); // prefix i = bits << 3: nodes i .. i+7 inside one block of 32
                    d0 = r4[1]; d1 = r4[2]; d2 = r4[3]; d3 = r4[4]; n5 = r4[5]; n6 = r4[6]; n7 = r4[7];
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- C's shadow: the bit reader's refill for this step (bitio/mod.rs:78-120), the factor both ends of the
                // new interval share (codec.rs:133-134)
                asm volatile("" : "+v"(S.bcnt));
                REDUX_DEC_READER
                double Y = __builtin_fma(R1d, rc, rc);
                asm volatile("" : "+v"(Y), "+v"(S.bbits), "+v"(S.bcnt));
                __builtin_amdgcn_sched_barrier(0);
                // ---- D: last round, narrowing + renormalisation (codec.rs:133-161)
                if constexpr (MODE == 1) {
                    REDUX_DEC_LEVEL(d3)                  // node i+4
                    const uint32_t e1  = left ? d1 : n6; // level 1: node i+2 or i+6
                    const uint32_t e0l = left ? d0 : n5, e0r = left ? d2 : n7; // level 0: i+1 / i+5 or i+3 / i+7
                    REDUX_DEC_LEVEL(e1)
                    const uint32_t x0 = left ? e0l : e0r;
                    REDUX_DEC_LEVEL(x0)
                }
                if constexpr (MODE != 2) {
                    sym = bits & 0xFFu;
                    lo  = v + q + 1u;  // v - rem = cum(s)
                    hi  = v + hq + 1u; // cum(s + 1): the upper boundary of the last level that went left
                }
                const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, lo, c) << sh);
                const uint32_t nihigh = 0u - (S.low + (scale_div<false, true>(R1, Y, hi, c) << sh));
                const uint32_t xx     = ~(nlow ^ nihigh);
                uint32_t       k;
                asm("v_ffbh_u32 %0, %1" : "=v"(k) : "v"(xx)); // (32-bit codes: low != high while count < 2^17; narrower ones: the
                                                              // padding below the code differs, so k <= code_bits)
                const uint32_t low2  = nlow << (k & 31u);
                const uint32_t ih2   = nihigh << (k & 31u);
                const uint32_t t2    = (low2 & ih2) << 1;
                const uint32_t j     = (uint32_t)__builtin_clz(~t2);
                const uint32_t n     = k + j; // bits pulled by get_bit (codec.rs:157)
                const uint32_t cons0 = S.consumed, cons2 = cons0 + n;
                const uint32_t e     = (eofq | (stream_bits - cons2)) & livemask;
                // ---- E: commit, every lane
                S.low      = (low2 << j) & 0x7FFFFFFFu;
                S.ihigh    = (ih2 << j) & 0x7FFFFFFFu;
                S.consumed = cons2;
                // [value | next 32 bits] << k, keep the top bit, << j, put it back (codec.rs:143-157).  A narrow code whose
                // interval collapsed (k == code_bits) takes that top bit from the new bits: both shifts are 64-bit there.
                const uint32_t nxt  = (uint32_t)(S.bbits >> 32);
                const uint64_t comb = CB32 ? (((uint64_t)S.W << 32) | nxt) : (((uint64_t)S.W << 32) | ((uint64_t)nxt << sh));
                const uint32_t h2   = (uint32_t)((comb << n) >> 32);
                const uint32_t h1   = CB32 ? S.W << k : (uint32_t)((comb << k) >> 32);
                S.W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
                S.bbits <<= n;
                S.bcnt -= n;
                const uint32_t obuf0 = S.obuf;
                S.obuf = obuf0 | (sym << (8 * K));
                // ---- the two ways a block ends here
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)e < 0) != 0, 0)) { // one scalar branch; selects inside, no exec masking
                    const bool fin        = (int32_t)e < 0;
                    const bool eof_symbol = (int32_t)eofq < 0; // decided first: decompress_symbol returns before renormalising
                    S.st     = fin && !eof_symbol ? REDUX_EOF : S.st;
                    fin_cons = fin ? (eof_symbol ? cons0 : cons2) : fin_cons;
                    fin_obuf = fin ? obuf0 : fin_obuf;
                    S.n_out  = fin ? p + K : S.n_out;
                    S.dflag  = fin ? 0x80000000u : S.dflag;
                    livemask = fin ? 0x7FFFFFFFu : livemask;
                }
            }
        }
#undef REDUX_DEC_LEVEL
    }
    if ((int32_t)S.dflag < 0) { // finished in the loop above (or never live): what the garbage steps since then did not touch
        S.obuf     = fin_obuf;
        S.consumed = fin_cons;
    }
    // ---------------- remaining steps (EOF symbol, frozen model, unaligned output) ----------------
    for (;; p++) {
        if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
            break;
        const double   rc  = rc_static;
        const uint32_t c   = c_const;
        if ((p & 3) == 0) {
            REDUX_DEC_RETIRE
            REDUX_DEC_STORE
            REDUX_DEC_REQUEST
        }
        REDUX_DEC_READER
        const uint32_t R1  = (~(S.ihigh + S.low)) >> sh;
        const uint32_t Vd  = (S.W - S.low) >> sh;
        const double   R1d = (double)R1;
        const uint32_t v   = dec_value(R1d, Vd, (double)c, (double)(c - 1u));
        const DecFound f   = REDUX_DEC_SEARCH(v);
        dec_commit_careful<CB32>(S, f, R1, R1d, rc, c, sh, stream_bits, p, p < capn, aligned4, dst, [](uint32_t) {});
    }
#undef REDUX_DEC_READER
#undef REDUX_DEC_RETIRE
#undef REDUX_DEC_STORE
#undef REDUX_DEC_REQUEST
#undef REDUX_DEC_SEARCH
    if (live) {
        if (aligned4) {
            // the 0..3 staged dwords (oldest first: the last k components of oq), then the partial one
            const uint32_t k = (staged - stored) >> 2;
            const uint32_t comp[4] = {oq.x, oq.y, oq.z, oq.w};
            for (uint32_t j = 0; j < k; j++) {
                const uint32_t idx = 4 - k + j;
                const uint32_t w   = idx == 0 ? comp[0] : idx == 1 ? comp[1] : idx == 2 ? comp[2] : comp[3];
                *reinterpret_cast<uint32_t *>(dst + stored + 4 * j) = w;
            }
            for (uint32_t i = staged; i < S.n_out; i++)
                dst[i] = (uint8_t)(S.obuf >> (8 * (i & 3)));
        }
        a.out_sizes[blk] = S.n_out;
        a.status[blk]    = S.st;
        if (a.in_used) { // the reader fetches whole bytes, and never past the end of the stream
            const uint64_t used = ((uint64_t)S.consumed + 7) / 8;
            a.in_used[blk]      = used < size ? used : size;
        }
    }
}

} // namespace redux
