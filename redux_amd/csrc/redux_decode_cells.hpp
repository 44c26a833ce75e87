// redux_decode_cells.hpp -- k_decode_cells: the lock-step decoder for the symbol widths other than 8 (symbol_bits 1 .. 12,
// code_bits <= 32), in the form k_decode_lock gave the 8-bit decoder (gfx950 only).  SURVEY section 8(f).3; the widths
// src/model/tests.rs:95-251 exercises (4 and 12) and everything between.
//
// decompress_stream (codec.rs:164-176) of LANES blocks per wave, one lane per block, all lanes on the same symbol index, so
// the model's total -- 2^symbol_bits + 1 + symbols decoded, until the freq_max freeze (adaptive_tree.rs:84) -- is
// wave-uniform and the divisions of codec.rs:131-134 are multiplications by reciprocals.  What is new against the
// per-level walk of rounds 2 and 3 (k_decode_gen: symbol_bits DEPENDENT probes per symbol):
//
//   * the Fenwick tree of adaptive_tree.rs:36-48 as CELLS of four levels.  Group g (g = 0: levels 3..0, g = 1: levels
//     7..4, g = 2: levels 11..8) holds one cell per prefix of the symbol bits above it; a cell is the fifteen nodes
//     (16 c + j) << 4g, j = 1 .. 15, under its prefix c -- so get_symbol's descent (adaptive_tree.rs:115-127) is ONE
//     load per four levels instead of four dependent ones.  The topmost cell is partial when symbol_bits is not a multiple
//     of four: the descent enters it at its level symbol_bits mod 4 - 1 and only ever touches its left-most nodes.
//   * nodes hold the tree value itself (lowbit + increments, as the reference's tree[] does): a probe is one add whose
//     wrap-around is the outcome (q = ~rem, q2 = q + t = ~(rem - t): top bit set = go right; redux_decode.hpp), and the
//     high end of the range falls out of the same probes (min over q2, seeded with the virtual root probe against
//     tree[2^symbol_bits] = count - 1, whose sign is the EOF test of adaptive_tree.rs:116).
//     symbol_bits >= 9: u16 nodes, 32-byte cells: two 16-byte halves (n1|n2, n3|n4, n5|n6, n7|n8), (n9|n10, n11|n12,
//     n13|n14, n15|-).  symbol_bits <= 7: u32 nodes (a 64 KiB block of 4-bit symbols counts to 131,072), 64-byte cells in
//     four pieces (n1..n4), (n5..n8), (n9..n12), (n13..n15, -).
//   * update(s + 1) (adaptive_tree.rs:83-92) increments a level's node exactly where the descent went LEFT, and having
//     gone left at the cell's top level is what puts the rest of the path into the same half as the top node: a cell's
//     share of the update is the half the descent already holds in registers plus a packed addend built from the
//     descent's own masks, written back with ONE 16-byte store (u32 nodes: two).  No atomics: a lane owns its cells.
//   * where the cells live.  Groups above the bottom one: LDS, lane l owning 16 bytes of every piece row (conflict-free
//     for any per-lane cell).  The bottom group (2^(symbol_bits - 4) cells per block: 8 KiB for 12-bit symbols) in LDS
//     too while 64 blocks' worth fits -- symbol_bits <= 10 -- and otherwise (GLOBAL0: 11- and 12-bit symbols) in the
//     workspace, block-major, one 32-byte sector per cell: a step is then two LDS round trips and ONE global round trip
//     where round 3's k_decode_gen<12, false> made twelve dependent global probes and twelve global atomics, and 64
//     blocks per wave, four waves per CU stay in flight.  (All-LDS forms with fewer lanes per wave -- 16 for 12-bit
//     symbols, 32 for 11 -- were built and measured: never faster from 2,048 blocks up, half the speed where they need
//     a second pass over the chip; the template still takes LANES < 64.)
//   * everything else as k_decode_lock: the stream through a ring of 16 dwords per lane in LDS fed by one unconditional
//     16-byte load per four steps; the code value by v_rcp_f64 + one exact remainder; closed-form renormalisation; every
//     lane computes and commits every step, a lane that ends (EOF symbol, codec.rs:136-138; stream exhausted,
//     bitio/mod.rs:107) records that in a block entered on a wave-level ballot and its later symbols are masked out of
//     the output image; U steps per loop turn, U * symbol_bits a multiple of 32, so the turn's symbols leave as whole
//     dwords at static bit positions (write_bits(symbol, symbol_bits) MSB-first, bitio/mod.rs:148-181).
//   * the lock-step loop runs while every step has room for its symbol, the count stays below 2^17 (no quotient fix-up,
//     scale_div) and the output is 4-byte aligned; the last < U symbols, the EOF symbol, a symbol that does not fit
//     (OutputTooSmall is decided byte by byte, codec.rs:171) are a per-lane loop over the same cells, one probe per level.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_decode.hpp"
#include "redux_gen.hpp"

namespace redux {

typedef uint32_t cl_u32x4 __attribute__((ext_vector_type(4)));

template <int SB, int LANES, bool GLOBAL0>
struct CellGeom {
    static_assert(SB >= 1 && SB <= 12, "symbol widths of the lock-step kernels"); // (8: blocks above 64 KiB, which k_decode_lock's u16 nodes do not hold)
    static_assert(!GLOBAL0 || SB > 8, "only the u16 bottom cells go to the workspace");
    static constexpr bool     kU32       = SB <= 8;            // node type
    static constexpr int      kGroups    = (SB + 3) / 4;
    static constexpr int      kTopLevels = SB - 4 * (kGroups - 1);
    static constexpr uint32_t kCellBytes = kU32 ? 64u : 32u;
    static constexpr uint32_t kPieces    = kCellBytes / 16u;   // 16-byte pieces of a cell
    static constexpr uint32_t kPiece     = 16u * LANES;        // LDS bytes of one piece row (all lanes)
    static constexpr uint32_t kCellPitch = kPieces * kPiece;   // LDS bytes of a cell row
    static constexpr uint32_t kSteps     = SB >= 9 ? (SB == 12 ? 8u : SB == 10 ? 16u : 32u) : 32u; // per loop turn
    static constexpr uint32_t kImgDwords = kSteps * SB / 32u;
    static_assert(kSteps * SB % 32u == 0, "a turn's symbols are whole dwords");
    static constexpr uint32_t cells(int g) { return g == kGroups - 1 ? 1u : 1u << (SB - 4 * (g + 1)); }
    static constexpr int      levels(int g) { return g == kGroups - 1 ? kTopLevels : 4; }
    // LDS regions, topmost group first
    static constexpr uint32_t region(int g)
    {
        uint32_t o = 0;
        for (int h = kGroups - 1; h > g; h--)
            o += cells(h) * kCellPitch;
        return o;
    }
    static constexpr uint32_t kRingBase  = region(GLOBAL0 ? 0 : -1);
    static constexpr uint32_t kLdsBytes  = kRingBase + 16u * 4u * LANES;
    static constexpr uint32_t kTreeBytes = GLOBAL0 ? cells(0) * kCellBytes : 0u; // per block, in the workspace
    static_assert(kLdsBytes <= 160u * 1024u, "LDS of a CU");
};

// every node = its lowbit (all frequencies 1, adaptive_tree.rs:43-45): the bottom cells of n blocks in the workspace
__global__ void k_fill_cells16(cl_u32x4 *cells, uint64_t npieces)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npieces)
        cells[i] = (i & 1u) ? cl_u32x4{0x00020001u, 0x00040001u, 0x00020001u, 0x00000001u}
                            : cl_u32x4{0x00020001u, 0x00040001u, 0x00020001u, 0x00080001u};
}

// the descent's running state (redux_decode.hpp, "Pieces shared by the lock-step decoders")
struct CellDescent {
    uint32_t q, hq, bits;
};
#define REDUX_CL_STEP(t)                                                                                               \
    q2     = D.q + (t);                                                                                                \
    D.bits = __builtin_amdgcn_alignbit(D.bits, q2, 31);                                                                \
    D.q    = D.q > q2 ? D.q : q2;                                                                                      \
    D.hq   = D.hq < q2 ? D.hq : q2;
#define REDUX_CL_MASK(m)                                                                                               \
    m = (uint32_t)((int32_t)q2 >> 31); /* all ones: the probe succeeded, go right */                                   \
    asm("" : "+v"(m))                  /* (opaque: the compiler would turn the picks back into compare + v_cndmask) */
#define REDUX_CL_PICK(m, l, r) (((m) & (r)) | (~(m) & (l)))

// LV levels of a u16 cell: h0 = (n1|n2, n3|n4, n5|n6, n7|n8), h1 = (n9|n10, n11|n12, n13|n14, n15|-).  Leaves the half the
// path lies in, with update(s + 1)'s increments added, in `neu`, and m3 = all ones when that is h1.  u1 / u10000: 1 and
// 0x10000 while the model updates, 0 once it is frozen (wave-uniform).
template <int LV>
__device__ __forceinline__ void cell16_descend(CellDescent &D, const cl_u32x4 &h0, const cl_u32x4 &h1, uint32_t u1, uint32_t u10000,
                                               cl_u32x4 &neu, uint32_t &m3)
{
    uint32_t q2, m2 = 0, m1 = 0;
    cl_u32x4 e = h0;
    m3         = 0;
    if (LV >= 4) {
        REDUX_CL_STEP(h0.w >> 16) // n8
        REDUX_CL_MASK(m3);
        e = cl_u32x4{REDUX_CL_PICK(m3, h0.x, h1.x), REDUX_CL_PICK(m3, h0.y, h1.y), REDUX_CL_PICK(m3, h0.z, h1.z),
                     REDUX_CL_PICK(m3, h0.w, h1.w)};
    }
    uint32_t u = e.x, w = e.y; // (n1|n2), (n3|n4) of the half
    if (LV >= 3) {
        REDUX_CL_STEP(e.y >> 16) // n4
        REDUX_CL_MASK(m2);
        u = REDUX_CL_PICK(m2, e.x, e.z);
        w = REDUX_CL_PICK(m2, e.y, e.w);
    }
    uint32_t x0 = u;
    if (LV >= 2) {
        REDUX_CL_STEP(u >> 16) // n2 / n6
        REDUX_CL_MASK(m1);
        x0 = REDUX_CL_PICK(m1, u, w);
    }
    REDUX_CL_STEP(x0 & 0xFFFFu) // n1 / n3 / n5 / n7
    // increments: +1 where the path went left (bit clear).  Fields of the half: (n1|n2, n3|n4, n5|n6, n7|n8)
    const uint32_t r0 = D.bits & 1u;
    const uint32_t c0 = (LV >= 2 ? u10000 : 0u) | (u1 & ~r0); // n1-type leaf in the low field, its level-1 parent in the high one
    const uint32_t c1 = m1 & u1 & ~r0;                       // n3-type leaf
    neu.x = e.x + (~m2 & ~m1 & c0);
    neu.y = e.y + (LV >= 2 ? (~m2 & (c1 | (LV >= 3 ? u10000 : 0u))) : 0u);
    neu.z = e.z + (LV >= 3 ? (m2 & ~m1 & c0) : 0u);
    neu.w = e.w + (LV >= 3 ? ((m2 & c1) | (LV >= 4 ? (~m3 & u10000) : 0u)) : 0u);
}

// the same over u32 nodes: pieces p0 = (n1..n4), p1 = (n5..n8), p2 = (n9..n12), p3 = (n13..n15, -); the octet the path
// lies in comes back as two pieces (its n1..n4 and n5..n8)
template <int LV>
__device__ __forceinline__ void cell32_descend(CellDescent &D, const cl_u32x4 &p0, const cl_u32x4 &p1, const cl_u32x4 &p2,
                                               const cl_u32x4 &p3, uint32_t u1, cl_u32x4 &neu0, cl_u32x4 &neu1, uint32_t &m3)
{
    uint32_t q2, m2 = 0, m1 = 0;
    cl_u32x4 a = p0, b = p1; // (n1..n4), (n5..n8) of the octet
    m3         = 0;
    if (LV >= 4) {
        REDUX_CL_STEP(p1.w) // n8
        REDUX_CL_MASK(m3);
        a = cl_u32x4{REDUX_CL_PICK(m3, p0.x, p2.x), REDUX_CL_PICK(m3, p0.y, p2.y), REDUX_CL_PICK(m3, p0.z, p2.z),
                     REDUX_CL_PICK(m3, p0.w, p2.w)};
        b = cl_u32x4{REDUX_CL_PICK(m3, p1.x, p3.x), REDUX_CL_PICK(m3, p1.y, p3.y), REDUX_CL_PICK(m3, p1.z, p3.z),
                     REDUX_CL_PICK(m3, p1.w, p3.w)};
    }
    uint32_t f1 = a.x, f2 = a.y, f3 = a.z;
    if (LV >= 3) {
        REDUX_CL_STEP(a.w) // n4
        REDUX_CL_MASK(m2);
        f1 = REDUX_CL_PICK(m2, a.x, b.x);
        f2 = REDUX_CL_PICK(m2, a.y, b.y);
        f3 = REDUX_CL_PICK(m2, a.z, b.z);
    }
    uint32_t g1 = f1;
    if (LV >= 2) {
        REDUX_CL_STEP(f2) // n2 / n6
        REDUX_CL_MASK(m1);
        g1 = REDUX_CL_PICK(m1, f1, f3);
    }
    REDUX_CL_STEP(g1)
    const uint32_t r0 = D.bits & 1u;
    const uint32_t l0 = u1 & ~r0; // the leaf's increment
    neu0.x = a.x + (~m2 & ~m1 & l0);
    neu0.y = a.y + (LV >= 2 ? (~m2 & ~m1 & u1) : 0u);
    neu0.z = a.z + (LV >= 2 ? (~m2 & m1 & l0) : 0u);
    neu0.w = a.w + (LV >= 3 ? (~m2 & u1) : 0u);
    neu1.x = b.x + (LV >= 3 ? (m2 & ~m1 & l0) : 0u);
    neu1.y = b.y + (LV >= 3 ? (m2 & ~m1 & u1) : 0u);
    neu1.z = b.z + (LV >= 3 ? (m2 & m1 & l0) : 0u);
    neu1.w = b.w + (LV >= 4 ? (~m3 & u1) : 0u);
}
#undef REDUX_CL_STEP
#undef REDUX_CL_MASK
#undef REDUX_CL_PICK

// The cells of one lane's block: the lock-step loop's loads and stores, and per-node access for the per-lane loop.
template <int SB, int LANES, bool GLOBAL0>
struct CellTree {
    typedef CellGeom<SB, LANES, GLOBAL0> G;
    char *lds;  // + 16 * lane: this lane's column
    char *glob; // GLOBAL0: this block's bottom cells in the workspace

    __device__ __forceinline__ void init(uint32_t *mem, uint32_t lane, void *trees, uint64_t blk)
    {
        lds  = reinterpret_cast<char *>(mem) + 16u * lane;
        glob = GLOBAL0 ? reinterpret_cast<char *>(trees) + blk * G::kTreeBytes : nullptr;
    }
    // every node = its lowbit (all frequencies 1, adaptive_tree.rs:43-45).  (One wave per workgroup and every lane owns
    // its columns: no barrier.  The workspace cells are filled by k_fill_cells16.)
    __device__ __forceinline__ void fill() const
    {
#pragma unroll
        for (int g = G::kGroups - 1; g >= (GLOBAL0 ? 1 : 0); g--) {
            const uint32_t s4 = 4u * g;
            for (uint32_t c = 0; c < G::cells(g); c++) {
                char *cell = lds + G::region(g) + c * G::kCellPitch;
                if (G::kU32) {
                    *reinterpret_cast<cl_u32x4 *>(cell)                 = cl_u32x4{1u << s4, 2u << s4, 1u << s4, 4u << s4};
                    *reinterpret_cast<cl_u32x4 *>(cell + G::kPiece)     = cl_u32x4{1u << s4, 2u << s4, 1u << s4, 8u << s4};
                    *reinterpret_cast<cl_u32x4 *>(cell + 2 * G::kPiece) = cl_u32x4{1u << s4, 2u << s4, 1u << s4, 4u << s4};
                    *reinterpret_cast<cl_u32x4 *>(cell + 3 * G::kPiece) = cl_u32x4{1u << s4, 2u << s4, 1u << s4, 0u};
                } else {
                    *reinterpret_cast<cl_u32x4 *>(cell) =
                        cl_u32x4{0x00020001u << s4, 0x00040001u << s4, 0x00020001u << s4, 0x00080001u << s4};
                    *reinterpret_cast<cl_u32x4 *>(cell + G::kPiece) =
                        cl_u32x4{0x00020001u << s4, 0x00040001u << s4, 0x00020001u << s4, 0x00000001u << s4};
                }
            }
        }
    }
    // where Fenwick node e (1 .. 2^SB - 1) lives: its group is ctz(e) / 4, its cell the bits above the group's four
    __device__ __forceinline__ char *node_ptr(uint32_t e) const
    {
        const uint32_t g  = (uint32_t)__builtin_ctz(e) >> 2;
        const uint32_t j  = (e >> (4u * g)) & 15u, c = e >> (4u * g + 4u);
        const uint32_t fb = G::kU32 ? (j - 1u) * 4u : (j - 1u) * 2u; // byte of the field inside the cell
        if (GLOBAL0 && g == 0)
            return glob + c * G::kCellBytes + fb;
        uint32_t reg = 0; // region(g), g a run-time value here
#pragma unroll
        for (int h = G::kGroups - 1; h >= 0; h--)
            reg = (uint32_t)h == g ? G::region(h) : reg;
        return lds + reg + c * G::kCellPitch + (fb >> 4) * G::kPiece + (fb & 15u);
    }
    // tree[e] of the reference (lowbit + increments)
    __device__ __forceinline__ uint32_t full(uint32_t e) const
    {
        const char *p = node_ptr(e);
        return G::kU32 ? *reinterpret_cast<const uint32_t *>(p) : *reinterpret_cast<const uint16_t *>(p);
    }
    __device__ __forceinline__ void bump(uint32_t e) const
    {
        char *p = node_ptr(e);
        if (G::kU32)
            *reinterpret_cast<uint32_t *>(p) += 1u;
        else
            *reinterpret_cast<uint16_t *>(p) += 1u;
    }
};

// FIX (u32 nodes only): the count may pass 2^17 inside a block -- symbol widths <= 7 in blocks of more than 2^17 symbols with
// a model that does not freeze below that.  The code value is then the quotient of a numerator of up to 62 bits: estimated
// with a Newton-refined reciprocal and settled by the exact 64-bit remainder (as k_decode_wave's), the two ends of the new
// interval take scale_div's fix-up, and the interval may collapse to low == high (k = 32: 64-bit shifts).  ~25 more
// instructions per step; without it such blocks ran on the one-lane kernels.
template <int SB, int LANES, bool GLOBAL0, bool FIX = false>
__global__ void __launch_bounds__(64) k_decode_cells(GenDecArgs a)
{
    static_assert(!FIX || SB <= 8, "the fix-up variant is for the u32-node widths");
    typedef CellGeom<SB, LANES, GLOBAL0> G;
    __shared__ __attribute__((aligned(16))) uint32_t lds[G::kLdsBytes / 4];
    // One wave per SIMD, by construction ("placement"): a lock-step wave that shares its SIMD takes
    // ~1.6 x as long.  The LDS admits up to four of these workgroups on a CU; claiming an accumulation register beyond the
    // half-file mark makes the descriptor ask for more than 256 registers.
    asm volatile("" ::: "a255");
    const uint32_t lane = threadIdx.x; // (blockDim.x == LANES: a wave of fewer live lanes simply has a shorter exec mask)
    const uint64_t blk  = (uint64_t)blockIdx.x * LANES + lane;
    const bool     live = blk < a.nblocks;
    constexpr uint32_t kCount0 = (1u << SB) + 1u;
    constexpr uint32_t kMask   = (1u << SB) - 1u;

    CellTree<SB, LANES, GLOBAL0> T;
    T.init(lds, lane, a.trees, blk);
    T.fill();

    const uint32_t cb = a.code_bits, sh = 32 - cb;
    uint64_t       size = 0;
    const uint8_t *sp   = a.in;
    if (live) {
        const uint64_t o0 = a.in_offsets[blk];
        size              = a.in_offsets[blk + 1] - o0;
        sp                = a.in + o0;
    }
    const uint32_t stream_bits = (uint32_t)(size * 8);
    uint8_t       *dst         = a.out + (live ? blk : 0) * (uint64_t)a.block_size;
    const uint32_t capn        = a.block_size; // bytes
    const uint32_t nfreeze     = a.nfreeze;
    const rc_ptr   rcp         = (rc_ptr)a.rc;

    // ---- stream side: ring of 16 dwords per lane in LDS, as k_decode_lock (redux_decode_adaptive.hpp) ----------------
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    typedef const __attribute__((address_space(1))) cl_u32x4 *gptr4;
    const bool      has      = live && size > 0;
    const uintptr_t sp_abs   = (uintptr_t)sp;
    const gptr      gin      = has ? (gptr)(sp_abs & ~(uintptr_t)3) : (gptr)(uintptr_t)a.in_offsets;
    const uint32_t  rpo_last = has ? (uint32_t)(((((sp_abs + size + 3) & ~(uintptr_t)3) - (sp_abs & ~(uintptr_t)3)) >> 2) - 1) : 0u;
    const uint32_t  skip     = has ? (uint32_t)(sp_abs & 3) * 8 : 0u;
    const gptr      gsafe    = (gptr)(uintptr_t)a.in_offsets;
    char *const     ring     = reinterpret_cast<char *>(lds) + G::kRingBase + 4u * lane;
    auto rd = [&](uint32_t o) { return gin[o < rpo_last ? o : rpo_last]; };
    auto ring_write = [&](uint32_t chunk, const cl_u32x4 &x) {
        uint32_t *q = reinterpret_cast<uint32_t *>(ring + (chunk & 3u) * (16u * LANES));
        q[0] = x.x; q[LANES] = x.y; q[2 * LANES] = x.z; q[3 * LANES] = x.w;
    };
    auto ring_read = [&](uint32_t d) { return *reinterpret_cast<const uint32_t *>(ring + (d & 15u) * (4u * LANES)); };
    uint32_t rpo = 2, wr = 0, pend_chunk = 0;
    cl_u32x4 ldq = {0, 0, 0, 0};
    uint64_t bbits;
    uint32_t bcnt;
    {
        uint32_t d0 = 0, d1 = 0;
        for (; wr < 3; wr++) {
            ldq = cl_u32x4{rd(4 * wr), rd(4 * wr + 1), rd(4 * wr + 2), rd(4 * wr + 3)};
            if (wr == 0) {
                d0 = has ? __builtin_bswap32(ldq.x) : 0u;
                d1 = (has && rpo_last >= 1) ? __builtin_bswap32(ldq.y) : 0u;
            }
            ring_write(wr, ldq);
            pend_chunk = wr; // (the first group "retires" chunk 2 once more)
        }
        bbits = (((uint64_t)d0 << 32) | d1) << skip;
        bcnt  = 64 - skip;
    }
    uint32_t fetched = ring_read(rpo);
    auto retire = [&]() { ring_write(pend_chunk, ldq); };
    auto request = [&]() {
        const bool     room = (int32_t)(4u * wr - rpo) <= 12;
        const uint32_t c    = room ? wr : wr - 1u;
        const bool     tail = 4u * c + 3u > rpo_last;
        pend_chunk          = c;
        // (unconditional: a load inside an exec-masked region that shares a join with a rarely entered block gets the
        // compiler's vmcnt(0) behind it; a chunk that crosses its stream's end is patched by selects)
        ldq = *reinterpret_cast<gptr4>(tail ? gsafe : gin + 4u * c);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tail) != 0, 0)) {
            const cl_u32x4 t = cl_u32x4{rd(4u * c), rd(4u * c + 1u), rd(4u * c + 2u), rd(4u * c + 3u)};
            ldq.x = tail ? t.x : ldq.x;
            ldq.y = tail ? t.y : ldq.y;
            ldq.z = tail ? t.z : ldq.z;
            ldq.w = tail ? t.w : ldq.w;
        }
        wr += room ? 1u : 0u;
    };

    uint32_t W        = (uint32_t)((bbits >> 1) >> (63 - cb)) << sh; // codec.rs:124-127
    bbits <<= cb;
    bcnt -= cb;
    uint32_t consumed = cb;
    uint32_t low      = 0;
    int32_t  st       = REDUX_OK;
    uint32_t dflag    = live ? 0u : 0x80000000u; // sign: the block is finished
    if (live && consumed > stream_bits) {        // stream shorter than code_bits: Err(Eof) at once
        st    = REDUX_EOF;
        dflag = 0x80000000u;
    }
    uint32_t n_out = 0; // symbols emitted when the block finished
    uint32_t p     = 0;
    uint32_t r1    = 0xFFFFFFFFu; // high - low, left-aligned

    // ---------------- lock-step turns of kSteps symbols ----------------
    const uint32_t pend    = (uint32_t)(((uint64_t)capn * 8) / SB);          // symbols a block has room for
    const uint32_t pnofix  = (1u << 17) - kCount0;                          // count < 2^17 while p < pnofix, or for good if it freezes below
    const uint32_t pfast   = (FIX || nfreeze < pnofix || pend < pnofix) ? pend : pnofix;
    const bool     aligned = (((uintptr_t)a.out | a.block_size) & 3u) == 0;
    if (aligned) {
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f64x4 *grc4;
        const grc4 rcv = (grc4)(uintptr_t)a.rc; // 256-byte aligned workspace; p is a multiple of 4 at a group's start
        f64x4      rcg = rcv[0], rcn;
        asm volatile("" : "+v"(rcg)); // arrived before the loop: no in-loop wait inherits this load
        const bool   freezes = nfreeze < pfast;
        const f64x4  rcF4 = rcv[freezes ? nfreeze >> 2 : 0u];
        const double rcF  = (nfreeze & 3) == 0 ? rcF4.x : (nfreeze & 3) == 1 ? rcF4.y : (nfreeze & 3) == 2 ? rcF4.z : rcF4.w;
        double   cdm1 = (double)(kCount0 - 1u), cd = (double)kCount0;
        uint32_t left = stream_bits - consumed; // stream bits not pulled yet: its sign is read_bits' Err(Eof) (bitio/mod.rs:107)
        uint32_t livemask = (int32_t)dflag < 0 ? 0x7FFFFFFFu : 0xFFFFFFFFu;
        uint32_t symmask  = (int32_t)dflag < 0 ? 0u : kMask;
        uint32_t fin_cons = consumed;
        uint32_t R1  = r1 >> sh;
        double   R1d = (double)R1, xd = R1d + 1.0, rinv = __builtin_amdgcn_rcp(xd);
        if (FIX)
            rinv = __builtin_fma(__builtin_fma(-xd, rinv, 1.0), rinv, rinv);
        for (; p + G::kSteps <= pfast; p += G::kSteps) {
            if (__builtin_amdgcn_ballot_w64((int32_t)dflag >= 0) == 0)
                break;
            const bool was_live = (int32_t)dflag >= 0;
            uint32_t   img[G::kImgDwords];
#pragma unroll
            for (uint32_t d = 0; d < G::kImgDwords; d++)
                img[d] = 0;
#pragma unroll
            for (uint32_t Gi = 0; Gi < G::kSteps / 4; Gi++) {
                // once per group of four steps: chunk requested a group ago -> ring, next request, next reciprocals
                retire();
                request();
                {
                    const uint32_t pg = p + 4 * Gi < nfreeze ? p + 4 * Gi : nfreeze; // (the table ends 32 entries behind the freeze point)
                    rcn = rcv[(pg >> 2) + 1];
                }
#pragma unroll
                for (uint32_t K = 0; K < 4; K++) {
                    const uint32_t idx = p + 4 * Gi + K;
                    const bool     upd = idx < nfreeze; // (wave-uniform) this step updates the model
                    const double   rc  = upd ? rcg[K] : rcF;
                    const uint32_t c   = kCount0 + (upd ? idx : nfreeze);
                    const uint32_t u1 = upd ? 1u : 0u, u10000 = upd ? 0x10000u : 0u;
                    // ---- the bit reader's refill (bitio/mod.rs:78-120)
                    {
                        uint32_t need = (uint32_t)((int32_t)(bcnt - 33u) >> 31); // all ones: refill
                        asm("" : "+v"(need));
                        const uint64_t add = (uint64_t)(__builtin_bswap32(fetched) & need) << ((32 - bcnt) & 63);
                        bbits |= add;
                        bcnt += need & 32u;
                        rpo -= need;
                        fetched = ring_read(rpo);
                    }
                    // ---- code value (codec.rs:129-131): dec_value() with the reciprocal of the range already at hand
                    const uint32_t Vd  = (W - low) >> sh;
                    uint32_t       v;
                    if (FIX) {
                        const uint64_t num = (uint64_t)Vd * c + (c - 1u); // (Vd+1)*c - 1: up to 62 bits
                        v                  = (uint32_t)((double)num * rinv);
                        const int64_t r    = (int64_t)(num - ((uint64_t)v * R1 + v));
                        v += r < 0 ? 0xFFFFFFFFu : ((uint64_t)r > (uint64_t)R1 ? 1u : 0u);
                    } else {
                        const double   nd  = __builtin_fma((double)Vd, cd, cdm1); // (Vd+1)*c - 1, exact (< 2^49)
                        const uint32_t v0p = (uint32_t)__builtin_fma(nd, rinv, -0x1p-6) + 1u;
                        const double   rem = __builtin_fma(-(double)v0p, xd, nd);
                        uint32_t fix = (uint32_t)((int32_t)(uint32_t)((uint64_t)__double_as_longlong(rem) >> 32) >> 31); // -1: v0p is one too many
                        asm("" : "+v"(fix));
                        v = v0p + fix;
                    }
                    // ---- get_symbol (adaptive_tree.rs:115-136), a cell per four levels, topmost first
                    CellDescent D;
                    D.q    = ~v;
                    D.hq   = D.q + (c - 1u);
                    D.bits = 0;
                    const uint32_t eofq = D.hq; // top bit set: v >= count - 1 -> the EOF symbol (adaptive_tree.rs:116)
                    double Y = 0;
#pragma unroll
                    for (int g = G::kGroups - 1; g >= 0; g--) {
                        const uint32_t cidx = g == G::kGroups - 1 ? 0u : D.bits; // the symbol bits above this group
                        uint32_t m3;
                        if (GLOBAL0 && g == 0) {
                            typedef __attribute__((address_space(1))) cl_u32x4 *gcell;
                            const gcell    cell = (gcell)(uintptr_t)(T.glob + cidx * 32u);
                            const cl_u32x4 h0 = cell[0], h1 = cell[1];
                            cl_u32x4       neu;
                            cell16_descend<4>(D, h0, h1, u1, u10000, neu, m3);
                            cell[m3 & 1u] = neu;
                        } else if (G::kU32) {
                            char *cell = T.lds + G::region(g) + cidx * G::kCellPitch;
                            const cl_u32x4 p0 = *reinterpret_cast<const cl_u32x4 *>(cell);
                            const cl_u32x4 p1 = G::levels(g) >= 3 ? *reinterpret_cast<const cl_u32x4 *>(cell + G::kPiece) : cl_u32x4{0, 0, 0, 0};
                            const cl_u32x4 p2 = G::levels(g) >= 4 ? *reinterpret_cast<const cl_u32x4 *>(cell + 2 * G::kPiece) : cl_u32x4{0, 0, 0, 0};
                            const cl_u32x4 p3 = G::levels(g) >= 4 ? *reinterpret_cast<const cl_u32x4 *>(cell + 3 * G::kPiece) : cl_u32x4{0, 0, 0, 0};
                            cl_u32x4       neu0, neu1;
                            if (g == G::kGroups - 1)
                                cell32_descend<G::kTopLevels>(D, p0, p1, p2, p3, u1, neu0, neu1, m3);
                            else
                                cell32_descend<4>(D, p0, p1, p2, p3, u1, neu0, neu1, m3);
                            char *half = cell + (m3 & (2u * G::kPiece));
                            *reinterpret_cast<cl_u32x4 *>(half) = neu0;
                            if (G::levels(g) >= 3)
                                *reinterpret_cast<cl_u32x4 *>(half + G::kPiece) = neu1;
                        } else {
                            char *cell = T.lds + G::region(g) + cidx * G::kCellPitch;
                            const cl_u32x4 h0 = *reinterpret_cast<const cl_u32x4 *>(cell);
                            const cl_u32x4 h1 = G::levels(g) >= 4 ? *reinterpret_cast<const cl_u32x4 *>(cell + G::kPiece) : cl_u32x4{0, 0, 0, 0};
                            cl_u32x4       neu;
                            if (g == G::kGroups - 1)
                                cell16_descend<G::kTopLevels>(D, h0, h1, u1, u10000, neu, m3);
                            else
                                cell16_descend<4>(D, h0, h1, u1, u10000, neu, m3);
                            *reinterpret_cast<cl_u32x4 *>(cell + (m3 & G::kPiece)) = neu;
                        }
                        if (g == G::kGroups - 1) { // off the chain: the factor both ends of the new interval share, the next count
                            Y = __builtin_fma(R1d, rc, rc);
                            const double inc = upd ? 1.0 : 0.0;
                            cdm1 += inc;
                            cd += inc;
                        }
                    }
                    const uint32_t sym = D.bits;
                    const uint32_t lo  = v + D.q + 1u;  // v - rem = cum(s)
                    const uint32_t hi  = v + D.hq + 1u; // cum(s + 1): the upper boundary of the last level that went left
                    // ---- narrowing and renormalisation (codec.rs:133-161), closed form as k_decode_lock
                    const uint32_t nlow   = low + (scale_div<FIX>(R1, Y, lo, c) << sh);
                    const uint32_t nihigh = 0u - (low + (scale_div<FIX, true>(R1, Y, hi, c) << sh));
                    const uint32_t xx     = ~(nlow ^ nihigh);
                    uint32_t       k;
                    asm("v_ffbh_u32 %0, %1" : "=v"(k) : "v"(xx)); // (32-bit codes: low != high while count < 2^17; narrower ones:
                                                                  // the padding below the code differs, so k <= code_bits)
                    if (FIX)
                        k = k < 32u ? k : 32u; // (low == high: v_ffbh's -1; all 32 bits are shared and shift out)
                    const uint32_t low2  = FIX ? (uint32_t)((uint64_t)nlow << k) : nlow << (k & 31u);
                    const uint32_t ih2   = FIX ? (uint32_t)((uint64_t)nihigh << k) : nihigh << (k & 31u);
                    const uint32_t t2    = (low2 & ih2) << 1;
                    const uint32_t j     = (uint32_t)__builtin_clz(~t2);
                    const uint32_t n     = k + j; // bits pulled by get_bit (codec.rs:157)
                    const uint32_t left0 = left, left2 = left0 - n;
                    const uint32_t e     = (eofq | left2) & livemask;
                    const uint32_t Ls    = low2 << j;
                    r1                   = ~(Ls + (ih2 << j));
                    low                  = Ls & 0x7FFFFFFFu;
                    left                 = left2;
                    // [value | next 32 bits] << k, keep the top bit, << j, put it back (codec.rs:143-157)
                    const uint32_t nxt  = (uint32_t)(bbits >> 32);
                    const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nxt << sh);
                    const uint32_t h2   = (uint32_t)((comb << n) >> 32);
                    const uint32_t h1   = (uint32_t)((comb << k) >> 32);
                    W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
                    bbits <<= n;
                    bcnt -= n;
                    R1   = r1 >> sh;
                    R1d  = (double)R1;
                    xd   = R1d + 1.0;
                    rinv = __builtin_amdgcn_rcp(xd);
                    if (FIX)
                        rinv = __builtin_fma(__builtin_fma(-xd, rinv, 1.0), rinv, rinv);
                    // ---- the two ways a block ends here
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)e < 0) != 0, 0)) { // one scalar branch; selects inside
                        const bool fin        = (int32_t)e < 0;
                        const bool eof_symbol = (int32_t)eofq < 0; // decided first: decompress_symbol returns before renormalising
                        st       = fin && !eof_symbol ? REDUX_EOF : st;
                        fin_cons = fin ? stream_bits - (eof_symbol ? left0 : left2) : fin_cons;
                        n_out    = fin ? idx : n_out;
                        dflag    = fin ? 0x80000000u : dflag;
                        livemask = fin ? 0x7FFFFFFFu : livemask;
                        symmask  = fin ? 0u : symmask;
                    }
                    // ---- write_bits(symbol, SB) (codec.rs:171): the turn's symbols sit MSB-first at static bit positions
                    {
                        const uint32_t sm = sym & symmask;
                        const uint32_t o = (4 * Gi + K) * SB, d = o >> 5, r = o & 31u;
                        if (r + SB <= 32)
                            img[d] |= sm << (32 - r - SB);
                        else {
                            img[d] |= sm >> (r + SB - 32);
                            img[d + 1] |= sm << (64 - r - SB);
                        }
                    }
                }
                rcg = rcn;
            }
            // the turn's bytes (a lane that finished inside the turn has zeros behind its last symbol: inside its block)
            if (was_live) {
                uint32_t *o = reinterpret_cast<uint32_t *>(dst + (uint64_t)p * SB / 8u); // (p * SB is a multiple of 32)
#pragma unroll
                for (uint32_t d = 0; d < G::kImgDwords; d++)
                    o[d] = __builtin_bswap32(img[d]);
            }
        }
        consumed = (int32_t)dflag < 0 ? fin_cons : stream_bits - left;
    }

    // ---------------- per-lane steps: the last symbols of a block, the EOF symbol, unaligned output ----------------
    // decompress_symbol statement by statement, one probe per level over the same cells; the bit reader restarts at the
    // consumed-bit count.
    bool     done  = (int32_t)dflag < 0;
    uint32_t high  = ~((~r1 - low) & 0x7FFFFFFFu);
    uint64_t obits = (uint64_t)(done ? n_out : p) * SB; // bits handed to write_bits so far: bytes [0, obits / 8) are in dst
    uint32_t oacc  = 0;                                 // the incomplete byte's bits, right-aligned (p * SB is a multiple of 32)
    uint64_t cons64 = consumed;
    BitIn    B;
    {
        const uint64_t skipb = done ? 0 : (consumed >> 3);
        B.init(sp + skipb, live && !done ? size - skipb : 0);
        if (!done)
            B.take(consumed & 7u);
    }
    for (;; p++) {
        if (__builtin_amdgcn_readfirstlane(__ballot(!done) == 0))
            break;
        const uint32_t nup = p < nfreeze ? p : nfreeze;
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
                uint32_t i = 0, rem = v, hb = c - 1u; // hb: cum of the last level that went left
                for (int b = SB - 1; b >= 0; b--) {
                    const uint32_t e  = i | (1u << b);
                    const uint32_t tv = T.full(e);
                    if (rem >= tv) {
                        i |= 1u << b;
                        rem -= tv;
                    } else {
                        hb = v - rem + tv;
                        if (p < nfreeze)
                            T.bump(e); // update(s + 1) increments exactly the nodes where the descent goes left
                    }
                }
                s  = i;
                lo = v - rem;
                hi = hb;
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
                cons64 += n;
                if (cons64 > (uint64_t)stream_bits) { // read_bits would hit Err(Eof) (bitio/mod.rs:107)
                    st   = REDUX_EOF;
                    done = true;
                } else {
                    const uint32_t nb   = B.take(n);
                    const uint64_t comb = ((uint64_t)W << 32) | ((uint64_t)nb << (32 + sh - n));
                    const uint64_t c1   = comb << k;
                    const uint64_t c2   = c1 << j;
                    W = (((uint32_t)(c2 >> 32) & 0x7FFFFFFFu) | ((uint32_t)(c1 >> 32) & 0x80000000u)) & (0xFFFFFFFFu << sh);
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
            const uint64_t used = (cons64 + 7) / 8;
            a.in_used[blk]      = used < size ? used : size;
        }
    }
}

} // namespace redux
