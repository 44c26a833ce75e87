// redux_decode_adaptive.hpp -- k_decode_lock: the default decoder (u16 trees, count < 2^17), gfx950 only.
//
// decompress_stream (codec.rs:164-176) of 64 blocks per wave, all lanes in lock-step, same results as k_decode.
// What shapes it: the model's tree (512 B per block) caps a CU at four groups of 64 blocks, so a decoder wave is ALONE
// on its SIMD and pays for every instruction slot itself (~4.1 cycles for a 4-byte encoding, ~5.2 for an 8-byte one,
// dependent or not; an LDS instruction ~6 plus the round trip when its result is needed).  Hence:
//
//   * tree layout made for the decoder.  Levels 7-5 (seven nodes) live in VGPRs.  Levels 4, 3 of the eight 32-symbol
//     groups sit in one 8-byte "B cell" per group -- nodes +16, +8, +24 -- and levels 2-0 in one 16-byte "octet cell"
//     per 8 symbols -- nodes +1 .. +7 --, so a descent is ONE ds_read_b64 + ONE ds_read_b128, and update()
//     (adaptive_tree.rs:83-92) of the five LDS levels is TWO ds_add_u64 (four u16 fields each; a field never carries
//     into its neighbour, see below).  36 KiB of tree + a 4 KiB stream ring = 40 KiB: four groups fill the CU.
//   * nodes hold the tree value itself (lowbit + increments, as the reference's tree[] does), so a probe is one
//     add-with-carry whose operand is a 16-bit half of a loaded dword (SDWA), no unpacking.  A u16 holds
//     lowbit + increments while increments <= 65519: the fast loop stops there, converts the LDS nodes to increments
//     and the predicated loop below (which adds the lowbit back, as round 1's decoder did) finishes the block.
//   * the descent keeps q = ~rem: for a node value t, q2 = q + t is ~(rem - t); its carry says "probe failed, go
//     left", the new q is max_u32(q, q2), and cum(s+1) is v + 1 + min_u32 over the levels of q2, seeded with the
//     virtual root probe against tree[256] = count - 1 (whose sign is the EOF test of adaptive_tree.rs:116).
//   * a step is computed AND committed for all 64 lanes without predication.  The two ways a block ends inside the
//     fast loop -- the EOF symbol (codec.rs:136-138), a stream that runs dry during renormalisation
//     (bitio/mod.rs:107) -- are terminal: the lane records its result in a rarely entered block and computes on
//     garbage from then on (every address it forms stays inside its own columns and its own stream).
//   * the step is laid out by hand (sched_barrier + opaque pins): the register levels' share of update() and the
//     bit reader's refill sit in the shadows of the two LDS round trips.
//   * sixteen steps per loop iteration: the decoded bytes of an iteration collect in four registers with static
//     roles and leave as one 16-byte store per lane; the stream comes through a ring in LDS fed by one 16-byte load
//     per lane and group of four steps, consumed a group later (a lock-step wave waits for the slowest of 64 lanes
//     on every vector-memory wait, so nothing in a step may depend on a load younger than that).  The chunk load
//     itself is UNCONDITIONAL (a chunk that crosses its stream's end is patched up in a rarely entered block, by
//     selects): inside an exec-masked region next to such a block, the join made the compiler wait for the load it
//     had just issued, once per group -- 24.6 against 20.8 ms.
//   * the sixteen-step loop exists twice: while every step updates the model, and -- for models that freeze inside a
//     block (frequency bits <= 16) -- with count, reciprocal and addends selected per step, so that a block past its
//     freeze point stays in lock-step form.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_decode.hpp"

namespace redux {

constexpr uint32_t kAdBBase      = 32 * 64 * 16;            // octet cells: 32 rows x 64 lanes x 16 B
constexpr uint32_t kAdRingBase   = kAdBBase + 8 * 64 * 8;   // B cells: 8 rows x 64 lanes x 8 B
constexpr uint32_t kAdRingDwords = 16;                      // per lane
constexpr uint32_t kAdLdsBytes   = kAdRingBase + kAdRingDwords * 256;
static_assert(kAdLdsBytes == 40960, "four groups fill the CU's 160 KiB");
// a u16 node holds lowbit (<= 16 in LDS) + increments: the fast loop runs while increments <= 65519
constexpr uint32_t kAdFullValueSteps = 65519;

// (the phases of a step are fenced for the scheduler: without the fences the kernel takes 25.9 instead of 24.6 ms)
#define REDUX_AD_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef uint32_t ad_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t ad_u32x4 __attribute__((ext_vector_type(4)));

struct AdTree {
    char    *base; // LDS
    uint32_t L8, L16; // this lane's column in the B rows / the octet rows

    __device__ __forceinline__ void init(uint32_t *lds, uint32_t lane)
    {
        base = reinterpret_cast<char *>(lds);
        L8   = lane * 8u;
        L16  = lane * 16u;
    }
    // octet cell m: fields (u16) = nodes 8m+1 .. 8m+7, spare;  B cell g: fields = nodes 32g+16, 32g+8, spare, 32g+24
    __device__ __forceinline__ uint32_t octet_addr(uint32_t m) const { return (m << 10) | L16; }
    __device__ __forceinline__ uint32_t bcell_addr(uint32_t g) const { return kAdBBase + ((g << 9) | L8); }
    __device__ __forceinline__ ad_u32x4 ld_octet(uint32_t addr) const { return *reinterpret_cast<const ad_u32x4 *>(base + addr); }
    __device__ __forceinline__ ad_u32x2 ld_bcell(uint32_t addr) const { return *reinterpret_cast<const ad_u32x2 *>(base + addr); }
    __device__ __forceinline__ void bump64(uint32_t addr, uint32_t lo, uint32_t hi) const
    {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(base + addr), ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // every node = its lowbit (all frequencies 1, adaptive_tree.rs:43-45) when FULL, else 0 increments
    __device__ __forceinline__ void fill(bool full) const
    {
        const ad_u32x4 o = full ? ad_u32x4{0x00020001u, 0x00040001u, 0x00020001u, 0x00000001u} : ad_u32x4{0, 0, 0, 0};
        const ad_u32x2 b = full ? ad_u32x2{0x00080010u, 0x00080000u} : ad_u32x2{0, 0};
        for (uint32_t m = 0; m < 32; m++)
            *reinterpret_cast<ad_u32x4 *>(base + octet_addr(m)) = o;
        for (uint32_t g = 0; g < 8; g++)
            *reinterpret_cast<ad_u32x2 *>(base + bcell_addr(g)) = b;
    }
    // tree values -> increments (this lane's column; LDS operations of a wave complete in order)
    __device__ __forceinline__ void to_increments() const
    {
        for (uint32_t m = 0; m < 32; m++) {
            ad_u32x4 *c = reinterpret_cast<ad_u32x4 *>(base + octet_addr(m));
            *c          = *c - ad_u32x4{0x00020001u, 0x00040001u, 0x00020001u, 0x00000001u};
        }
        for (uint32_t g = 0; g < 8; g++) {
            ad_u32x2 *c = reinterpret_cast<ad_u32x2 *>(base + bcell_addr(g));
            *c          = *c - ad_u32x2{0x00080010u, 0x00080000u};
        }
    }
};

// get_symbol (adaptive_tree.rs:115-136) + the high end of get_frequency (:105-113) over INCREMENT nodes: the
// predicated loop's search.  Safe for any v (finished lanes run it on garbage).
__device__ __forceinline__ DecFound ad_search_slow(const AdTree &A, const DecTop &T, uint32_t v, uint32_t c)
{
    uint32_t q = ~v, hq = q + (c - 1u), bits = 0, q2;
    DecFound f;
    f.eofq = hq;
    bool left;
#define REDUX_AD_LEVEL(t)                                                                                              \
    left = __builtin_uadd_overflow(q, (t), &q2);                                                                       \
    bits = __builtin_amdgcn_alignbit(bits, q2, 31);                                                                    \
    q    = q > q2 ? q : q2;                                                                                            \
    hq   = hq < q2 ? hq : q2;
    REDUX_AD_LEVEL(T.n128)
    const uint32_t x6 = left ? T.n64 : T.n192, c5l = left ? T.n32 : T.n160, c5r = left ? T.n96 : T.n224;
    REDUX_AD_LEVEL(x6)
    const uint32_t x5 = left ? c5l : c5r;
    REDUX_AD_LEVEL(x5)
    const ad_u32x2 b = A.ld_bcell(A.bcell_addr(bits & 7u));
    REDUX_AD_LEVEL((b.x & 0xFFFFu) + 16u)
    const uint32_t x3 = left ? b.x : b.y;
    REDUX_AD_LEVEL((x3 >> 16) + 8u)
    const ad_u32x4 o = A.ld_octet(A.octet_addr(bits & 31u));
    REDUX_AD_LEVEL((o.y >> 16) + 4u)            // node +4
    const uint32_t u = left ? o.x : o.z;        // (+1, +2) or (+5, +6)
    const uint32_t w = left ? o.y : o.w;        // (+3, +4) or (+7, spare)
    REDUX_AD_LEVEL((u >> 16) + 2u)              // node +2 or +6
    const uint32_t x0 = left ? u : w;
    REDUX_AD_LEVEL((x0 & 0xFFFFu) + 1u)         // node +1 / +5 or +3 / +7
#undef REDUX_AD_LEVEL
    f.s  = bits & 0xFFu;
    f.lo = v + q + 1u;
    f.hi = v + hq + 1u;
    return f;
}

// update(s+1) (adaptive_tree.rs:83-92): +1 on the levels where bit b of s is clear
__device__ __forceinline__ void ad_update(const AdTree &A, DecTop &T, uint32_t s)
{
    dec_update_regs(T, s);
    const uint32_t n = ~s;
    const uint32_t n0 = n & 1u, n1 = (n >> 1) & 1u, n2 = (n >> 2) & 1u, n3 = (n >> 3) & 1u, n4 = (n >> 4) & 1u;
    // B cell of group s >> 5: +16 iff bit 4 clear; +8 iff bits 4, 3 clear; +24 iff bit 4 set, bit 3 clear
    A.bump64(A.bcell_addr((s >> 5) & 7u), n4 | ((n4 & n3) << 16), ((n4 ^ 1u) & n3) << 16);
    // half of the octet cell of s >> 3 that holds the nodes of s's quad: fields (+1 | +5, +2 | +6, +3 | +7, +4 | spare)
    A.bump64(A.octet_addr((s >> 3) & 31u) + ((s & 4u) << 1), (n1 & n0) | (n1 << 16), ((n1 ^ 1u) & n0) | (n2 << 16));
}

template <bool CB32>
__device__ __forceinline__ void decode_adaptive_body(const DecArgs &a, uint32_t *lds)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = slot < a.nblocks && !(a.table && a.table[slot].index == 0xFFFFFFFFu /* idle entry */);

    AdTree A;
    A.init(lds, lane);
    A.fill(true); // (one wave per workgroup and every lane owns its columns: no barrier)

    // which block this lane decodes, where its output goes and how much room it has there
    uint64_t blk = slot, dst_off = slot * (uint64_t)a.block_size;
    uint32_t capn = a.block_size;
    if (a.table && live) {
        const redux_block e = a.table[slot];
        blk     = e.index;
        dst_off = e.offset;
        capn    = e.length;
    }
    const uint32_t cb = CB32 ? 32u : a.code_bits, sh = CB32 ? 0u : 32 - cb;
    uint64_t       size = 0;
    const uint8_t *sp   = a.in;
    if (live) {
        const uint64_t o0 = a.in_offsets[blk];
        size              = a.in_offsets[blk + 1] - o0;
        sp                = a.in + o0;
    }
    const uint32_t stream_bits = (uint32_t)(size * 8);
    uint8_t       *dst         = a.out + (live ? dst_off : 0);
    const rc_ptr   rcp         = (rc_ptr)a.rc;
    const uint32_t nfreeze     = a.nfreeze;
    const bool     aligned4    = a.aligned4 != 0;
    const bool     aligned16   = a.aligned4 == 2;

    // ---- stream side: ring of 16 dwords per lane in LDS (see the header comment) -----------------------------
    //   * dword d of lane l: ring byte kAdRingBase + ((d & 15) << 8) + 4l;
    //   * once per group the chunk requested a group ago is written to the ring and the next one is requested:
    //     chunk wr (dwords 4wr .. 4wr+3) when 4wr - rpo <= 12 -- its slot's old content, chunk wr-4, is consumed
    //     then --, otherwise chunk wr-1 once more (it lands on its own copy): no lane predicate on the way;
    //   * a step consumes at most one dword, a group at most four: the ring never runs dry (initial fill: 12);
    //   * indices are clamped to the stream's last dword: bits past the end of a stream are never USED (consuming
    //     them is the Eof error, detected by the bit count), but the loads must stay inside the buffer.  A lane
    //     without a stream reads the offsets table instead (always mapped) and is finished from the start.
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    typedef const __attribute__((address_space(1))) ad_u32x4 *gptr4;
    const bool      has      = live && size > 0;
    const uintptr_t sp_abs   = (uintptr_t)sp;
    const gptr      gin      = has ? (gptr)(sp_abs & ~(uintptr_t)3) : (gptr)(uintptr_t)a.in_offsets;
    const uint32_t  rpo_last = has ? (uint32_t)(((((sp_abs + size + 3) & ~(uintptr_t)3) - (sp_abs & ~(uintptr_t)3)) >> 2) - 1) : 0u;
    const uint32_t  skip     = has ? (uint32_t)(sp_abs & 3) * 8 : 0u;
    const gptr      gsafe    = (gptr)(uintptr_t)a.in_offsets;
    const uint32_t  L4       = lane * 4u;
    auto rd = [&](uint32_t o) { return gin[o < rpo_last ? o : rpo_last]; };
    auto ring_write = [&](uint32_t chunk, const ad_u32x4 &x) {
        uint32_t *q = reinterpret_cast<uint32_t *>(A.base + kAdRingBase + ((chunk & 3u) << 10) + L4);
        q[0] = x.x; q[64] = x.y; q[128] = x.z; q[192] = x.w;
    };
    auto ring_read = [&](uint32_t d) { return *reinterpret_cast<const uint32_t *>(A.base + kAdRingBase + ((d & 15u) << 8) + L4); };
    uint32_t rpo = 2, wr = 0, pend_chunk = 0;
    ad_u32x4 ldq = {0, 0, 0, 0};
    DecLane  S;
    {
        uint32_t d0 = 0, d1 = 0;
        for (; wr < 3; wr++) {
            ldq = ad_u32x4{rd(4 * wr), rd(4 * wr + 1), rd(4 * wr + 2), rd(4 * wr + 3)};
            if (wr == 0) {
                d0 = has ? __builtin_bswap32(ldq.x) : 0u;
                d1 = (has && rpo_last >= 1) ? __builtin_bswap32(ldq.y) : 0u;
            }
            ring_write(wr, ldq);
            pend_chunk = wr; // (the first group "retires" chunk 2 once more)
        }
        S.bbits = (((uint64_t)d0 << 32) | d1) << skip;
        S.bcnt  = 64 - skip;
    }
    uint32_t fetched = ring_read(rpo);
    auto retire = [&]() { ring_write(pend_chunk, ldq); };
    auto request = [&]() {
        const bool     room = (int32_t)(4u * wr - rpo) <= 12;
        const uint32_t c    = room ? wr : wr - 1u;
        const bool     tail = 4u * c + 3u > rpo_last;
        pend_chunk          = c;
        // (a chunk that crosses the end of its stream is loaded dword by dword with clamped indices, in a rarely entered
        // block; the 16-byte load of such a lane reads the offsets table instead: always mapped, 16 bytes or more)
        ldq = *reinterpret_cast<gptr4>(tail ? gsafe : gin + 4u * c);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tail) != 0, 0)) {
            const ad_u32x4 t = ad_u32x4{rd(4u * c), rd(4u * c + 1u), rd(4u * c + 2u), rd(4u * c + 3u)};
            ldq.x = tail ? t.x : ldq.x;
            ldq.y = tail ? t.y : ldq.y;
            ldq.z = tail ? t.z : ldq.z;
            ldq.w = tail ? t.w : ldq.w;
        }
        wr += room ? 1u : 0u;
    };
#define REDUX_AD_READER                                                                                                \
    {                                                                                                                  \
        const bool     need = S.bcnt <= 32;                                                                            \
        const uint64_t add  = (uint64_t)(need ? __builtin_bswap32(fetched) : 0u) << ((32 - S.bcnt) & 63);              \
        S.bbits |= add;                                                                                                \
        S.bcnt += need ? 32u : 0u;                                                                                     \
        rpo += need ? 1u : 0u;                                                                                         \
        fetched = ring_read(rpo);                                                                                      \
    }

    // the same in the fast loop: "fewer than 33 bits left" as a sign mask instead of a compare (no lane mask in scalar
    // registers: two wait states between its VALU writer and its VALU reader on gfx950)
#define REDUX_AD_READER_FAST                                                                                           \
    {                                                                                                                  \
        uint32_t need = (uint32_t)((int32_t)(S.bcnt - 33u) >> 31); /* all ones: refill */                              \
        asm("" : "+v"(need));                                                                                          \
        const uint64_t add = (uint64_t)(__builtin_bswap32(fetched) & need) << ((32 - S.bcnt) & 63);                    \
        S.bbits |= add;                                                                                                \
        S.bcnt += need & 32u;                                                                                          \
        rpo -= need;                                                                                                   \
        fetched = ring_read(rpo);                                                                                      \
    }

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
    uint4    oqp    = make_uint4(0, 0, 0, 0); // the lock-step loop: the turn before oq's, while it waits for its partner
    uint32_t p      = 0;
    DecTop   T      = dec_top_new();

    // ---------------- lock-step iterations of sixteen symbols ----------------
    // While p < min(every live lane's capacity, freeze point, what a u16 tree value can count) every step updates the
    // model and has room for its symbol.
    uint32_t capw = capn;
    if (a.table) { // (uniform) capacities differ by lane: the shortest live one bounds the loop
        capw = live ? capn : 0xFFFFFFFFu;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t w = __shfl_xor(capw, o);
            capw             = w < capw ? w : capw;
        }
        capw = __builtin_amdgcn_readfirstlane(capw);
        capw = capw == 0xFFFFFFFFu ? 0u : capw;
    }
    uint32_t pfast = capw < nfreeze ? capw : nfreeze;
    pfast          = pfast < kAdFullValueSteps ? pfast : kAdFullValueSteps;
    uint32_t livemask = (int32_t)S.dflag < 0 ? 0x7FFFFFFFu : 0xFFFFFFFFu; // sign bit cleared once the block is finished
    uint32_t fin_cons = S.consumed;                                       // what a lane that finishes in this loop ends with
    if (aligned16) {
        double cdm1 = 256.0, cd = 257.0;
        // The group's four reciprocals are loaded a group ahead with VECTOR loads (every lane the same 32 bytes), behind
        // the ring's chunk request: their latency is covered by the one vmcnt wait of the next group.  A scalar load
        // would share lgkmcnt with the LDS, return out of order and so sit in front of the next LDS wait.
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f64x4 *grc4;
        const grc4 rcv = (grc4)(uintptr_t)a.rc; // 256-byte aligned workspace, p a multiple of 4
        f64x4      rcg = rcv[0], rcn;
        asm volatile("" : "+v"(rcg)); // arrived before the loop: no in-loop wait inherits this load
        // update() of the register levels by table: bit b3 = s >> 5 of byte k says whether node k is incremented
        uint32_t kp0 = 0x0130030Fu, kp1 = 0x00401004u; // bytes: n128, n64, n192, n32 | n96, n160, n224
        uint32_t k10001 = 0x10001u, k10000 = 0x10000u, k01010101 = 0x01010101u;
        asm volatile("" : "+v"(kp0), "+v"(kp1), "+v"(k10001), "+v"(k10000), "+v"(k01010101));
        uint32_t left = stream_bits - S.consumed; // stream bits not pulled yet: its sign is read_bits' Err(Eof) (bitio/mod.rs:107)
        uint32_t r1 = ~(S.ihigh + S.low); // high - low of the interval, left-aligned: carried instead of ~high (see the commit)
        // range = high - low + 1 and its reciprocal are formed at the END of a step, as soon as the new interval is known:
        // the division chain of the next step's code value (codec.rs:131) starts under the rest of the commit
        uint32_t R1  = r1 >> sh;
        double   R1d = (double)R1, xd = R1d + 1.0, rinv = __builtin_amdgcn_rcp(xd);
        // The loop exists twice.  GEN == false: every step updates the model (p + 16 <= freeze point): constants pinned in
        // registers, the count as an induction variable.  GEN == true: the steps from the freeze point on update nothing
        // (adaptive_tree.rs:84) -- the count, its reciprocal and every addend are selected per step by the wave-uniform
        // "this step still updates", so the turn that crosses the freeze point and all turns behind it stay in lock-step
        // form instead of falling to the general loop below (a model of 14 frequency bits freezes after a quarter of a
        // 64 KiB block).  The frozen reciprocal is read once.
        // (only a model that freezes inside a block of this size has the second loop -- and a table that reaches its
        // freeze point: geometry() sizes it min(block_size, freeze point) + 33 entries)
        const bool   freezes = nfreeze < kAdFullValueSteps && nfreeze < a.block_size;
        const f64x4  rcF4 = rcv[freezes ? nfreeze >> 2 : 0u];
        const double rcF = (nfreeze & 3) == 0 ? rcF4.x : (nfreeze & 3) == 1 ? rcF4.y : (nfreeze & 3) == 2 ? rcF4.z : rcF4.w;
        auto fast_loop = [&](auto gen_tag, const uint32_t pend) {
        constexpr bool GEN = decltype(gen_tag)::value;
        for (; p + 16 <= pend; p += 16) {
            if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
                break;
            uint32_t img[4] = {0, 0, 0, 0}; // this iteration's sixteen bytes
            // The part of a step's commit that nothing in the same step waits for -- the octet cell's update, the stream
            // window's shift, the output byte -- is issued at the top of the NEXT step, where the division chain of the
            // code value leaves issue slots free (and where it shares a basic block with it: the fix-up branch at the end
            // of a step is a scheduling boundary).
            uint32_t t_aC = 0, t_lo = 0, t_hi = 0, t_n = 0, t_sym = 0;
#pragma unroll
            for (int G = 0; G < 4; G++) {
                // once per group, in this order (vmcnt counts loads AND stores, in order, so the one wait of a group
                // must find nothing younger than a group in flight): chunk -> ring, output store, next request
                retire();
                if (G == 0 && p != 0) {
                    // Output: the sixteen bytes of the turn before this one wait in oq.  They leave TOGETHER with the turn
                    // before them, as two adjacent 16-byte stores every second turn: one 16-byte store per turn lands in
                    // half a 32-byte sector, and the L2's background cleaning writes such a sector back before its other
                    // half arrives -- twice the decoded bytes in fabric writes (WRITE_SIZE 8.43e6 KiB for 4 GiB).
                    if (p & 16u) // (wave-uniform)
                        oqp = oq;
                    else if ((int32_t)S.dflag >= 0) {
                        *reinterpret_cast<uint4 *>(dst + (p - 32)) = oqp;
                        *reinterpret_cast<uint4 *>(dst + (p - 16)) = oq;
                    }
                }
                request();
                if (GEN) { // (the table ends 32 entries behind the freeze point)
                    const uint32_t pg = p + 4 * G < nfreeze ? p + 4 * G : nfreeze;
                    rcn = rcv[(pg >> 2) + 1];
                } else
                    rcn = rcv[((p + 4 * G) >> 2) + 1]; // the table has 32 entries of slack (geometry())
#pragma unroll
                for (int K = 0; K < 4; K++) {
                    const uint32_t idx = p + 4 * G + K;
                    const bool     upd = !GEN || idx < nfreeze; // (wave-uniform) this step updates the model
                    const double   rc  = upd ? rcg[K] : rcF;
                    const uint32_t c   = 257u + (upd ? idx : nfreeze);
                    // the update's constants: the pinned registers while every step updates, scalar selects otherwise
                    const uint32_t u01010101 = GEN ? (upd ? 0x01010101u : 0u) : k01010101;
                    const uint32_t u10001 = GEN ? (upd ? 0x10001u : 0u) : k10001, u10000 = GEN ? (upd ? 0x10000u : 0u) : k10000;
                    const uint32_t u1 = GEN ? (upd ? 1u : 0u) : 1u;
#define REDUX_AD_TAIL                                                                                                  \
    if (G != 0 || K != 0) {                                                                                            \
        const int PG = K == 0 ? G - 1 : G, PK = K == 0 ? 3 : K - 1; /* (constants once the loops are unrolled) */      \
        A.bump64(t_aC, t_lo, t_hi);                                                                                    \
        img[PG] = PK == 0 ? t_sym : (img[PG] | (t_sym << (8 * PK)));                                                   \
    }
                    // ---- the previous step's deferred commit
                    if (G != 0 || K != 0) {
                        S.bbits <<= t_n;
                        S.bcnt -= t_n;
                    }
                    REDUX_AD_TAIL
                    // ---- the bit reader's refill for this step (bitio/mod.rs:78-120): independent of everything below until
                    // the commit, so it fills the issue slots the division chain leaves
                    REDUX_AD_READER_FAST
                    // ---- A: code value (codec.rs:129-131): dec_value() with the reciprocal already at hand
                    const uint32_t Vd = (S.W - S.low) >> sh;
                    const double   nd = __builtin_fma((double)Vd, cd, cdm1); // (Vd+1)*c - 1, exact (< 2^49)
                    // (quotient estimate q or q - 1, dec_value(); the correction as a sign: nd - (v0 + 1) * xd is negative iff
                    // v0 is the quotient already)
                    const uint32_t v0p = (uint32_t)__builtin_fma(nd, rinv, -0x1p-6) + 1u;
                    const double   rem = __builtin_fma(-(double)v0p, xd, nd);
                    uint32_t       fix = (uint32_t)((int32_t)(uint32_t)((uint64_t)__double_as_longlong(rem) >> 32) >> 31); // -1: no
                    asm("" : "+v"(fix));
                    const uint32_t v = v0p + fix;
                    // ---- B: get_symbol (adaptive_tree.rs:115-136), levels 7-5 from registers.  No carries, no lane masks in
                    // scalar registers: a VALU result that goes through an SGPR pair reaches the next VALU instruction two wait
                    // states late on gfx950, and the descent is the step's serial chain.  "Went right" is the sign of q2;
                    // spread over the dword (one shift) it selects the next candidates by v_bfi.
                    uint32_t cm1 = c - 1u; // (one scalar register: the compiler would otherwise split it into p + a literal, two adds)
                    asm volatile("" : "+s"(cm1));
                    uint32_t q = ~v, hq = q + cm1, bits = 0, q2, m;
                    const uint32_t eofq = hq; // top bit set: v >= count - 1 -> the EOF symbol (adaptive_tree.rs:116)
#define REDUX_AD_STEP(t)                                                                                               \
    q2   = q + (t);                                                                                                    \
    bits = __builtin_amdgcn_alignbit(bits, q2, 31);                                                                    \
    q    = q > q2 ? q : q2;                                                                                            \
    hq   = hq < q2 ? hq : q2;
#define REDUX_AD_MASK()                                                                                                \
    m = (uint32_t)((int32_t)q2 >> 31); /* all ones: the probe succeeded, go right */                                   \
    asm("" : "+v"(m))                  /* (opaque: the compiler would turn the picks back into compare + v_cndmask) */
#define REDUX_AD_PICK(l, r) ((m & (r)) | (~m & (l)))
                    REDUX_AD_STEP(T.n128)
                    REDUX_AD_MASK();
                    const uint32_t x6 = REDUX_AD_PICK(T.n64, T.n192), c5l = REDUX_AD_PICK(T.n32, T.n160), c5r = REDUX_AD_PICK(T.n96, T.n224);
                    REDUX_AD_STEP(x6)
                    REDUX_AD_MASK();
                    const uint32_t x5 = REDUX_AD_PICK(c5l, c5r);
                    REDUX_AD_STEP(x5)
                    const uint32_t aB = A.bcell_addr(bits); // (bits = s >> 5)
                    ad_u32x2 bc = A.ld_bcell(aB);
                    REDUX_AD_FENCE();
                    // ---- B's shadow: update(s+1), adaptive_tree.rs:83-92, for the levels kept in registers, which the
                    // top three bits of s decide.  (volatile, so that the block stays between the load and its first use:
                    // the compiler would otherwise sink it to the next step, where the values are used.)
                    {
                        uint32_t b3 = bits; // (an opaque copy: without it the kernel measures 0.5 % slower, one v_mov fewer or not)
                        asm volatile("" : "+v"(b3));
                        // byte k of t0 / t1 = 1 iff node k is incremented; each add takes its byte as an SDWA operand
                        const uint32_t t0 = (kp0 >> b3) & u01010101, t1 = (kp1 >> b3) & u01010101;
#define REDUX_AD_ADD_BYTE(n, t, B)                                                                                     \
    asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B : "+v"(n) : "v"(t));
                        REDUX_AD_ADD_BYTE(T.n128, t0, 0)
                        REDUX_AD_ADD_BYTE(T.n64, t0, 1)
                        REDUX_AD_ADD_BYTE(T.n192, t0, 2)
                        REDUX_AD_ADD_BYTE(T.n32, t0, 3)
                        REDUX_AD_ADD_BYTE(T.n96, t1, 0)
                        REDUX_AD_ADD_BYTE(T.n160, t1, 1)
                        REDUX_AD_ADD_BYTE(T.n224, t1, 2)
#undef REDUX_AD_ADD_BYTE
                    }
                    REDUX_AD_FENCE();
                    // ---- C: levels 4, 3 from the B cell (+16 | +8, spare | +24)
                    REDUX_AD_STEP(bc.x & 0xFFFFu)
                    REDUX_AD_MASK();
                    const uint32_t m4 = m;
                    const uint32_t x3 = REDUX_AD_PICK(bc.x, bc.y);
                    REDUX_AD_STEP(x3 >> 16)
                    REDUX_AD_MASK();
                    const uint32_t m3 = m;
                    const uint32_t aC = A.octet_addr(bits); // (bits = s >> 3)
                    ad_u32x4 oc = A.ld_octet(aC);
                    REDUX_AD_FENCE();
                    // ---- C's shadow: the B cell's update (+16 iff bit 4 clear, +8 iff bits 4 and 3 clear, +24 iff bit 4
                    // set and bit 3 clear), the factor both ends of the new interval share (codec.rs:133-134)
                    {
                        const uint32_t t34 = (~m3 & u10000) | u1;
                        A.bump64(aB, ~m4 & t34, m4 & ~m3 & u10000);
                    }
                    double Y = __builtin_fma(R1d, rc, rc);
                    if (GEN) { // the next step's count: one more while this step updated
                        const double inc = upd ? 1.0 : 0.0;
                        cdm1 += inc;
                        cd += inc;
                    } else {
                        cdm1 = cd;
                        cd += 1.0;
                    }
                    asm volatile("" : "+v"(Y), "+v"(cd));
                    REDUX_AD_FENCE();
                    // ---- D: levels 2-0 from the octet cell (+1 | +2, +3 | +4, +5 | +6, +7 | spare), narrowing and
                    // renormalisation (codec.rs:133-161)
                    REDUX_AD_STEP(oc.y >> 16) // node +4
                    REDUX_AD_MASK();
                    const uint32_t m2 = m;
                    const uint32_t u  = REDUX_AD_PICK(oc.x, oc.z);
                    const uint32_t w  = REDUX_AD_PICK(oc.y, oc.w);
                    REDUX_AD_STEP(u >> 16)    // node +2 or +6
                    REDUX_AD_MASK();
                    const uint32_t m1 = m;
                    const uint32_t x0 = REDUX_AD_PICK(u, w);
                    REDUX_AD_STEP(x0 & 0xFFFFu) // node +1 / +5 or +3 / +7
                    const uint32_t sym = bits & 0xFFu;
                    uint32_t lo  = v + q + 1u;  // v - rem = cum(s)
                    uint32_t hi  = v + hq + 1u; // cum(s + 1): the upper boundary of the last level that went left
                    uint32_t nlow   = S.low + (scale_div<false>(R1, Y, lo, c) << sh);
                    uint32_t nihigh = 0u - (S.low + (scale_div<false, true>(R1, Y, hi, c) << sh));
                    const uint32_t xx     = ~(nlow ^ nihigh);
                    uint32_t       k;
                    asm("v_ffbh_u32 %0, %1" : "=v"(k) : "v"(xx)); // (32-bit codes: low != high while count < 2^17; narrower ones:
                                                                  // the padding below the code differs, so k <= code_bits)
                    const uint32_t low2  = nlow << (k & 31u);
                    const uint32_t ih2   = nihigh << (k & 31u);
                    const uint32_t t2    = (low2 & ih2) << 1;
                    const uint32_t j     = (uint32_t)__builtin_clz(~t2);
                    const uint32_t n     = k + j; // bits pulled by get_bit (codec.rs:157)
                    const uint32_t left0 = left, left2 = left0 - n;
                    uint32_t e     = (eofq | left2) & livemask;
                    // ---- E: commit, every lane.  The half of the octet cell that holds the quad of s: fields
                    // (+1 | +5, +2 | +6, +3 | +7, +4 | spare) get (bits 1, 0 clear; bit 1 clear; bit 1 set, bit 0 clear;
                    // bit 2 clear); issued at the top of the next step
                    {
                        const uint32_t r0 = sym & 1u;
                        t_lo  = ~m1 & u10001 & ~r0;
                        t_hi  = (~m2 & u10000) | (m1 & ~r0 & u1);
                        t_aC  = aC | (m2 & 8u);
                        t_n   = n;
                        t_sym = sym;
                    }
                    // The j E3 steps drop bit 30 j times: shift by j, clear bit 31.  Bit 31 of both shifted values is the same
                    // (j >= 1: both had ones there; j == 0: both have 0 after the k shared bits), so the next interval width
                    // ~(low + ~high) is the same with or without the two clears: only low, which is added to, gets its clear,
                    // and ~high is not kept at all.
                    const uint32_t Ls = low2 << j;
                    r1                = ~(Ls + (ih2 << j));
                    S.low             = Ls & 0x7FFFFFFFu;
                    left              = left2;
                    // [value | next 32 bits] << k, keep the top bit, << j, put it back (codec.rs:143-157).  A narrow code
                    // whose interval collapsed (k == code_bits) takes that top bit from the new bits: both shifts 64-bit.
                    const uint32_t nxt  = (uint32_t)(S.bbits >> 32);
                    const uint64_t comb = CB32 ? (((uint64_t)S.W << 32) | nxt) : (((uint64_t)S.W << 32) | ((uint64_t)nxt << sh));
                    const uint32_t h2   = (uint32_t)((comb << n) >> 32);
                    const uint32_t h1   = CB32 ? S.W << k : (uint32_t)((comb << k) >> 32);
                    S.W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
                    R1   = r1 >> sh;
                    R1d  = (double)R1;
                    xd   = R1d + 1.0;
                    rinv = __builtin_amdgcn_rcp(xd);
                    const uint32_t img0 = img[G]; // (this step's byte joins it at the top of the next step)
                    // ---- the two ways a block ends here
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)e < 0) != 0, 0)) { // one scalar branch; selects inside
                        const bool fin        = (int32_t)e < 0;
                        const bool eof_symbol = (int32_t)eofq < 0; // decided first: decompress_symbol returns before renormalising
                        S.st     = fin && !eof_symbol ? REDUX_EOF : S.st;
                        fin_cons = fin ? stream_bits - (eof_symbol ? left0 : left2) : fin_cons;
                        S.n_out  = fin ? p + 4 * G + K : S.n_out;
                        S.dflag  = fin ? 0x80000000u : S.dflag;
                        livemask = fin ? 0x7FFFFFFFu : livemask;
                        // (the turn before this one, if it is still waiting for its partner: see the top of the turn)
                        if ((p & 16u) && fin)
                            *reinterpret_cast<uint4 *>(dst + (p - 16)) = oqp;
                        // the lane's unstored output in the form the end of the kernel writes out: G whole dwords in the
                        // LAST components of oq, the K bytes of the current one in obuf
                        stored = fin ? p : stored;
                        staged = fin ? p + 4 * G : staged;
                        S.obuf = fin ? (K == 0 ? 0u : img0) : S.obuf;
                        if (G == 1)
                            oq.w = fin ? img[0] : oq.w;
                        if (G == 2) {
                            oq.z = fin ? img[0] : oq.z;
                            oq.w = fin ? img[1] : oq.w;
                        }
                        if (G == 3) {
                            oq.y = fin ? img[0] : oq.y;
                            oq.z = fin ? img[1] : oq.z;
                            oq.w = fin ? img[2] : oq.w;
                        }
                    }
                }
                rcg = rcn;
            }
            A.bump64(t_aC, t_lo, t_hi); // the last step's deferred commit
            S.bbits <<= t_n;
            S.bcnt -= t_n;
            img[3] |= t_sym << 24;
            // the iteration's sixteen bytes: stored at the top of the next iteration (or below)
            oq.x = (int32_t)S.dflag >= 0 ? img[0] : oq.x;
            oq.y = (int32_t)S.dflag >= 0 ? img[1] : oq.y;
            oq.z = (int32_t)S.dflag >= 0 ? img[2] : oq.z;
            oq.w = (int32_t)S.dflag >= 0 ? img[3] : oq.w;
        }
        };
        fast_loop(std::false_type(), pfast);
        if (freezes) {
            p = __builtin_amdgcn_readfirstlane(p); // (uniform anyway; behind the first loop's exits the compiler no longer knows)
            fast_loop(std::true_type(), capw);
        }
#undef REDUX_AD_STEP
#undef REDUX_AD_MASK
#undef REDUX_AD_PICK
        if (p != 0 && (int32_t)S.dflag >= 0) {
            if ((p & 16u) == 0) // the last turn's partner is still waiting too
                *reinterpret_cast<uint4 *>(dst + (p - 32)) = oqp;
            *reinterpret_cast<uint4 *>(dst + (p - 16)) = oq;
            stored = p;
            staged = p;
        }
        S.ihigh = (~r1 - S.low) & 0x7FFFFFFFu; // (the predicated loop keeps ~high ...
        S.consumed = stream_bits - left;        //  ... and counts the bits pulled)
    }
    if ((int32_t)S.dflag < 0) // finished in the loop above (or never live): what the garbage steps since then did not touch
        S.consumed = fin_cons;
    A.to_increments();

    // ---------------- remaining steps (EOF symbol, frozen model, unaligned output, the last symbols of a full 64 KiB block) ----
    for (;; p++) {
        if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
            break;
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   rc  = rcp[nup];
        const uint32_t c   = 257u + nup;
        if ((p & 3) == 0) {
            retire();
            // Output: a finished group's dword is staged; 16-byte aligned blocks get one 16-byte store per four groups.
            // (A 4-byte store every four steps per lane is what the L2's background cleaning of resident dirty lines
            // turns into ten times the output in fabric writes.)
            if (aligned16) {
                if ((int32_t)S.dflag >= 0 && p > staged) { // a live lane has emitted p symbols
                    oq     = make_uint4(oq.y, oq.z, oq.w, S.obuf);
                    S.obuf = 0;
                    staged = p;
                    if ((p & 15u) == 0) {
                        *reinterpret_cast<uint4 *>(dst + (p - 16)) = oq;
                        stored = p;
                    }
                }
            } else if (aligned4 && (int32_t)S.dflag >= 0 && p > stored) {
                *reinterpret_cast<uint32_t *>(dst + (p - 4)) = S.obuf;
                S.obuf = 0;
                stored = p;
                staged = p;
            }
            request();
        }
        REDUX_AD_READER
        const uint32_t R1  = (~(S.ihigh + S.low)) >> sh;
        const uint32_t Vd  = (S.W - S.low) >> sh;
        const double   R1d = (double)R1;
        const uint32_t v   = dec_value(R1d, Vd, (double)c, (double)(c - 1u));
        const DecFound f   = ad_search_slow(A, T, v, c);
        dec_commit_careful<CB32>(S, f, R1, R1d, rc, c, sh, stream_bits, p, p < capn, aligned4, dst,
                                 [&](uint32_t s) { if (p < nfreeze) ad_update(A, T, s); });
    }
#undef REDUX_AD_READER
#undef REDUX_AD_READER_FAST
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

template <bool CB32>
__global__ void __launch_bounds__(64) k_decode_lock(DecArgs a)
{
    __shared__ uint32_t lds[kAdLdsBytes / 4]; // tree (36 KiB) + stream ring (4 KiB): four groups fill the CU's 160 KiB
    // One wave per SIMD, by construction: the LDS admits four of these workgroups on a CU but says nothing about which
    // SIMDs they land on, and two lock-step waves on one SIMD take ~1.6 x as long ("placement").
    // Claiming an accumulation register beyond the half-file mark makes the descriptor ask for more than 256 registers.
    asm volatile("" ::: "a255");
    decode_adaptive_body<CB32>(a, lds);
}

} // namespace redux
