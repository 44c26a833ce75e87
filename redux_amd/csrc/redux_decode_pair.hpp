// redux_decode_pair.hpp -- k_decode_pair: the lock-step decoder as TWO waves per 64 blocks (gfx950 only).
//
// Why.  A decoder lane is a serial chain (code value -> symbol search -> narrowing -> next code value,
// codec.rs:123-161), the tree caps a CU at four groups of 64 blocks, so k_decode_lock's wave is ALONE on its
// SIMD and pays ~4.5 cycles for every instruction it issues.  Running parts of the step twice
// (-DREDUX_DEC_DUP, profiles/r02_decode/dup_costs.txt) prices them, of 28.1 ms: code value 1.8, search 9.7,
// narrowing 1.7 -- the chain -- and, OFF the chain: LDS levels of the model update 5.2, bit reader 3.8,
// register levels of the update 3.8.  The off-chain work needs only the decoded symbol and the number of
// bits the step consumed, so a second wave can do it while the first one goes on:
//
//   CORE wave   code value, search (register levels + LDS reads), narrowing, renormalisation, code-value
//               update, the three register levels of the tree.  s_setprio 3.
//   AUX wave    the LDS levels of update(s+1) (adaptive_tree.rs:83-92), the whole bit reader
//               (bitio/mod.rs:78-120: global loads -> LDS ring -> the next 64 stream bits of every lane,
//               published one step ahead), and decompress_stream's write_bits(symbol, 8) (codec.rs:171).
//
// Hand-over through LDS mailboxes with the step number in band; nobody waits at a barrier:
//   MSG[lane]   core -> aux, one word per step, written right after the search:
//                   (step + 1) << 15 | bits consumed by the PREVIOUS step << 8 | symbol of this step
//   ACK         aux -> core: number of steps whose tree update is in LDS.  The core checks it between the
//               register levels and the first LDS probe of the next step's search (LDS executes one wave's
//               operations in order, so seeing the ACK means the atomics before it are done).
//   WIN[lane]   aux -> core: {step + 1, 64 stream bits starting at the position consumed BEFORE that step's
//               bits}.  The commit of step p wants the 32 bits at the position after step p-1; it takes them
//               from WIN(p-1) shifted by the bits step p-1 consumed (<= 32; more is an exceptional step).
//               So the aux wave learns a step's bit count one step late and still is a step ahead.
//   STOP/DONE   core -> aux, in the message slot once every message is acknowledged: "stop, (do / do not) write the
//               last symbol"; aux -> core "flushed".
//
// The pair covers the steps in which nothing exceptional happens in any lane (no lane finished, no EOF
// symbol, no interval collapse, no stream exhausted, no step consuming more than 32 bits).  At the first
// exceptional step -- for full blocks: the EOF symbol after the last byte -- the core stops the aux wave,
// takes the reader over at its bit position and finishes the blocks with k_decode_lock's per-lane careful
// loop.  Results are identical to k_decode / k_decode_lock by construction of each part; the parity suite
// and the decoder fuzz tests run through this kernel for every 4-byte aligned shape.
//
// Both spin loops are bounded: a wave that waits ~2^22 polls gives up, the block reports REDUX_IO_ERROR,
// and both waves leave.  A lost message is a bug, not a hang.
#pragma once

#include "redux_decode.hpp"
#include "redux_encode.hpp" // kClaimWords

namespace redux {

constexpr uint32_t kDpTreeBytes = 128 * 64 * 4;          // 32 KiB, layout of k_decode_lock
constexpr uint32_t kDpRing      = kDpTreeBytes;          // aux ring: 16 dwords per lane + 2 mirror rows (no wrap inside a 3-dword read)
constexpr uint32_t kDpRingRows  = 16;
constexpr uint32_t kDpWin       = kDpRing + (kDpRingRows + 2) * 256; // 2 x 64 x {hi, lo, seq+1, -}: record with field f in half f & 1
constexpr uint32_t kDpMsg       = kDpWin + 2 * 64 * 16;  // 64 words
constexpr uint32_t kDpCtrl      = kDpMsg + 256;          // [0] ACK, [2] DONE, [4..7] role booking scratch
constexpr uint32_t kDpLdsBytes  = kDpTreeBytes + 32 * 64 * 4; // 40 KiB: the tail reuses [kDpRing, end) as k_decode_lock's 32-dword ring
static_assert(kDpCtrl + 32 <= kDpLdsBytes, "mailboxes must fit beside the tree");
constexpr uint32_t kDpSpinLimit = 1u << 22;

#ifdef REDUX_DP_STATS // diagnostic build: how often and how long the core wave waits (tools/dp_stats.py)
__device__ unsigned long long g_dp_stats[8]; // [0] ack waits, [1] ack polls, [2] win waits, [3] win polls, [4] steps
#define DP_COUNT(i, n)                                                                                                 \
    if (blockIdx.x == 7 && lane == 0) {                                                                                \
        g_dp_stats[2 * (i)] += 1;                                                                                      \
        g_dp_stats[2 * (i) + 1] += (n);                                                                                \
    }
#else
#define DP_COUNT(i, n)
#endif
#ifndef REDUX_DP_AUX_SLEEP // s_sleep argument in the aux wave's poll loop (0: none)
#define REDUX_DP_AUX_SLEEP 1
#endif
#ifndef REDUX_DP_AUX_URGENT_PRIO // the aux wave's priority from a message's arrival to its acknowledgement
#define REDUX_DP_AUX_URGENT_PRIO 0
#endif

template <bool CB32>
__global__ void __launch_bounds__(128) k_decode_pair(DecArgs a)
{
    __shared__ uint32_t lds[kDpLdsBytes / 4];
    char *const    ldsb = reinterpret_cast<char *>(lds);
    // mailboxes are accessed through LDS-address-space volatile pointers: a volatile access through a generic
    // pointer becomes a FLAT instruction with an s_waitcnt behind each one
    typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
    typedef __attribute__((address_space(3))) char              lds_char;
    lds_char *const l3 = (lds_char *)ldsb;
    const uint32_t w8   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk  = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = blk < a.nblocks;

    for (uint32_t i = threadIdx.x; i < kDpLdsBytes / 16; i += 128)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // ---- roles: every SIMD must hold one core and one aux wave (same booking as k_encode_pair) ----
    lds_vu32 *ctrl = (lds_vu32 *)(l3 + kDpCtrl);
    uint32_t           role = w8, claim_delta = 0;
    {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        lds_vu32 *book = ctrl + 4;
        if (lane == 0)
            book[w8] = (hwid >> 4) & 3u; // my SIMD
        __syncthreads();
        uint32_t *claim_word = a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu));
        if (w8 == 0 && lane == 0) {
            const uint32_t s0 = book[0], s1 = book[1];
            const uint32_t straight = (1u << (4 * s0)) | (1u << (16 + 4 * s1));
            const uint32_t flipped  = (1u << (4 * s1)) | (1u << (16 + 4 * s0));
            uint32_t       old      = __hip_atomic_load(claim_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t       flip;
            do {
                const uint32_t cs = ((old >> (4 * s0)) & 15u) + ((old >> (16 + 4 * s1)) & 15u);
                const uint32_t cf = ((old >> (4 * s1)) & 15u) + ((old >> (16 + 4 * s0)) & 15u);
                flip              = cf < cs;
            } while (!__hip_atomic_compare_exchange_strong(claim_word, &old, old + (flip ? flipped : straight),
                                                           __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            book[2] = flip;
            book[3] = flip ? flipped : straight;
        }
        __syncthreads();
        role ^= __builtin_amdgcn_readfirstlane(book[2]);
        claim_delta = book[3];
    }

    // ---- what both waves know about the lane's stream -------------------------------------------
    const uint32_t L  = lane * 4u;
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
    const uint32_t nfreeze     = a.nfreeze;
    const bool     aligned16   = a.aligned4 == 2;
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4 *gptr4;
    const bool      has      = live && size > 0;
    const uintptr_t sp_abs   = (uintptr_t)sp;
    const gptr      gin      = has ? (gptr)(sp_abs & ~(uintptr_t)3) : (gptr)(uintptr_t)a.in_offsets;
    const uint32_t  rpo_last = has ? (uint32_t)(((((sp_abs + size + 3) & ~(uintptr_t)3) - (sp_abs & ~(uintptr_t)3)) >> 2) - 1) : 0u;
    const uint32_t  skip     = has ? (uint32_t)(sp_abs & 3) * 8 : 0u;
    auto rd = [&](uint32_t o) { return gin[o < rpo_last ? o : rpo_last]; };
    // steps the pair may cover: the model adapts and the block has room (as k_decode_lock's fast loop)
    const uint32_t pfast = (capn < nfreeze ? capn : nfreeze) & ~3u;

    lds_vu32 *msg = (lds_vu32 *)(l3 + kDpMsg) + lane;
    // WIN record of this lane with sequence field f: {hi, lo} then {f} at kDpWin + (f & 1) * 1024 + 16 * lane.  The
    // writer stores the data first and the field last, the reader loads the field first and the data last; LDS
    // executes a wave's operations in order, so a matching field vouches for the data.  Two halves, because the
    // aux wave may publish record f + 1 (on the message of step f) before the core has consumed record f.
    lds_vu32 *winbase = (lds_vu32 *)(l3 + kDpWin) + lane * 4;

    if (role == 1) {
        // =========================================================================================
        // AUX wave
        // =========================================================================================
        uint32_t A[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            A[b] = (b ? (1u << (b + 7)) : 0u) | L;
            asm volatile("" : "+v"(A[b]));
        }
        // ring: dword d of the lane at kDpRing + ((d & 15) << 8) + 4*lane, stored in STREAM order (byte-swapped
        // on the way in), rows 0 and 1 mirrored behind row 15
        auto ring_put = [&](uint32_t chunk, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
            const uint32_t r = (chunk & 3u) << 10;
            uint32_t      *q = reinterpret_cast<uint32_t *>(ldsb + kDpRing + r + L);
            x0 = __builtin_bswap32(x0); x1 = __builtin_bswap32(x1); x2 = __builtin_bswap32(x2); x3 = __builtin_bswap32(x3);
            q[0] = x0; q[64] = x1; q[128] = x2; q[192] = x3;
            if ((chunk & 3u) == 0) {
                q[16 * 64] = x0;
                q[17 * 64] = x1;
            }
        };
        uint32_t wr = 0;
        for (; wr < 3; wr++)
            ring_put(wr, rd(4 * wr), rd(4 * wr + 1), rd(4 * wr + 2), rd(4 * wr + 3));
        bool     pend = false;
        uint32_t pend_chunk = 0;
        u32x4    ldq = {0, 0, 0, 0};
        uint32_t C = cb; // stream bits consumed by the steps the aux wave knows of (codec.rs:124-127 primes code_bits)
        // the 64 stream bits at bit position `at` of the lane's stream -> WIN with sequence field `field`
        auto publish = [&](uint32_t field) {
            const uint32_t G  = skip + C;
            const uint32_t d  = G >> 5, shb = G & 31u;
            lds_vu32      *q  = (lds_vu32 *)(l3 + kDpRing + ((d & 15u) << 8) + L);
            const uint32_t d0 = q[0];
            const uint32_t d1 = q[64];
            const uint32_t d2 = q[128];
            const uint32_t hi = (uint32_t)(((((uint64_t)d0 << 32) | d1) << shb) >> 32);
            const uint32_t lo = (uint32_t)(((((uint64_t)d1 << 32) | d2) << shb) >> 32);
            lds_vu32 *r = winbase + (field & 1u) * 256;
            r[0] = hi;
            r[1] = lo;
            asm volatile("" ::: "memory");
            r[2] = field;
            return d;
        };
        uint32_t dcur = publish(0);
        uint32_t next = 0;          // messages processed
        uint32_t prev_s = 0;        // symbol of message next-1, not yet written out
        uint32_t obuf = 0, stored = 0, staged = 0;
        uint4    oq = make_uint4(0, 0, 0, 0);
        bool     flush_last = false, gave_up = false;
        // symbol number `idx` of the block (a live lane; idx is wave-uniform)
        auto emit = [&](uint32_t idx, uint32_t s) {
            obuf |= s << (8 * (idx & 3u));
            if ((idx & 3u) == 3u) {
                if (aligned16) {
                    oq     = make_uint4(oq.y, oq.z, oq.w, obuf);
                    staged = idx + 1;
                    if (((idx + 1) & 15u) == 0) {
                        *reinterpret_cast<uint4 *>(dst + (idx + 1 - 16)) = oq;
                        stored = idx + 1;
                    }
                } else {
                    *reinterpret_cast<uint32_t *>(dst + (idx - 3)) = obuf;
                    stored = staged = idx + 1;
                }
                obuf = 0;
            }
        };
        for (;;) {
            // ---- wait for message `next`, or for the stop word in the same slot (sent once every message has
            //      been acknowledged) ----
            uint32_t m = 0, spins = 0;
            for (;;) {
                m = *msg;
                if (__builtin_amdgcn_ballot_w64((m >> 15) != next + 1) == 0) // every lane's word is the new one
                    break;
                if (__builtin_amdgcn_ballot_w64((m >> 15) != 0x1FFFFu) == 0) {
                    flush_last = m & 1u;
                    gave_up    = (m & 2u) != 0;
                    m          = 0xFFFFFFFFu;
                    break;
                }
                if (++spins > kDpSpinLimit) {
                    gave_up = true;
                    m       = 0xFFFFFFFFu;
                    break;
                }
#if REDUX_DP_AUX_SLEEP
                __builtin_amdgcn_s_sleep(REDUX_DP_AUX_SLEEP);
#endif
            }
            if (m == 0xFFFFFFFFu)
                break;
            __builtin_amdgcn_s_setprio(REDUX_DP_AUX_URGENT_PRIO);
            const uint32_t s = m & 0xFFu, nprev = (m >> 8) & 0x7Fu;
            // (1) urgent: the LDS levels of update(s + 1), then the acknowledgement
            dec_update_lds(lds, A, s);
            asm volatile("" ::: "memory"); // the acknowledgement is issued after the atomics
            ctrl[0] = next + 1;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            // (2) the window the core wants at the commit of step next + 1
            C += nprev;
            dcur = publish(next + 1);
            // (3) write_bits(symbol, 8) of the previous step (this step's symbol may still turn out to belong
            //     to an exceptional step, which the core commits itself)
            if (next > 0)
                emit(next - 1, prev_s);
            prev_s = s;
            // (4) the ring's producer, once per four steps (as k_decode_lock's RETIRE / REQUEST)
            if ((next & 3u) == 3u) {
                if (pend)
                    ring_put(pend_chunk, ldq.x, ldq.y, ldq.z, ldq.w);
                // chunk wr (dwords 4wr..4wr+3) takes the rows of chunk wr-4: free once the reader's first dword
                // is past them; the reader touches dwords dcur..dcur+2 and advances <= 1 dword per step
                const bool room = (int32_t)(4u * wr - dcur) <= 12;
                const bool tail = 4u * wr + 3u > rpo_last;
                pend       = room;
                pend_chunk = wr;
                if (room && !tail)
                    ldq = *reinterpret_cast<gptr4>(gin + 4u * wr);
                if (__builtin_amdgcn_ballot_w64(room && tail) != 0) {
                    if (room && tail) {
                        ldq.x = rd(4u * wr);
                        ldq.y = rd(4u * wr + 1u);
                        ldq.z = rd(4u * wr + 2u);
                        ldq.w = rd(4u * wr + 3u);
                    }
                }
                wr += room ? 1u : 0u;
            }
            next++;
        }
        // ---- stopped: flush what has been decoded by the pair (bytes [0, nout) of every live lane) ----
        if (!gave_up) {
            if (flush_last && next > 0)
                emit(next - 1, prev_s);
            const uint32_t nout = flush_last ? next : (next ? next - 1 : 0);
            if (live) {
                const uint32_t k = (staged - stored) >> 2;
                const uint32_t comp[4] = {oq.x, oq.y, oq.z, oq.w};
                for (uint32_t j = 0; j < k; j++) {
                    const uint32_t idx = 4 - k + j;
                    const uint32_t w   = idx == 0 ? comp[0] : idx == 1 ? comp[1] : idx == 2 ? comp[2] : comp[3];
                    *reinterpret_cast<uint32_t *>(dst + stored + 4 * j) = w;
                }
                for (uint32_t i = staged; i < nout; i++)
                    dst[i] = (uint8_t)(obuf >> (8 * (i & 3)));
            }
        }
        __builtin_amdgcn_s_waitcnt(0); // the ring's last chunk request must not land after the core reuses the LDS
        ctrl[2] = 1;
        return;
    }

    // =============================================================================================
    // CORE wave
    // =============================================================================================
    __builtin_amdgcn_s_setprio(3);
    uint32_t A[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        A[b] = (b ? (1u << (b + 7)) : 0u) | L;
        asm volatile("" : "+v"(A[b]));
    }
    const rc_ptr rcp = (rc_ptr)a.rc;
    DecLane      S;
    {
        const uint32_t d0 = has ? __builtin_bswap32(rd(0)) : 0u;
        const uint32_t d1 = (has && rpo_last >= 1) ? __builtin_bswap32(rd(1)) : 0u;
        const uint64_t bb = (((uint64_t)d0 << 32) | d1) << skip;
        S.W               = (uint32_t)((bb >> 1) >> (63 - cb)) << sh; // codec.rs:124-127
    }
    S.bbits = 0; S.bcnt = 0;
    S.consumed = cb;
    S.low = 0; S.ihigh = 0;
    S.st = REDUX_OK;
    S.dflag = live ? 0u : 0x80000000u;
    if (live && S.consumed > stream_bits) { // stream shorter than code_bits: Err(Eof) at once
        S.st    = REDUX_EOF;
        S.dflag = 0x80000000u;
    }
    S.sbits = (int32_t)S.dflag < 0 ? 0u : stream_bits;
    S.n_out = 0;
    S.obuf  = 0;
    DecTop   T = dec_top_new();
    uint32_t p = 0;
    bool     lost = false; // the aux wave stopped answering (never expected)

    struct AckWait { // between the register levels and the first LDS probe: the previous step's LDS update must be in
        lds_vu32 *ctrl;
        uint32_t           want;
        uint32_t          *ackv; // read early in the step, so its LDS latency is not on the chain
        bool              *lost;
        uint32_t           lane;
        __device__ __forceinline__ void operator()(uint32_t &bits) const
        {
            if (__builtin_amdgcn_readfirstlane(*ackv) < want) {
                // The aux wave shares this SIMD at a lower priority: spinning at priority 3 would starve the
                // very wave being waited for.
                __builtin_amdgcn_s_setprio(0);
                uint32_t spins = 0;
                do {
                    __builtin_amdgcn_s_sleep(1);
                    *ackv = ctrl[0];
                    if (++spins > kDpSpinLimit) {
                        *lost = true;
                        break;
                    }
                } while (__builtin_amdgcn_readfirstlane(*ackv) < want);
                __builtin_amdgcn_s_setprio(3);
                DP_COUNT(0, spins);
            }
            asm volatile("" : "+v"(bits)::"memory"); // no tree read moves above the wait
        }
    };

    // ---- the pair's steps: everything committed here is unexceptional in all 64 lanes ----
    uint32_t sent = 0;          // messages written
    bool     step_open = false; // the pair stopped AT a step: searched (message out), not committed
    DecFound fo{};              // ... that step's search result and operands
    uint32_t R1o = 0, co = 0;
    double   R1do = 0.0, rco = 0.0;
    {
        double cdm1 = 256.0, cd = 257.0;
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f64x4 *grc4;
        const grc4 rcv = (grc4)(uintptr_t)a.rc;
        f64x4      rcg = rcv[0], rcn;
        asm volatile("" : "+v"(rcg));
        uint32_t nprev = 0;
        while (p + 4 <= pfast && !step_open) {
            rcn = rcv[(p >> 2) + 1]; // the table has 32 entries of slack (geometry())
#pragma unroll
            for (int K = 0; K < 4; K++) {
                uint32_t       ackv = ctrl[0];
                const double   rc   = rcg[K];
                const uint32_t c    = 257u + p + K;
                const uint32_t R1   = (~(S.ihigh + S.low)) >> sh;
                const uint32_t Vd   = (S.W - S.low) >> sh;
                const double   R1d  = (double)R1;
                const uint32_t v    = dec_value(R1d, Vd, cd, cdm1);
                const DecFound f    = dec_search(lds, L, T, v, c, AckWait{ctrl, p + K, &ackv, &lost, lane});
                // the symbol is known: the aux wave can start on the tree while this wave narrows
                asm volatile("" ::: "memory");
                *msg = ((p + K + 1u) << 15) | (nprev << 8) | f.s;
                sent = p + K + 1;
                // WIN record with field p+K: published a step ago; consumed ~25 instructions from here
                lds_vu32 *wr_ = winbase + ((p + K) & 1u) * 256;
                uint32_t           wf = wr_[2];
                asm volatile("" ::: "memory");
                uint32_t whi = wr_[0], wlo = wr_[1];
                const double   Y      = __builtin_fma(R1d, rc, rc);
                const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, f.lo, c) << sh);
                const uint32_t nihigh = 0u - (S.low + (scale_div<false>(R1, Y, f.hi, c) << sh));
                const uint32_t xx     = ~(nlow ^ nihigh);
                uint32_t       k;
                asm("v_ffbh_u32 %0, %1" : "=v"(k) : "v"(xx)); // -1 (sign bit) for low == high
                const uint32_t low2  = nlow << (k & 31u);
                const uint32_t ih2   = nihigh << (k & 31u);
                const uint32_t t2    = (low2 & ih2) << 1;
                const uint32_t j     = (uint32_t)__builtin_clz(~t2);
                const uint32_t n     = k + j;
                const uint32_t cons2 = S.consumed + n;
                // exceptional: lane finished / EOF symbol / interval collapse / stream exhausted (as k_decode_lock),
                // or more than 32 bits consumed (the next step's window extraction shifts by n)
                const uint32_t e = (CB32 ? (f.eofq | k | (S.sbits - cons2)) : (f.eofq | (cb - 1u - k) | (S.sbits - cons2))) | (32u - n);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)e < 0) != 0 || lost, 0)) {
                    // Step p+K is NOT committed here.  Its message is out: the aux wave applies this step's
                    // LDS-level update in every lane -- harmless where the lane is finished or at its EOF symbol
                    // (nothing reads those trees again), wanted everywhere else -- and does not write its symbol.
                    step_open = true;
                    fo = f; R1o = R1; co = c; R1do = R1d; rco = rc;
                    p += K;
                    break;
                }
                dec_update_regs(T, f.s);
                S.low      = (low2 << j) & 0x7FFFFFFFu;
                S.ihigh    = (ih2 << j) & 0x7FFFFFFFu;
                S.consumed = cons2;
                {
                    if (__builtin_amdgcn_ballot_w64(wf != p + K) != 0) {
                        __builtin_amdgcn_s_setprio(0);
                        uint32_t spins = 0;
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            wf = wr_[2];
                            asm volatile("" ::: "memory");
                            whi = wr_[0];
                            wlo = wr_[1];
                            if (++spins > kDpSpinLimit) {
                                lost = true;
                                break;
                            }
                        } while (__builtin_amdgcn_ballot_w64(wf != p + K) != 0);
                        __builtin_amdgcn_s_setprio(3);
                        DP_COUNT(1, spins);
                    }
                    const uint64_t win  = ((uint64_t)whi << 32) | wlo;
                    const uint32_t nxt  = (uint32_t)((win << nprev) >> 32);
                    const uint64_t comb = CB32 ? (((uint64_t)S.W << 32) | nxt) : (((uint64_t)S.W << 32) | ((uint64_t)nxt << sh));
                    const uint32_t h2   = (uint32_t)((comb << n) >> 32);
                    const uint32_t h1   = S.W << k;
                    S.W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
                }
                nprev = n;
                cdm1  = cd;
                cd += 1.0;
            }
            if (!step_open) {
                p += 4;
                rcg = rcn;
            }
        }
    }
#ifdef REDUX_DP_STATS
    if (blockIdx.x == 7 && lane == 0)
        g_dp_stats[4] += p;
#endif
    // ---- stop the aux wave: once it has acknowledged every message, the stop word goes into the message slot;
    //      its last symbol is written out only if that step was committed here ----
    __builtin_amdgcn_s_setprio(0);
    {
        uint32_t spins = 0;
        while (!lost && __builtin_amdgcn_readfirstlane(ctrl[0]) < sent) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kDpSpinLimit)
                lost = true;
        }
        asm volatile("" ::: "memory");
        *msg = 0xFFFF8000u | (lost ? 2u : 0u) | (step_open ? 0u : 1u);
        spins = 0;
        while (__builtin_amdgcn_readfirstlane(ctrl[2]) == 0 && ++spins < kDpSpinLimit)
            __builtin_amdgcn_s_sleep(1);
        if (spins >= kDpSpinLimit)
            lost = true;
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(3);
    if (lost) { // the hand-over failed: report it instead of decoding on a tree nobody vouches for (never expected)
        if (live) {
            a.out_sizes[blk] = 0;
            a.status[blk]    = REDUX_IO_ERROR;
            if (a.in_used)
                a.in_used[blk] = 0;
        }
    } else {
        // =========================================================================================
        // tail: k_decode_lock's per-lane careful loop, bytes stored one by one, reader at bit S.consumed.
        // The LDS behind the tree is this wave's alone now: it becomes the 32-dword ring of k_decode_lock.
        // =========================================================================================
        constexpr uint32_t RB = kDpRing;
        auto ring_write = [&](uint32_t chunk, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
            uint32_t *q = reinterpret_cast<uint32_t *>(ldsb + RB + ((chunk & 7u) << 10) + L);
            q[0] = x0; q[64] = x1; q[128] = x2; q[192] = x3;
        };
        auto ring_read = [&](uint32_t d) {
            return *reinterpret_cast<const uint32_t *>(ldsb + RB + ((d & 31u) << 8) + L);
        };
        // seek: the upcoming bits start at stream bit S.consumed = bit (skip + consumed) of the aligned dwords
        const uint32_t G  = skip + S.consumed;
        const uint32_t i0 = G >> 5, off = G & 31u;
        {
            const uint32_t d0 = has ? __builtin_bswap32(rd(i0)) : 0u;
            const uint32_t d1 = has ? __builtin_bswap32(rd(i0 + 1)) : 0u;
            S.bbits           = (((uint64_t)d0 << 32) | d1) << off;
            S.bcnt            = 64 - off;
        }
        uint32_t rpo = i0 + 2, wr = rpo >> 2;
        for (const uint32_t w0 = wr; wr < w0 + 6; wr++)
            ring_write(wr, rd(4 * wr), rd(4 * wr + 1), rd(4 * wr + 2), rd(4 * wr + 3));
        uint32_t fetched = ring_read(rpo);
        bool     pend = false;
        uint32_t pend_chunk = 0;
        u32x4    ldq = {0, 0, 0, 0};
        auto reader = [&]() { // REDUX_DEC_READER of k_decode_lock
            const bool     need = S.bcnt <= 32;
            const uint64_t add  = (uint64_t)(need ? __builtin_bswap32(fetched) : 0u) << ((32 - S.bcnt) & 63);
            S.bbits |= add;
            S.bcnt += need ? 32u : 0u;
            rpo += need ? 1u : 0u;
            fetched = ring_read(rpo);
        };
        if (step_open) {
            // the step the pair stopped at: searched on the tree BEFORE its update (fo), the LDS levels of its
            // update applied by the aux wave since; commit it per lane, register levels here
            reader();
            dec_commit_careful<CB32, false>(S, T, fo, lds, A, R1o, R1do, rco, co, sh, stream_bits, p, true, true, false, dst);
            p++;
        }
        for (;; p++) {
            if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
                break;
            const uint32_t nup = p < nfreeze ? p : nfreeze;
            const double   rc  = rcp[nup];
            const uint32_t c   = 257u + nup;
            if ((p & 3) == 0) { // producer: retire the chunk requested a group ago, request the next one
                if (pend)
                    ring_write(pend_chunk, ldq.x, ldq.y, ldq.z, ldq.w);
                const bool room = (int32_t)(4u * wr - rpo) <= 28;
                const bool tl   = 4u * wr + 3u > rpo_last;
                pend       = room;
                pend_chunk = wr;
                if (room && !tl)
                    ldq = *reinterpret_cast<gptr4>(gin + 4u * wr);
                if (__builtin_amdgcn_ballot_w64(room && tl) != 0) {
                    if (room && tl) {
                        ldq.x = rd(4u * wr);
                        ldq.y = rd(4u * wr + 1u);
                        ldq.z = rd(4u * wr + 2u);
                        ldq.w = rd(4u * wr + 3u);
                    }
                }
                wr += room ? 1u : 0u;
            }
            reader();
            const uint32_t R1  = (~(S.ihigh + S.low)) >> sh;
            const uint32_t Vd  = (S.W - S.low) >> sh;
            const double   R1d = (double)R1;
            const uint32_t v   = dec_value(R1d, Vd, (double)c, (double)(c - 1u));
            const DecFound f   = dec_search(lds, L, T, v, c);
            dec_commit_careful<CB32, true>(S, T, f, lds, A, R1, R1d, rc, c, sh, stream_bits, p, p < nfreeze, p < capn, false, dst);
        }
        if (live) {
            a.out_sizes[blk] = S.n_out;
            a.status[blk]    = S.st;
            if (a.in_used) { // the reader fetches whole bytes, and never past the end of the stream
                const uint64_t used = ((uint64_t)S.consumed + 7) / 8;
                a.in_used[blk]      = used < size ? used : size;
            }
        }
    }
    if (lane == 0) { // waves do not migrate: the same CU as at the start
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_sub(a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu)), claim_delta, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace redux
