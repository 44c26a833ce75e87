// redux_hip.hip -- gfx950 kernels + the C ABI of include/redux_hip.h.
//
// Kernels (all hand-written for CDNA4, wave64):
//   k_fill_rc        per-step reciprocal table 1/(257+i), biased up 4 ulp
//   k_encode<..>     one lane = one block: tree in LDS, interval state in registers
//   k_decode<..>     the inverse
//   k_scan_sizes     sizes -> offsets (exclusive scan) + status summary
//   k_compact        gather padded slots into the dense output
//   k_gen_iid/zipf   synthetic workloads generated straight into HBM
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC redux_hip.hip -o libredux_hip.so
#include "redux_coder.hpp"
#include "redux_any.hpp"

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace redux {

// ======================================================================================
// reciprocal table
// ======================================================================================
__global__ void k_fill_rc(double *rc, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double r = 1.0 / (double)(257u + i); // correctly rounded IEEE division
        rc[i] = __longlong_as_double(__double_as_longlong(r) + 4);
    }
}

// ======================================================================================
// encode
// ======================================================================================
struct EncArgs {
    const uint8_t *in;
    uint64_t       in_len;
    uint64_t       nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    const double  *rc;
    uint32_t       block_size;
    uint32_t       slot_cap;  // usable bytes of a slot
    uint32_t       nfreeze;   // freq_max - 257: number of updates before the freeze
    uint32_t       code_bits;
    uint32_t       aligned16; // in and block_size are 16-byte multiples
    uint32_t       lanes;     // live lanes per wave: 64, or 1 when 64 slots overflow 32-bit offsets
    uint32_t      *claims;    // kClaimWords words, zero at launch: k_encode_pair's per-CU role book
};

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// 16 consecutive symbols of every lane's block, all lanes alive, no EOF: the hot loop body.
// UPD: the model is still adaptive for the whole chunk (updates so far = q = p + i);
// otherwise it is frozen (adaptive_tree.rs:84) and nup = nfreeze for every symbol.
// A pending run longer than 32 bits can add any number of bytes, so the chunk's byte budget
// is guarded by the caller only for the common path (4 bytes per symbol) plus slack; the
// careful path inside encode_symbol_fast is entered at most once per such run and the caller
// re-checks the budget every chunk.
template <bool U16, bool FIXUP, bool UPD>
__device__ __forceinline__ void encode_chunk(const Tree<U16> &T, EncState &S, const uint4 cur, uint32_t p,
                                             uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst)
{
    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
    double         r[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        r[i] = rc[UPD ? p + i : nfreeze];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s   = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const uint32_t nup = UPD ? p + i : nfreeze; // wave-uniform
        uint32_t       lo, hi;
        T.template get_frequency<UPD>(s, nup, true, lo, hi);
        encode_symbol_fast<FIXUP>(S, lo, hi, 257u + nup, r[i], sh, wdst);
    }
}

template <bool U16, bool FIXUP>
__global__ void __launch_bounds__(64) k_encode(EncArgs a)
{
    __shared__ uint32_t lds[Tree<U16>::kDwords];
    const uint32_t lane = threadIdx.x;
    const uint64_t blk0 = (uint64_t)blockIdx.x * a.lanes; // wave-uniform
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < a.lanes && blk < a.nblocks;

    for (uint32_t i = lane; i < Tree<U16>::kDwords / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    Tree<U16> T;
    T.init(lds, lane);

    uint32_t len = 0;
    if (live) {
        const uint64_t rem = a.in_len - blk * a.block_size;
        len                = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    // wave-uniform bases (SGPR pairs) + 32-bit per-lane offsets
    const uint8_t *wsrc  = a.in + blk0 * a.block_size;
    const uint32_t soff  = live ? lane * a.block_size : 0u;
    uint8_t       *wdst  = a.slots + blk0 * a.slot_bytes;
    // Dead lanes of the last wave run the same instruction stream on block blk0's bytes and
    // store into the spare slot behind the last real one, so the hot loop needs no predicate.
    const uint32_t off0  = live ? lane * (uint32_t)a.slot_bytes : (uint32_t)(a.nblocks - blk0) * (uint32_t)a.slot_bytes;
    const uint32_t limit = off0 + a.slot_cap;

    // The lock-step loop covers [0, maxlen]; the unrolled path covers whole 16-byte chunks
    // strictly below the shortest live block's last symbol.
    const uint32_t minlen  = __builtin_amdgcn_readfirstlane(wave_min(live ? len : 0xFFFFFFFFu));
    // bytes the unrolled path may add per chunk without any per-store check: 16 x 4 + slack
    constexpr uint32_t kChunkBudget = 16 * 4 + 32;
    const uint32_t maxlen  = __builtin_amdgcn_readfirstlane(wave_max(live ? len : 0u));
    const uint32_t sh      = 32 - a.code_bits;
    const uint32_t nfreeze = a.nfreeze;
    const rc_ptr   rc      = (rc_ptr)a.rc;

    EncState S;
    enc_init(S, off0);

    uint32_t p        = 0;
    uint32_t main_end = 0;
    if (a.aligned16 && minlen != 0xFFFFFFFFu && minlen > 16)
        main_end = (minlen - 1) & ~15u;

    if (main_end) {
        // (A) adaptive chunks
        const uint32_t a_end = main_end < (nfreeze & ~15u) ? main_end : (nfreeze & ~15u);
        if (p < a_end) {
            uint4 cur = *reinterpret_cast<const uint4 *>(wsrc + soff);
            for (; p < a_end; p += 16) {
                if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit)) {
                    main_end = p; // a slot is nearly full: finish in the checked tail loop
                    break;
                }
                uint4 nxt = cur;
                if (p + 16 < a_end)
                    nxt = *reinterpret_cast<const uint4 *>(wsrc + soff + p + 16);
                encode_chunk<U16, FIXUP, true>(T, S, cur, p, nfreeze, rc, sh, wdst);
                cur = nxt;
            }
        }
        // (M) the one chunk that crosses the freeze point, symbol by symbol
        if (p < main_end && p < nfreeze) {
            const uint32_t m_end = p + 16;
            for (; p < m_end; p++) {
                const uint32_t nup = p < nfreeze ? p : nfreeze;
                uint32_t       lo, hi;
                T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze, lo, hi);
                encode_symbol<FIXUP>(S, lo, hi, 257u + nup, rc[nup], sh, false, wdst, limit);
            }
        }
        // (F) frozen chunks: static model, no LDS writes
        if (p < main_end) {
            uint4 cur = *reinterpret_cast<const uint4 *>(wsrc + soff + p);
            for (; p < main_end; p += 16) {
                if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit)) {
                    main_end = p;
                    break;
                }
                uint4 nxt = cur;
                if (p + 16 < main_end)
                    nxt = *reinterpret_cast<const uint4 *>(wsrc + soff + p + 16);
                encode_chunk<U16, FIXUP, false>(T, S, cur, p, nfreeze, rc, sh, wdst);
                cur = nxt;
            }
        }
    }

    // Tail: symbol by symbol with per-lane predicates (ragged lengths, the EOF symbol).
    for (; p <= maxlen; p++) {
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   r   = rc[nup];
        const uint32_t c   = 257u + nup;
        if (live && p < len) {
            uint32_t lo, hi;
            // The update of a block's last symbol is never observed (the EOF range is
            // derived), and skipping it keeps every u16 node below 65536.
            T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze && p + 1 != len, lo, hi);
            encode_symbol<FIXUP>(S, lo, hi, c, r, sh, false, wdst, limit);
        } else if (live && p == len) {
            // EOF symbol (codec.rs:108): cum(256) = count-1, cum(257) = count
            const uint32_t shifts = encode_symbol<FIXUP>(S, c - 1, c, c, r, sh, true, wdst, limit);
            const uint32_t size   = encode_finish(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[blk]  = size;
            a.status[blk] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
}

// ======================================================================================
// encode, two waves per 64 blocks (the production path for u16 trees)
//
// The tree (32 KiB per 64 blocks) caps a CU at four groups of 64 blocks -- one wave per SIMD
// if a group is one wave, and a lone wave issues one instruction per ~4 cycles.  The model
// (tree query + update) does not depend on the coder's interval state, so a group is split
// into a MODEL wave and a CODER wave that share the group's LDS:
//   wave 0: input bytes -> get_frequency -> (low, high) pairs into an LDS ring
//   wave 1: ring -> interval narrowing, renormalisation, bit output
// Eight waves per CU = two per SIMD, so each SIMD always has a second instruction stream to
// issue from.  (Measured, profiles/r01_ubench: a gfx950 SIMD retires the VOP3-type ops this
// code is made of at ~4.5 cycles per wave-instruction however many waves feed it, so the two
// streams together run at the SIMD's VALU rate.  Two three-wave variants were built, passed
// the whole parity suite and were removed because they were slower: the model split by tree
// level (+35 %: duplicated per-symbol work) and a three-stage pipeline nodes -> sums -> coder
// with no duplicated work (+83 %: the LDS only has room for 2-symbol ring halves, and three
// synchronised waves get LESS aggregate VALU throughput than two, 5.7 vs 4.9 cycles per
// instruction in tools/ubench "40 VALU + barrier").)  The ring holds 2 x 8 symbols x 64 lanes x 8 B = 8 KiB (40 KiB per workgroup,
// four workgroups = the CU's 160 KiB exactly); one s_barrier per 8 symbols hands a half over.
// Only LDS traffic must be complete at the hand-off, so the barrier waits on lgkmcnt alone:
// the coder's stores and the model's prefetch loads stay in flight across it.
// ======================================================================================
constexpr uint32_t kRingSlots = 8;                               // symbols per hand-off
constexpr uint32_t kRingBytes = 2 * kRingSlots * 64 * 8;         // two halves of uint2[8][64]

#ifdef REDUX_STAMPS
// Diagnostic build only (never timed, never shipped): every ring barrier is bracketed by
// s_memtime; lane 0 of each wave accumulates {last stamp, cycles between barriers, cycles
// inside barriers, count} in the tree's unused row 0 (LDS bytes 0..63) and the kernel copies
// them to the spare slot at exit.  The barrier drains lgkmcnt anyway, so the stamps do not
// change what the waves overlap.
typedef __attribute__((address_space(3))) unsigned long long *lds64p;
__device__ __forceinline__ void pair_barrier()
{
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        lds64p a = (lds64p)(uintptr_t)((threadIdx.x >> 6) * 32);
        const unsigned long long prev = a[0];
        if (prev)
            a[1] += t0 - prev;
        a[2] += t1 - t0;
        a[0] = t1;
        a[3] += 1;
    }
}
#else
__device__ __forceinline__ void pair_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif

// Input path of the model wave: every lane reads its own block one whole 128-byte line at a
// time (eight 16-byte loads issued back to back), so each line crosses the L2 <-> fabric
// boundary once: FETCH_SIZE 2.14e6 KiB per 4 GiB pass, identical to a clean streaming read,
// against 7.4e6-9.5e6 KiB with 16 bytes per visit (the line was evicted between visits), and
// WRITE_SIZE drops 38 % as well (less L2 pollution).  c[] is the current line, n[] the next
// one, already in flight; the chunk index is wave-uniform (the lanes advance in lock-step),
// so pop() is a scalar switch.  Costs 2.8 % of kernel time (profiles/r01_traffic_matrix.txt);
// -DREDUX_NO_LINE_QUEUE restores the 16-byte prefetch for A/B runs.
struct ChunkQueue {
    const uint8_t *base; // wave-uniform
    uint32_t       soff; // this lane's block offset
    uint32_t       last; // offset of the last 16-byte chunk the unrolled path reads
    uint32_t       nextp;
    uint32_t       idx;  // next chunk of c[] (wave-uniform)
    uint4          c[8], n[8];

    // chunks past the end re-read the last valid chunk (never used)
    __device__ __forceinline__ void prefetch()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t p = nextp + 16 * i;
            n[i] = *reinterpret_cast<const uint4 *>(base + soff + (p < last ? p : last));
        }
        nextp += 128;
    }
    // current line <- prefetched line, then put the line after it in flight
    __device__ __forceinline__ void swap()
    {
#pragma unroll
        for (int i = 0; i < 8; i++)
            c[i] = n[i];
        idx = 0;
        prefetch();
    }
    __device__ __forceinline__ void init(const uint8_t *b, uint32_t so, uint32_t e)
    {
        base = b; soff = so; last = e - 16; nextp = 0;
        prefetch();
        swap();
    }
    // the next 16 bytes of every lane's block
    __device__ __forceinline__ uint4 pop()
    {
        uint4 r;
        switch (idx) {
        case 0: r = c[0]; break;
        case 1: r = c[1]; break;
        case 2: r = c[2]; break;
        case 3: r = c[3]; break;
        case 4: r = c[4]; break;
        case 5: r = c[5]; break;
        case 6: r = c[6]; break;
        default: r = c[7]; break;
        }
        if (++idx == 8)
            swap();
        return r;
    }
};

#ifndef REDUX_KEEP8
#define REDUX_KEEP8 1
#endif
#ifndef REDUX_MODEL_PRIO
#define REDUX_MODEL_PRIO 3
#endif
#ifndef REDUX_ROWS // 1: the pair kernel writes ROW-major group areas (row r = dword r of the 64 lanes), k_compact_rows gathers them
#define REDUX_ROWS 0
#endif
constexpr int kPairStride = REDUX_ROWS ? 256 : 4;
#ifndef REDUX_MODEL_DEPTH
#define REDUX_MODEL_DEPTH 1
#endif
template <bool UPD>
__device__ __forceinline__ void model_chunk(const Tree<true> &T, uint2 *ring, uint32_t lane, const uint4 cur,
                                            uint32_t p, uint32_t nfreeze)
{
    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
    auto sym = [&](int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; };
    // software pipeline: symbol i+1's LDS ops are in flight while symbol i's sums are formed
    // (depths 2 and 3 measured no faster: the wave is bound by its own issue rate, not by LDS)
    constexpr int D = REDUX_MODEL_DEPTH;
    Tree<true>::Nodes q[D + 1];
#pragma unroll
    for (int d = 0; d < D; d++)
        q[d] = T.template issue<UPD>(sym(d), true);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s   = sym(i);
        const uint32_t nup = UPD ? p + i : nfreeze;
        // At the hand-over in the middle of the chunk the next symbol's eight LDS ops are issued
        // AFTER this symbol's ring write and stay in flight across the barrier (LDS ops of a wave
        // complete in order, so lgkmcnt <= 8 means the ring half is written).
        const bool late = REDUX_KEEP8 && i == 7;
        if (i + D < 16 && !late) {
            q[D] = T.template issue<UPD>(sym(i + D), true);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t lo, hi;
        T.finish(s, nup, q[0], lo, hi);
        ring[i * 64 + lane] = make_uint2(lo, hi);
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            q[D] = T.template issue<UPD>(sym(i + D), true);
            asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
        } else if ((i & 7) == 7)
            pair_barrier();
#pragma unroll
        for (int d = 0; d < D; d++)
            q[d] = q[d + 1];
    }
}

// MODE 0: adaptive chunk (reciprocals rc[p..p+15]); MODE 1: frozen chunk (rc[nfreeze]).
// MODE 0 takes the reciprocals of its first eight symbols in r[] and leaves those of the next
// chunk's first eight there: each half loads the following half's eight right after its own
// ring reads have arrived, at the start of an eight-symbol stretch with no lgkmcnt wait in it.
// (SMEM shares lgkmcnt with LDS and returns out of order, so any LDS wait or ring barrier also
// waits for every scalar load in flight; loaded at the top of the chunk, their miss latency sat
// in front of the first barrier.  Sixteen at a time would need 64 SGPRs: spills.)
template <bool FIXUP, int MODE, bool CB32>
__device__ __forceinline__ void coder_chunk(EncState &S, const uint2 *ring, uint32_t lane, uint32_t p,
                                            uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst, double (&r)[8])
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        pair_barrier();
        uint2 lh[8]; // the whole half at once: one LDS round trip per 8 symbols
#pragma unroll
        for (int i = 0; i < 8; i++)
            lh[i] = ring[(h * 8 + i) * 64 + lane];
        double rn[8];
        if (MODE == 0) {
            uint32_t zero; // opaque 0 that "depends" on the ring data: pins the loads behind the LDS wait
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero) : "v"(lh[0].x));
            const rc_ptr nb = rc + (p + 8u * h + 8u + zero); // the table has 32 entries of slack (geometry())
#pragma unroll
            for (int i = 0; i < 8; i++)
                rn[i] = nb[i];
        }
#ifdef REDUX_CODER_BRANCHY // the older form: a ballot branch inside every symbol
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t nup = MODE == 0 ? p + h * 8 + i : nfreeze;
            uint32_t       hi  = lh[i].y;
            asm volatile("" : "+v"(hi)); // keeps ISel from turning (u64 >> 32) -> f64 into a 64-bit conversion (+1 v_add_f64)
            encode_symbol_fast<FIXUP, CB32>(S, lh[i].x, hi, 257u + nup, MODE == 0 ? r[i] : rc[nfreeze], sh, wdst);
        }
#else
        // Eight symbols straight-line; the rare symbol whose pending run needs more than one
        // 32-bit append only raises a flag, and the half is then redone from the saved state
        // with the general encode_symbol (no per-symbol branch, no merge of two state versions).
        const EncState S0 = S;
        uint64_t       bad = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t nup = MODE == 0 ? p + h * 8 + i : nfreeze;
            uint32_t       hi  = lh[i].y;
            // keeps ISel from turning (u64 >> 32) -> f64 into a 64-bit conversion (+1 v_add_f64); converting
            // as signed avoids that too, but v_cvt_f64_i32 measured 9 % slower for the whole kernel
            asm volatile("" : "+v"(hi));
            bad |= encode_symbol_spec<FIXUP, CB32, kPairStride>(S, lh[i].x, hi, 257u + nup, MODE == 0 ? r[i] : rc[nfreeze], sh, wdst);
        }
        if (__builtin_expect(bad != 0, 0)) {
            S = S0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t nup = MODE == 0 ? p + h * 8 + i : nfreeze;
                encode_symbol<FIXUP, kPairStride>(S, lh[i].x, lh[i].y, 257u + nup, MODE == 0 ? r[i] : rc[nfreeze], sh, false, wdst, 0xFFFFFFFFu);
            }
        }
#endif
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                r[i] = rn[i];
        }
    }
}

// any chunk, rolled, every store checked against the slot limit
template <bool FIXUP>
__device__ __forceinline__ void coder_chunk_checked(EncState &S, const uint2 *ring, uint32_t lane, uint32_t p,
                                                    uint32_t nfreeze, rc_ptr rc, uint32_t sh, uint8_t *wdst,
                                                    uint32_t limit)
{
    for (uint32_t i = 0; i < 16; i++) {
        if ((i & 7) == 0)
            pair_barrier();
        const uint2    lh  = ring[i * 64 + lane];
        const uint32_t q   = p + i;
        const uint32_t nup = q < nfreeze ? q : nfreeze;
        encode_symbol<FIXUP, kPairStride>(S, lh.x, lh.y, 257u + nup, rc[nup], sh, false, wdst, limit);
    }
}

// The pair kernel needs every SIMD to hold exactly ONE model wave and ONE coder wave.  Where the
// two waves of a 128-thread workgroup land is up to the dispatcher: launched on an idle chip it
// alternates them perfectly, launched right after another kernel it puts two first-waves on
// some SIMDs (profiles/r01_final/placement_census.txt) -- two model waves at half speed each,
// which the whole lock-step kernel then waits for (0.5-3 ms of 13).  So the roles are not tied
// to the wave index: they are booked per CU at run time (REDUX_CLAIMS, below).
// (Tried instead: whole-CU workgroups of eight waves = four pairs, waves w and w+4 sharing a
// SIMD.  Placement is then perfect by construction, but the eight-wave s_barrier couples the
// four pairs and the kernel takes 14.05 ms against 12.8 ms.)
#ifndef REDUX_CLAIMS
#define REDUX_CLAIMS 1
#endif
constexpr uint32_t kClaimWords = 2048; // (xcc:3, se:3, sh:1, cu:4) -> one word per CU
constexpr uint32_t kPairDwords = Tree<true>::kDwords + kRingBytes / 4;

template <bool FIXUP, bool CB32>
__global__ void __launch_bounds__(128) k_encode_pair(EncArgs a)
{
    __shared__ uint32_t lds[kPairDwords];
    const uint32_t w8   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // first or second wave of the workgroup
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk0 = (uint64_t)blockIdx.x * a.lanes;
    const uint64_t blk  = blk0 + lane;
    const bool     live = lane < a.lanes && blk < a.nblocks;

    for (uint32_t i = threadIdx.x; i < Tree<true>::kDwords / 4; i += 128)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    uint32_t role = w8;
    uint2   *ring = reinterpret_cast<uint2 *>(lds + Tree<true>::kDwords);
#if REDUX_CLAIMS
    // claims[cu] counts the model waves (bits 4s..4s+3) and coder waves (bits 16+4s..) booked on
    // SIMD s of that CU.  A workgroup whose waves sit on SIMDs (s0, s1) books (model, coder) =
    // (s0, s1), or (s1, s0) when that collides with fewer roles already booked, and returns its
    // booking when its coder wave ends.  Greedy, so not always perfect, but on an idle chip the
    // dispatcher's own choice is kept and after a compaction every SIMD still gets (1, 1).
    uint32_t claim_delta = 0; // (a VGPR on purpose: it is live across the whole kernel)
    {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // (the last 16 bytes of the ring: the model wave writes there only after the first hand-off)
        volatile uint32_t *book = reinterpret_cast<volatile uint32_t *>(ring) + kRingBytes / 4 - 4;
        if (lane == 0)
            book[w8] = (hwid >> 4) & 3u; // my SIMD
        __syncthreads();
        uint32_t *claim_word = a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu)); // (xcc, se, sh, cu)
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
#else
    __syncthreads();
#endif
    const uint32_t wave = role; // 0 = model, 1 = coder
    Tree<true> T;
    T.init(lds, lane);

    uint32_t len = 0;
    if (live) {
        const uint64_t rem = a.in_len - blk * a.block_size;
        len                = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    const uint8_t *wsrc  = a.in + blk0 * a.block_size;
    const uint32_t soff  = live ? lane * a.block_size : 0u;
#if REDUX_ROWS
    // row-major group area: dword r of lane l at wdst + 256 r + 4 l (dead lanes own a column too).
    // The areas are an ODD number of 128-byte lines apart: all groups write row r at about the
    // same time, and an even stride folds those lines onto a fraction of the L2 sets.
    uint8_t       *wdst  = a.slots + (uint64_t)blockIdx.x * (64 * a.slot_bytes + 128);
    const uint32_t off0  = lane * 4u;
    const uint32_t limit = off0 + (a.slot_cap / 4u) * 256u;
#else
    uint8_t       *wdst  = a.slots + blk0 * a.slot_bytes;
    const uint32_t off0  = live ? lane * (uint32_t)a.slot_bytes : (uint32_t)(a.nblocks - blk0) * (uint32_t)a.slot_bytes;
    const uint32_t limit = off0 + a.slot_cap;
#endif

    const uint32_t minlen  = __builtin_amdgcn_readfirstlane(wave_min(live ? len : 0xFFFFFFFFu));
    const uint32_t maxlen  = __builtin_amdgcn_readfirstlane(wave_max(live ? len : 0u));
    const uint32_t sh      = 32 - a.code_bits;
    const uint32_t nfreeze = a.nfreeze;
    const rc_ptr   rc      = (rc_ptr)a.rc;
    constexpr uint32_t kChunkBudget = (16 * 4 + 32) * (kPairStride / 4);

    // both waves derive the same chunk schedule from wave-uniform values
    uint32_t main_end = 0;
    if (a.aligned16 && minlen != 0xFFFFFFFFu && minlen > 16)
        main_end = (minlen - 1) & ~15u;
    const uint32_t a_end = main_end < (nfreeze & ~15u) ? main_end : (nfreeze & ~15u); // adaptive chunks
    const uint32_t m_end = (a_end < main_end && a_end < nfreeze) ? a_end + 16 : a_end; // freeze-crossing chunk

    EncState S;
    enc_init(S, off0);

    if (wave == 0) {
        // ---------------- model wave ----------------
        // The model wave is the pair's critical path (it works ~590 cycles per symbol, the coder wave
        // ~430 and then waits at the ring barrier), but the SIMD's arbiter serves the two waves
        // round-robin: raising the model wave's issue priority lets it run at nearly the lone-wave
        // rate while the coder wave fills the gaps.  15.96 -> 14.0 ms (REDUX_MODEL_PRIO=0 for the A/B).
        __builtin_amdgcn_s_setprio(REDUX_MODEL_PRIO);
        if (main_end) {
            uint32_t p = 0;
#ifndef REDUX_NO_LINE_QUEUE
            ChunkQueue Q;
            Q.init(wsrc, soff, main_end);
#define NEXT_CHUNK() Q.pop()
#else
            uint4 cur = *reinterpret_cast<const uint4 *>(wsrc + soff);
            auto  next_chunk = [&](uint32_t pp) {
                const uint4 r = cur;
                if (pp + 16 < main_end)
                    cur = *reinterpret_cast<const uint4 *>(wsrc + soff + pp + 16);
                return r;
            };
#define NEXT_CHUNK() next_chunk(p)
#endif
            for (; p < a_end; p += 16)
                model_chunk<true>(T, ring, lane, NEXT_CHUNK(), p, nfreeze);
            for (; p < m_end; p += 16) { // rolled: the update stops in the middle of this chunk
                (void)NEXT_CHUNK();
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t q   = p + i;
                    const uint32_t nup = q < nfreeze ? q : nfreeze;
                    uint32_t       lo, hi;
                    T.template get_frequency<true>(wsrc[soff + q], nup, q < nfreeze, lo, hi);
                    ring[i * 64 + lane] = make_uint2(lo, hi);
                    if ((i & 7) == 7)
                        pair_barrier();
                }
            }
            for (; p < main_end; p += 16)
                model_chunk<false>(T, ring, lane, NEXT_CHUNK(), p, nfreeze);
#undef NEXT_CHUNK
        }
    } else {
        // ---------------- coder wave ----------------
#ifdef REDUX_CODER_PRIO
        __builtin_amdgcn_s_setprio(REDUX_CODER_PRIO);
#endif
        uint32_t p = 0;
        double   r[8];      // reciprocals of the first eight symbols of chunk r_at
        uint32_t r_at = ~0u;
        for (; p < main_end; p += 16) {
            if (__builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit) || (p >= a_end && p < m_end))
                coder_chunk_checked<FIXUP>(S, ring, lane, p, nfreeze, rc, sh, wdst, limit);
            else if (p < a_end) {
                if (r_at != p) { // first chunk, or the previous one took the checked path
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        r[i] = rc[p + i];
                }
                coder_chunk<FIXUP, 0, CB32>(S, ring, lane, p, nfreeze, rc, sh, wdst, r);
                r_at = p + 16;
            } else
                coder_chunk<FIXUP, 1, CB32>(S, ring, lane, p, nfreeze, rc, sh, wdst, r);
        }
    }
    __syncthreads(); // the model wave's last updates are in LDS before the tail reads the tree
#ifdef REDUX_STAMPS
    if (lane == 0) {
        lds64p st = (lds64p)(uintptr_t)(w8 * 32);
        unsigned long long *dstp = reinterpret_cast<unsigned long long *>(a.slots + a.nblocks * a.slot_bytes) + (blockIdx.x * 2 + wave) * 4;
        uint32_t hwid, xcc; // where this wave ran: (xcc, se, sh, cu, simd) -- the pair needs one model and one coder wave per SIMD
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dstp[0] = st[1]; dstp[1] = st[2]; dstp[2] = st[3];
        dstp[3] = wave | ((unsigned long long)(hwid & 0xFFFFu) << 8) | ((unsigned long long)(xcc & 0xFu) << 24);
    }
#endif
    if (wave == 0)
        return;

    // Tail (coder wave only): symbol by symbol with per-lane predicates.
    for (uint32_t p = main_end; p <= maxlen; p++) {
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   r   = rc[nup];
        const uint32_t c   = 257u + nup;
        if (live && p < len) {
            uint32_t lo, hi;
            T.template get_frequency<true>(wsrc[soff + p], nup, p < nfreeze && p + 1 != len, lo, hi);
            encode_symbol<FIXUP, kPairStride>(S, lo, hi, c, r, sh, false, wdst, limit);
        } else if (live && p == len) {
            const uint32_t shifts = encode_symbol<FIXUP, kPairStride>(S, c - 1, c, c, r, sh, true, wdst, limit);
            const uint32_t size   = encode_finish<kPairStride>(S, shifts, a.code_bits, off0, wdst, limit);
            a.sizes[blk]  = size;
            a.status[blk] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
        }
    }
#if REDUX_CLAIMS
    if (lane == 0) { // waves do not migrate: the same CU as at the start
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_sub(a.claims + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xFFu)), claim_delta, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

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
};

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
    const uint64_t blk  = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = blk < a.nblocks;

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
    uint8_t       *dst         = a.out + (live ? blk : 0) * (uint64_t)a.block_size;
    const uint32_t capn        = a.block_size;
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
        const double   rc  = rcp[nup];
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
            } else if (p >= capn) {
                st   = REDUX_OUTPUT_TOO_SMALL;
                done = true;
            } else {
                const double   Y     = __builtin_fma((double)R1, rc, rc);
                const uint32_t nlow  = low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
                const uint32_t nhigh = low + (scale_div<FIXUP>(R1, Y, hi, c) << sh) - 1u;
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
// Lock-step decoder (the default for u16 trees, count < 2^17).  Same results as k_decode; what changes is the instruction count of a step, which is what a lone wave
// per SIMD pays for:
//   * own tree layout: lane l owns dword column l; dword k of the column holds nodes 2k (low
//     half) and 2k+1 (high half): byte address (k << 8) | 4*l.  Levels 1-7 are even nodes, so
//     their half is static (low); level 0 is always a high half.  No per-lane half select.
//   * the descent keeps q = ~rem.  For a node value t, q2 = q + t is ~(rem - t): its top bit
//     is the "go right" flag, the new q is max_u32(q, q2) (q2 wraps to a small number when the
//     probe fails), and the flags are shifted into the symbol by v_alignbit.  cum(s+1) falls
//     out of the same probes: it is the upper boundary of the LAST level where the descent
//     went left, i.e. v + 1 + min_u32 over the levels of q2 (failed probes give the small
//     values and the boundary only shrinks on the way down; the virtual root probe against
//     tree[256] = count - 1 seeds the minimum).  Five VALU ops per level, no second masked sum.
//   * all 64 lanes stay in lock-step while nothing exceptional happens: the step is computed
//     for every lane, and ONE ballot (EOF symbol, low == high, stream exhausted) decides whether
//     it is committed without predication.  The first exceptional step leaves the fast loop
//     with nothing committed and the predicated loop below redoes it and finishes the blocks.
//   * the bit reader refills without a branch: the dword at rpo is (re)loaded every step, a
//     whole step before it can be needed, and consumed when fewer than 33 bits are left.
// --------------------------------------------------------------------------------------
#ifndef REDUX_DEC_DUP
#define REDUX_DEC_DUP 0
#endif
#ifdef REDUX_DEC_CENSUS // diagnostic build: where each decode wave ran (tools/dec_census.py)
__device__ uint32_t g_dec_hw[4096];
#endif
#ifdef REDUX_DEC_STAMPS // diagnostic build: cycle stamps inside the lock-step step (tools/dec_stamps.sh)
__device__ uint64_t g_dec_ts[8];
#define DEC_STAMP(i, dep)                                                                                              \
    {                                                                                                                  \
        uint64_t t_;                                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                        \
        dec_ts[i] += t_ - dec_t0;                                                                                      \
        dec_t0 = t_;                                                                                                   \
    }
#define DEC_STAMP_ARGS , uint64_t (&dec_ts)[8], uint64_t &dec_t0
#define DEC_STAMP_PASS , dec_ts, dec_t0
#else
#define DEC_STAMP(i, dep)
#define DEC_STAMP_ARGS
#define DEC_STAMP_PASS
#endif

struct DecFound {
    uint32_t s, lo, hi;
    uint32_t eofq; // top bit set: v >= count - 1, the first probe of get_symbol fails -> EOF (adaptive_tree.rs:116)
};

// The seven nodes of levels 7, 6, 5 (128; 64, 192; 32, 96, 160, 224) are at fixed positions,
// so a decoder lane keeps them in VGPRs: the first three probes of every descent need no LDS
// round trip, and their updates are compare + add-with-carry instead of LDS atomics.  (A
// lock-step decoder wave is alone on its SIMD and the four waves of a CU share one LDS
// pipeline: 8 cycles per ds_read_b32 and 16 per ds_add, tools/ubench/lone.hip.)
struct DecTop {
    uint32_t n128, n64, n192, n32, n96, n160, n224; // full tree values (lowbit + increments): u32, no overflow to think about
};
__device__ __forceinline__ DecTop dec_top_new() { return {128u, 64u, 64u, 32u, 32u, 32u, 32u}; }

// get_symbol (adaptive_tree.rs:115-136) + the high end of get_frequency (:105-113), layout above.
// Safe for any v (lanes that are already done run it on garbage): every address stays inside
// the 32 KiB tree.
__device__ __forceinline__ DecFound dec_search(const uint32_t *lds, uint32_t L, const DecTop &T, uint32_t v,
                                               uint32_t c DEC_STAMP_ARGS)
{
    auto ld = [&](uint32_t byte) { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + byte); };
    uint32_t q = ~v, hq = q + (c - 1u), bits = 0, q2;
    DecFound f;
    f.eofq = hq;
#define REDUX_DEC_LEVEL(t)                                                                                             \
    left = __builtin_uadd_overflow(q, (t), &q2); /* carries exactly when the probe fails (rem < t: go left) */          \
    bits = __builtin_amdgcn_alignbit(bits, q2, 31);                                                                    \
    q    = q > q2 ? q : q2;                                                                                            \
    hq   = hq < q2 ? hq : q2;
    bool left;
    // levels 7, 6, 5: registers (full tree values: lowbit + increments)
    REDUX_DEC_LEVEL(T.n128)
    const uint32_t x6 = left ? T.n64 : T.n192, c5l = left ? T.n32 : T.n160, c5r = left ? T.n96 : T.n224;
    REDUX_DEC_LEVEL(x6)
    const uint32_t x5 = left ? c5l : c5r;
    REDUX_DEC_LEVEL(x5)
    DEC_STAMP(2, bits)
    // round B: levels 4, 3 under the prefix i = bits << 5: nodes i+16; i+8, i+24 (three dwords)
    uint32_t       ib  = ((bits & 7u) << 12) | L;
    const uint32_t w16 = ld(ib + (16u << 7));
    const uint32_t w8 = ld(ib + (8u << 7)), w24 = ld(ib + (24u << 7));
    REDUX_DEC_LEVEL((w16 & 0xFFFFu) + 16u)
    const uint32_t x3 = left ? w8 : w24;
    REDUX_DEC_LEVEL((x3 & 0xFFFFu) + 8u)
    DEC_STAMP(3, bits)
    // round C: levels 2, 1, 0 under i = bits << 3.  The four dwords i/2 .. i/2+3 hold nodes
    // (i, i+1), (i+2, i+3), (i+4, i+5), (i+6, i+7): all seven candidates.
    ib                = ((bits & 31u) << 10) | L;
    const uint32_t d0 = ld(ib), d1 = ld(ib + 256u), d2 = ld(ib + 512u), d3 = ld(ib + 768u);
    REDUX_DEC_LEVEL((d2 & 0xFFFFu) + 4u) // node i+4
    const uint32_t e1 = left ? d1 : d3;  // level 1: node i+2 or i+6 (low halves)
    const uint32_t e0 = left ? d0 : d2;  // level 0 if level 1 goes left: node i+1 or i+5 (high halves)
    REDUX_DEC_LEVEL((e1 & 0xFFFFu) + 2u)
    const uint32_t x0 = left ? e0 : e1;  // ... if it goes right: node i+3 or i+7, the high half of level 1's dword
    REDUX_DEC_LEVEL((x0 >> 16) + 1u)
    DEC_STAMP(4, bits)
#undef REDUX_DEC_LEVEL
    f.s  = bits & 0xFFu;
    f.lo = v + q + 1u;  // v - rem
    f.hi = v + hq + 1u; // upper boundary of the last level that went left
    return f;
}

// update(s+1) (adaptive_tree.rs:83-92): +1 on the levels where bit b of s is clear.  Levels 7-5
// live in registers: node e of level b is incremented iff s lies in [e - 2^b, e), an unsigned
// range compare + add-with-carry; levels 4-0 are fire-and-forget ds_add_u32.
__device__ __forceinline__ void dec_update(uint32_t *lds, const uint32_t (&A)[8], DecTop &T, uint32_t s)
{
    T.n128 += s < 128u ? 1u : 0u;
    T.n64 += s < 64u ? 1u : 0u;
    T.n192 += (s - 128u) < 64u ? 1u : 0u;
    T.n32 += s < 32u ? 1u : 0u;
    T.n96 += (s - 64u) < 32u ? 1u : 0u;
    T.n160 += (s - 128u) < 32u ? 1u : 0u;
    T.n224 += (s - 192u) < 32u ? 1u : 0u;
    const uint32_t ss = s << 7, ns = ~s;
#pragma unroll
    for (int b = 0; b < 5; b++) {
        const uint32_t keep = b ? (((0xFFu << b) & 0xFFu) << 7) : (0xFEu << 7);
        const uint32_t addr = (ss & keep) | A[b];
        const uint32_t inc  = b ? ((ns >> b) & 1u) : ((ns & 1u) << 16);
        __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + addr), inc, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// value = floor(((V - low + 1) * count - 1) / range) (codec.rs:129-131) in f64.  Numerator nd
// (< 2^49) and range xd (an integer in [1, 2^32]) are exact.  r' = v_rcp_f64(xd) * (1 - 2^-22):
// the raw v_rcp_f64 of gfx950 is within 2^-24 (measured: 2^-24.4) of 1/xd for EVERY integer xd in [1, 2^32]
// (checked exhaustively on the device by redux_debug_rcp_check, tests/test_gpu_parity.py), so
// (1 - 2^-21)/xd <= r' <= 1/xd and, the quotient being < 2^17.1, the truncated product is q or
// q - 1; one exact f64 remainder (fma; v * xd < 2^50) adds the 1 back.
__device__ __forceinline__ uint32_t dec_value(double R1d, uint32_t Vd, double cd, double cdm1)
{
    const double xd = R1d + 1.0;
    const double nd = __builtin_fma((double)Vd, cd, cdm1); // (Vd+1)*c - 1, exact (< 2^49)
#ifdef REDUX_DEC_NEWTON // the older form: one Newton step, bias 2^-40
    double r = __builtin_amdgcn_rcp(xd);
    r        = __builtin_fma(__builtin_fma(-xd, r, 1.0), r, r);
    uint32_t v = (uint32_t)(nd * (r * 0.99999999999909050530));
#else
    uint32_t v = (uint32_t)(nd * (__builtin_amdgcn_rcp(xd) * 0.99999976158142089844)); // 1 - 2^-22
#endif
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
    uint32_t sbits;      // stream length in bits while the block is live, 0 once it is finished
    int32_t  st;
};

// Per-lane, predicated end of a step: decompress_symbol after the model answered
// (codec.rs:133-161) + decompress_stream's emission (:170-172).  `may_update`: the model is not
// frozen; `room`: p < block capacity.
template <bool CB32>
__device__ __forceinline__ void dec_commit_careful(DecLane &S, DecTop &T, const DecFound &f, uint32_t *lds, const uint32_t (&A)[8],
                                                   uint32_t R1, double R1d, double rc, uint32_t c, uint32_t sh,
                                                   uint32_t stream_bits, uint32_t p, bool may_update, bool room,
                                                   bool aligned4, uint8_t *dst)
{
    if ((int32_t)S.dflag < 0)
        return;
    if ((int32_t)f.eofq < 0) { // codec.rs:136-138: returns before any renormalisation
        S.dflag = 0x80000000u;
        S.sbits = 0;
        S.n_out = p;
        return;
    }
    if (!room) {
        S.st    = REDUX_OUTPUT_TOO_SMALL;
        S.dflag = 0x80000000u;
        S.sbits = 0;
        S.n_out = p;
        return;
    }
    if (may_update)
        dec_update(lds, A, T, f.s);
    const double   Y      = __builtin_fma(R1d, rc, rc);
    const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, f.lo, c) << sh);
    const uint32_t nihigh = 0u - (S.low + (scale_div<false>(R1, Y, f.hi, c) << sh));
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
        S.sbits = 0;
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

template <bool CB32>
__global__ void __launch_bounds__(64) k_decode_lock(DecArgs a)
{
    __shared__ uint32_t lds[128 * 64 + 32 * 64]; // tree (32 KiB) + stream ring (8 KiB): four groups fill the CU's 160 KiB
    const uint32_t lane = threadIdx.x;
    const uint64_t blk  = (uint64_t)blockIdx.x * 64 + lane;
    const bool     live = blk < a.nblocks;

    for (uint32_t i = lane; i < 128 * 64 / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t L = lane * 4u;
    uint32_t       A[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        A[b] = (b ? (1u << (b + 7)) : 0u) | L;
        asm volatile("" : "+v"(A[b]));
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
    uint8_t       *dst         = a.out + (live ? blk : 0) * (uint64_t)a.block_size;
    const uint32_t capn        = a.block_size;
    const rc_ptr   rcp         = (rc_ptr)a.rc;
    const uint32_t nfreeze     = a.nfreeze;
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
    auto rd = [&](uint32_t o) { return gin[o < rpo_last ? o : rpo_last]; };
    constexpr uint32_t RB = 128 * 64 * 4;
    auto ring_write = [&](uint32_t chunk, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
        uint32_t *q = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + RB + ((chunk & 7u) << 10) + L);
        q[0] = x0; q[64] = x1; q[128] = x2; q[192] = x3;
    };
    auto ring_read = [&](uint32_t d) {
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + RB + ((d & 31u) << 8) + L);
    };
    uint32_t rpo = 2, wr = 0;
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
        }
        S.bbits = (((uint64_t)d0 << 32) | d1) << skip;
        S.bcnt  = 64 - skip;
    }
    uint32_t fetched = ring_read(rpo);
    bool     pend = false; // a chunk is in flight: requested by the previous group, not yet in the ring
    uint32_t pend_chunk = 0;
    u32x4    ldq = {0, 0, 0, 0};
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
    S.sbits = (int32_t)S.dflag < 0 ? 0u : stream_bits;
    S.n_out = 0;
    S.obuf  = 0;
    uint32_t stored = 0; // bytes [0, stored) of the block are in memory
    uint32_t staged = 0; // bytes [stored, staged) are whole dwords waiting in oq (newest in .w)
    uint4    oq     = make_uint4(0, 0, 0, 0);
    uint32_t p      = 0;
    DecTop   T      = dec_top_new();

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
#define REDUX_DEC_RETIRE                                                                                               \
    if (pend)                                                                                                          \
        ring_write(pend_chunk, ldq.x, ldq.y, ldq.z, ldq.w);
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
        const bool room = (int32_t)(4u * wr - rpo) <= 28;                                                              \
        const bool tail = 4u * wr + 3u > rpo_last;                                                                     \
        pend       = room;                                                                                             \
        pend_chunk = wr;                                                                                               \
        if (room && !tail)                                                                                             \
            ldq = *reinterpret_cast<gptr4>(gin + 4u * wr);                                                             \
        if (__builtin_amdgcn_ballot_w64(room && tail) != 0) {                                                          \
            if (room && tail) {                                                                                        \
                ldq.x = rd(4u * wr);                                                                                   \
                ldq.y = rd(4u * wr + 1u);                                                                              \
                ldq.z = rd(4u * wr + 2u);                                                                              \
                ldq.w = rd(4u * wr + 3u);                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        wr += room ? 1u : 0u;                                                                                          \
    }

    // ---------------- lock-step groups of four symbols ----------------
    // While p < min(capacity, freeze point) every step updates the model and has room for its
    // symbol.  A step is computed for all 64 lanes; if no lane is finished, reaches the EOF
    // symbol, collapses to low == high or runs out of stream (one v_or3 + one compare on sign
    // bits), it is committed without predication; otherwise the careful per-lane commit runs.
    const uint32_t pfast = capn < nfreeze ? capn : nfreeze;
#ifdef REDUX_DEC_CENSUS
    if (lane == 0 && blockIdx.x < 4096) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_dec_hw[blockIdx.x] = 0x80000000u | ((xcc & 0xFu) << 16) | (hwid & 0xFFFFu);
    }
#endif
#ifdef REDUX_DEC_STAMPS
    uint64_t dec_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dec_t0 = clock64();
#endif
    if (aligned4) {
        double cdm1 = 256.0, cd = 257.0;
        // The group's four reciprocals are loaded a group ahead with VECTOR loads (every lane the
        // same 32 bytes), behind the ring's chunk request: their latency is covered by the one
        // vmcnt wait of the next group.  A scalar load would share lgkmcnt with the LDS, return
        // out of order and so sit in front of the next LDS wait wherever it is issued.
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f64x4 *grc4;
        const grc4 rcv = (grc4)(uintptr_t)a.rc; // 256-byte aligned workspace, p a multiple of 4
        f64x4      rcg = rcv[0], rcn;
        asm volatile("" : "+v"(rcg)); // arrived before the loop: no in-loop wait inherits this load
        for (; p + 4 <= pfast; p += 4) {
            if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
                break;
            REDUX_DEC_RETIRE
            REDUX_DEC_STORE
            REDUX_DEC_REQUEST
            rcn = rcv[(p >> 2) + 1]; // the table has 32 entries of slack (geometry())
#pragma unroll
            for (int K = 0; K < 4; K++) {
                const double   rc = rcg[K];
                const uint32_t c  = 257u + p + K;
                REDUX_DEC_READER
                DEC_STAMP(0, S.bcnt)
                const uint32_t R1  = (~(S.ihigh + S.low)) >> sh;
                const uint32_t Vd  = (S.W - S.low) >> sh;
                const double   R1d = (double)R1;
#if REDUX_DEC_DUP == 1 // timing experiments: run one part of the step twice, results unchanged
                uint32_t Vd_ = Vd;
                {
                    const uint32_t v0 = dec_value(R1d, Vd, cd, cdm1);
                    asm volatile("" : "+v"(Vd_) : "v"(v0));
                }
                const uint32_t v = dec_value(R1d, Vd_, cd, cdm1);
#else
                const uint32_t v   = dec_value(R1d, Vd, cd, cdm1);
#endif
                DEC_STAMP(1, v)
#if REDUX_DEC_DUP == 2
                uint32_t v_ = v;
                {
                    const DecFound f0 = dec_search(lds, L, T, v, c DEC_STAMP_PASS);
                    asm volatile("" : "+v"(v_) : "v"(f0.s), "v"(f0.lo), "v"(f0.hi));
                }
                const DecFound f = dec_search(lds, L, T, v_, c DEC_STAMP_PASS);
#else
                const DecFound f   = dec_search(lds, L, T, v, c DEC_STAMP_PASS);
#endif
                // narrowing + renormalisation (codec.rs:133-161), all lanes
                const double   Y      = __builtin_fma(R1d, rc, rc);
#if REDUX_DEC_DUP == 3
                uint32_t lo_ = f.lo;
                {
                    const uint32_t a0 = S.low + (scale_div<false>(R1, Y, f.lo, c) << sh);
                    const uint32_t b0 = 0u - (S.low + (scale_div<false>(R1, Y, f.hi, c) << sh));
                    const uint32_t x0 = ~(a0 ^ b0);
                    uint32_t       k0;
                    asm("v_ffbh_u32 %0, %1" : "=v"(k0) : "v"(x0));
                    const uint32_t t0 = ((a0 << (k0 & 31u)) & (b0 << (k0 & 31u))) << 1;
                    const uint32_t j0 = (uint32_t)__builtin_clz(~t0);
                    asm volatile("" : "+v"(lo_) : "v"(j0));
                }
                const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, lo_, c) << sh);
#else
                const uint32_t nlow   = S.low + (scale_div<false>(R1, Y, f.lo, c) << sh);
#endif
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
                const uint32_t e     = f.eofq | k | (S.sbits - cons2); // sbits is 0 for a finished lane, cons2 > 0
                DEC_STAMP(5, e)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)e < 0) == 0, 1)) {
                    dec_update(lds, A, T, f.s);
                    S.low      = (low2 << j) & 0x7FFFFFFFu;
                    S.ihigh    = (ih2 << j) & 0x7FFFFFFFu;
                    S.consumed = cons2;
                    const uint32_t nxt  = (uint32_t)(S.bbits >> 32);
                    const uint64_t comb = CB32 ? (((uint64_t)S.W << 32) | nxt) : (((uint64_t)S.W << 32) | ((uint64_t)nxt << sh));
                    const uint32_t h2   = (uint32_t)((comb << n) >> 32);
                    const uint32_t h1   = S.W << k;
                    S.W = ((h2 & 0x7FFFFFFFu) | (h1 & 0x80000000u)) & (0xFFFFFFFFu << sh);
                    S.bbits <<= n;
                    S.bcnt -= n;
                    S.obuf |= f.s << (8 * K);
                } else {
                    dec_commit_careful<CB32>(S, T, f, lds, A, R1, R1d, rc, c, sh, stream_bits, p + K, true, true, true, dst);
                }
                cdm1 = cd;
                cd += 1.0;
                DEC_STAMP(6, S.low + S.W)
            }
            rcg = rcn;
        }
    }
#ifdef REDUX_DEC_STAMPS
    if (blockIdx.x == 7 && lane == 0)
        for (int i = 0; i < 8; i++)
            g_dec_ts[i] = i < 7 ? dec_ts[i] : p;
#endif
    // ---------------- remaining steps (EOF symbol, frozen model, unaligned output) ----------------
    for (;; p++) {
        if (__builtin_amdgcn_ballot_w64((int32_t)S.dflag >= 0) == 0)
            break;
        const uint32_t nup = p < nfreeze ? p : nfreeze;
        const double   rc  = rcp[nup];
        const uint32_t c   = 257u + nup;
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
        const DecFound f   = dec_search(lds, L, T, v, c DEC_STAMP_PASS);
        dec_commit_careful<CB32>(S, T, f, lds, A, R1, R1d, rc, c, sh, stream_bits, p, p < nfreeze, p < capn, aligned4, dst);
    }
#undef REDUX_DEC_READER
#undef REDUX_DEC_RETIRE
#undef REDUX_DEC_STORE
#undef REDUX_DEC_REQUEST
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

__global__ void k_summarize(const int32_t *status, uint64_t nblocks, int32_t *summary)
{
    uint32_t bad = 0;
    uint64_t first = ~0ull;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += (uint64_t)gridDim.x * blockDim.x)
        if (status[b] != REDUX_OK) {
            bad++;
            if (first == ~0ull)
                first = b;
        }
    if (bad) {
        atomicAdd(&summary[1], (int32_t)bad);
        atomicCAS(&summary[0], REDUX_OK, status[first]);
    }
}

// ======================================================================================
// sizes -> offsets, status summary
// ======================================================================================
struct ScanArgs {
    const uint32_t *sizes;
    const int32_t  *status;
    uint64_t       *offsets; // nblocks + 1
    int32_t        *summary; // may be null: [first bad status, #bad]
    uint64_t        nblocks;
};

__global__ void __launch_bounds__(1024) k_scan_sizes(ScanArgs a)
{
    __shared__ uint64_t part[1024];
    __shared__ uint32_t bad_cnt;
    __shared__ uint64_t bad_first; // (index << 8) | status, minimised
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        bad_cnt   = 0;
        bad_first = ~0ull;
    }
    const uint64_t per = (a.nblocks + 1023) / 1024;
    const uint64_t b0  = per * tid < a.nblocks ? per * tid : a.nblocks;
    const uint64_t b1  = b0 + per < a.nblocks ? b0 + per : a.nblocks;
    uint64_t       sum = 0;
    uint32_t       nb  = 0;
    uint64_t       fb  = ~0ull;
    // Up to 64 blocks per thread in whole quads (the 65,536-block configuration): the sizes stay
    // in registers between the two passes and move as 16-byte loads, all in flight at once,
    // instead of 3 x 64 dependent 4-byte accesses per thread.
    const bool quads = per <= 64 && (per & 3) == 0 && (a.nblocks % per) == 0 &&
                       ((((uintptr_t)a.sizes) | ((uintptr_t)a.status) | ((uintptr_t)a.offsets)) & 15) == 0;
    uint4      sz[16];
    if (quads) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(a.sizes + b0);
        const int4  *t4 = reinterpret_cast<const int4 *>(a.status + b0);
        int4         stv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool in = b0 + 4 * i < b1;
            sz[i]  = in ? s4[i] : make_uint4(0, 0, 0, 0);
            stv[i] = in ? t4[i] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            sum += (uint64_t)sz[i].x + sz[i].y + sz[i].z + sz[i].w;
            const int32_t st4[4] = {stv[i].x, stv[i].y, stv[i].z, stv[i].w};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (st4[k] != REDUX_OK) {
                    nb++;
                    if (fb == ~0ull)
                        fb = ((b0 + 4 * i + k) << 8) | (uint32_t)st4[k];
                }
        }
    } else {
        for (uint64_t b = b0; b < b1; b++) {
            sum += a.sizes[b];
            const int32_t st = a.status[b];
            if (st != REDUX_OK) {
                nb++;
                if (fb == ~0ull)
                    fb = (b << 8) | (uint32_t)st;
            }
        }
    }
    part[tid] = sum;
    __syncthreads();
    if (nb) {
        atomicAdd(&bad_cnt, nb);
        atomicMin((unsigned long long *)&bad_first, (unsigned long long)fb);
    }
    // Hillis-Steele inclusive scan over the 1024 partials
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    if (quads) {
        ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(a.offsets + b0); // b0 is a multiple of 4: 16-byte aligned
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (b0 + 4 * i < b1) {
                const uint64_t r1 = run + sz[i].x, r2 = r1 + sz[i].y, r3 = r2 + sz[i].z;
                o2[2 * i]     = make_ulonglong2(run, r1);
                o2[2 * i + 1] = make_ulonglong2(r2, r3);
                run           = r3 + sz[i].w;
            }
    } else {
        for (uint64_t b = b0; b < b1; b++) {
            a.offsets[b] = run;
            run += a.sizes[b];
        }
    }
    if (tid == 1023)
        a.offsets[a.nblocks] = part[1023];
    if (tid == 0 && a.summary) {
        a.summary[0] = bad_cnt ? (int32_t)(bad_first & 0xFF) : REDUX_OK;
        a.summary[1] = (int32_t)bad_cnt;
    }
}

// ======================================================================================
// compaction: slot b [0, size_b) -> out + offsets[b]
// ======================================================================================
struct CompactArgs {
    const uint8_t  *slots;
    uint64_t        slot_bytes;
    const uint64_t *offsets;
    uint8_t        *out;
    uint64_t        out_cap;
    int32_t        *status;
    int32_t        *summary;
    uint64_t        nblocks;
    const uint32_t *mode;     // 0: linear slots (k_compact), != 0: row-major group areas (k_compact_rows)
    uint32_t        cap_rows; // rows of a group area
};

__global__ void __launch_bounds__(256) k_compact(CompactArgs a)
{
    const uint64_t b = blockIdx.x;
    if (b >= a.nblocks || *a.mode != 0)
        return;
    const uint64_t o0 = a.offsets[b], o1 = a.offsets[b + 1];
    const uint32_t tid = threadIdx.x;
    if (o1 > a.out_cap) { // the dense buffer is too small for this block: report, never write
        if (tid == 0) {
            if (a.status[b] == REDUX_OK)
                a.status[b] = REDUX_OUTPUT_TOO_SMALL;
            if (a.summary) {
                atomicCAS(&a.summary[0], REDUX_OK, REDUX_OUTPUT_TOO_SMALL);
                atomicAdd(&a.summary[1], 1);
            }
        }
        return;
    }
    const uint32_t n   = (uint32_t)(o1 - o0);
    const uint8_t *src = a.slots + b * a.slot_bytes; // 16-byte aligned
    uint8_t       *dst = a.out + o0;

    // head: bytes up to the first 16-byte boundary of dst
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n)
        head = n;
    if (tid < head)
        dst[tid] = src[tid];
    // body: 16-byte dst chunks; the source is misaligned by the uniform amount `head`
    const uint32_t nchunks = (n - head) >> 4;
    const uint32_t dq = head >> 2, r = head & 3;
    const uint4   *s16 = reinterpret_cast<const uint4 *>(src);
    uint4         *d16 = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t i = tid; i < nchunks; i += 256) {
        const uint4    A = s16[i], B = s16[i + 1]; // slot padding keeps i+1 inside the slot
        const uint32_t d[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
        uint32_t       v[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            v[k] = dq == 0 ? d[k] : dq == 1 ? d[k + 1] : dq == 2 ? d[k + 2] : d[k + 3];
        uint4 o;
        o.x = __builtin_amdgcn_alignbyte(v[1], v[0], r);
        o.y = __builtin_amdgcn_alignbyte(v[2], v[1], r);
        o.z = __builtin_amdgcn_alignbyte(v[3], v[2], r);
        o.w = __builtin_amdgcn_alignbyte(v[4], v[3], r);
        d16[i] = o;
    }
    // tail
    const uint32_t done = head + (nchunks << 4);
    if (tid < n - done)
        dst[done + tid] = src[done + tid];
}

// Row-major group areas (REDUX_ROWS): row r of group g holds dword r of its 64 streams
// (slots + g * 64 * slot_bytes + 256 r + 4 l).  One workgroup gathers a tile of 64 rows: the
// rows are read whole (coalesced) into LDS, then every stream's 64 dwords of the tile leave as
// one 256-byte run of ALIGNED dwords of the dense output: output dword j of a stream that starts
// at byte offset sh (0..3) inside its first aligned dword is the byte-funnel of source dwords
// j-1 and j.  Only a stream's first and last output dword can be partial: those go bytewise.
constexpr uint32_t kTileRows = 64;
__global__ void __launch_bounds__(256) k_compact_rows(CompactArgs a)
{
    if (*a.mode == 0)
        return;
    __shared__ uint32_t tile[(kTileRows + 1) * 65]; // +1 leading row (source dword j-1); pitch 65: conflict-free column reads
    __shared__ uint64_t s_dst[64];                  // aligned dword that holds each stream's first byte (0: skip the stream)
    __shared__ uint32_t s_n[64], s_sh[64];
    __shared__ uint32_t s_maxj;
    const uint32_t tiles = (a.cap_rows + kTileRows - 1) / kTileRows + 1;
    const uint64_t g     = blockIdx.x / tiles;
    const uint32_t r0    = (blockIdx.x % tiles) * kTileRows;
    const uint32_t tid   = threadIdx.x;
    if (tid == 0)
        s_maxj = 0;
    __syncthreads();
    if (tid < 64) { // where does each stream go, and how many output dwords does the longest one need?
        const uint64_t b = g * 64 + tid;
        uint64_t       d = 0;
        uint32_t       n = 0, sh = 0;
        if (b < a.nblocks) {
            const uint64_t o0 = a.offsets[b], o1 = a.offsets[b + 1];
            if (o1 <= a.out_cap) {
                n  = (uint32_t)(o1 - o0);
                sh = (uint32_t)((uintptr_t)(a.out + o0) & 3);
                d  = (uint64_t)(uintptr_t)(a.out + o0) - sh;
                atomicMax(&s_maxj, (sh + n + 3) >> 2);
            } else if (r0 == 0) { // the dense buffer is too small for this block: report, never write
                if (a.status[b] == REDUX_OK)
                    a.status[b] = REDUX_OUTPUT_TOO_SMALL;
                if (a.summary) {
                    atomicCAS(&a.summary[0], REDUX_OK, REDUX_OUTPUT_TOO_SMALL);
                    atomicAdd(&a.summary[1], 1);
                }
            }
        }
        s_dst[tid] = d;
        s_n[tid]   = n;
        s_sh[tid]  = sh;
    }
    __syncthreads();
    if (r0 >= s_maxj)
        return;
    const uint4 *area = reinterpret_cast<const uint4 *>(a.slots + g * (64 * a.slot_bytes + 128));
    // tile row i (0..64) = source row r0 - 1 + i; a row is 16 uint4.  All loads first, then the LDS writes.
    uint4 v[4], lead = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t e = tid + 256 * q, r = r0 + (e >> 4);
        v[q] = r < a.cap_rows ? area[(uint64_t)r * 16 + (e & 15)] : make_uint4(0, 0, 0, 0);
    }
    if (tid < 16 && r0 > 0)
        lead = area[(uint64_t)(r0 - 1) * 16 + tid];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t e = tid + 256 * q;
        uint32_t      *w = &tile[((e >> 4) + 1) * 65 + (e & 15) * 4];
        w[0] = v[q].x; w[1] = v[q].y; w[2] = v[q].z; w[3] = v[q].w;
    }
    if (tid < 16) {
        uint32_t *w = &tile[tid * 4];
        w[0] = lead.x; w[1] = lead.y; w[2] = lead.z; w[3] = lead.w;
    }
    __syncthreads();
    // thread -> (stream, four consecutive output dwords): one 16-byte store where the whole quad is inside the stream
    const uint32_t wave = tid >> 6, t = tid & 63;
#pragma unroll 2
    for (uint32_t it = 0; it < 4; it++) {
        const uint32_t l  = wave * 16 + it * 4 + (t >> 4);
        const uint32_t jq = (t & 15) * 4; // tile-relative first output dword
        const uint64_t d  = s_dst[l];
        const uint32_t n = s_n[l], sh = s_sh[l];
        const uint32_t j0 = r0 + jq;
        if (d == 0 || j0 >= ((sh + n + 3) >> 2))
            continue;
        uint32_t src[5];
#pragma unroll
        for (int c = 0; c < 5; c++)
            src[c] = tile[(jq + c) * 65 + l]; // source dwords j0-1 .. j0+3
        uint32_t w[4];
#pragma unroll
        for (int c = 0; c < 4; c++)
            w[c] = sh ? __builtin_amdgcn_alignbyte(src[c + 1], src[c], 4 - sh) : src[c + 1];
        uint8_t      *A     = reinterpret_cast<uint8_t *>((uintptr_t)d) + 4 * (uint64_t)j0;
        const int64_t first = (int64_t)4 * j0 - sh; // stream index of the quad's byte 0
        if (first >= 0 && first + 16 <= (int64_t)n) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
            *reinterpret_cast<u32x4 *>(A) = u32x4{w[0], w[1], w[2], w[3]};
        } else { // a stream's head or tail: dwords where whole, bytes where not
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int64_t f = first + 4 * c;
                if (f >= 0 && f + 4 <= (int64_t)n) {
                    *reinterpret_cast<uint32_t *>(A + 4 * c) = w[c];
                } else {
                    for (int i = 0; i < 4; i++)
                        if (f + i >= 0 && f + i < (int64_t)n)
                            A[4 * c + i] = (uint8_t)(w[c] >> (8 * i));
                }
            }
        }
    }
}

// ======================================================================================
// synthetic workloads
// ======================================================================================
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__constant__ uint32_t c_zipf[256] = {
#include "zipf_table.inc"
};
static const uint32_t h_zipf[256] = {
#include "zipf_table.inc"
};

// byte j of the stream = byte (j mod 8) of splitmix64(seed + j/8); first_byte must be a
// multiple of 8 for the fast path, any value otherwise.
__global__ void k_gen_iid(uint8_t *out, uint64_t len, uint64_t first, uint64_t seed)
{
    const uint64_t nwords = (len + 7) / 8 + 1;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
         w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j0 = ((first >> 3) + w) << 3; // stream byte index of this word
        const uint64_t v  = splitmix64(seed + (j0 >> 3));
        if (j0 >= first && j0 + 8 <= first + len && (((uintptr_t)(out + (j0 - first))) & 7) == 0) {
            *reinterpret_cast<uint64_t *>(out + (j0 - first)) = v;
        } else {
            for (int k = 0; k < 8; k++) {
                const uint64_t j = j0 + k;
                if (j >= first && j < first + len)
                    out[j - first] = (uint8_t)(v >> (8 * k));
            }
        }
    }
}

__global__ void k_gen_zipf(uint8_t *out, uint64_t len, uint64_t first, uint64_t seed)
{
    const uint64_t ngroups = (len + 3) / 4;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t packed = 0;
        for (int k = 0; k < 4; k++) {
            const uint64_t j = g * 4 + k;
            const uint32_t u = (uint32_t)(splitmix64(seed + first + j) >> 32);
            // smallest r-1 with u <= thresholds[r-1]: 8-step binary search
            uint32_t lo = 0, hi = 255;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (u <= c_zipf[mid])
                    hi = mid;
                else
                    lo = mid + 1;
            }
            packed |= lo << (8 * k);
        }
        if (g * 4 + 4 <= len && (((uintptr_t)out) & 3) == 0) {
            reinterpret_cast<uint32_t *>(out)[g] = packed;
        } else {
            for (int k = 0; k < 4; k++)
                if (g * 4 + k < len)
                    out[g * 4 + k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

// ======================================================================================
// general parameters (redux_any.hpp): one lane per block, tree in the workspace
// ======================================================================================
struct AnyEncArgs {
    const uint8_t *in;
    uint64_t       in_len, nblocks;
    uint8_t       *slots;
    uint64_t       slot_bytes;
    uint32_t      *sizes;
    int32_t       *status;
    uint32_t      *trees;
    uint64_t       tree_words; // u32 entries per block
    uint32_t       block_size, slot_cap;
    uint32_t       sb, fb, cb;
};

__global__ void __launch_bounds__(64) k_encode_any(AnyEncArgs a)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.nblocks)
        return;
    const uint64_t o0  = blk * a.block_size;
    const uint64_t rem = a.in_len > o0 ? a.in_len - o0 : 0;
    const uint64_t len = rem < a.block_size ? rem : a.block_size;
    const any::Params P = any::make_params(a.sb, a.fb, a.cb);
    uint64_t  bi, bo;
    const int st = any::compress_stream(P, a.trees + blk * a.tree_words, a.in + o0, len, a.slots + blk * a.slot_bytes,
                                        a.slot_cap, bi, bo);
    a.sizes[blk]  = (uint32_t)bo;
    a.status[blk] = st;
}

struct AnyDecArgs {
    const uint8_t  *in;
    const uint64_t *in_offsets;
    uint64_t        nblocks;
    uint8_t        *out;
    uint32_t       *out_sizes;
    int32_t        *status;
    uint64_t       *in_used;
    uint32_t       *trees;
    uint64_t        tree_words;
    uint32_t        block_size;
    uint32_t        sb, fb, cb;
};

__global__ void __launch_bounds__(64) k_decode_any(AnyDecArgs a)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.nblocks)
        return;
    const uint64_t o0 = a.in_offsets[blk], o1 = a.in_offsets[blk + 1];
    const any::Params P = any::make_params(a.sb, a.fb, a.cb);
    uint64_t  bi, bo;
    const int st = any::decompress_stream(P, a.trees + blk * a.tree_words, a.in + o0, o1 - o0,
                                          a.out + blk * (uint64_t)a.block_size, a.block_size, bi, bo);
    a.out_sizes[blk] = (uint32_t)bo;
    a.status[blk]    = st;
    if (a.in_used)
        a.in_used[blk] = bi;
}

// ======================================================================================
// host side of the ABI
// ======================================================================================
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Geometry {
    uint64_t nblocks;
    uint64_t slot_bytes; // stride between slots (16-byte multiple, >= cap + 32)
    uint32_t slot_cap;   // usable bytes
    uint32_t rc_n;       // reciprocal table entries
    uint32_t nfreeze;
    bool     u16, fixup;
    bool     any;        // general-parameter path (redux_any.hpp): symbol_bits != 8 or code_bits > 32
    uint64_t tree_bytes; // any: per-block tree in the workspace
    // workspace layout (encode)
    uint64_t off_rc, off_sizes, off_mode, off_slots, off_trees, total;
};

static int check_params(const redux_params *p)
{
    if (!p)
        return REDUX_INVALID_INPUT;
    const int st = redux_params_check(p->symbol_bits, p->freq_bits, p->code_bits);
    if (st != REDUX_OK)
        return st;
    if (p->symbol_bits > 16) // a tree of 2^symbol_bits + 2 entries per block
        return REDUX_UNSUPPORTED;
    return REDUX_OK;
}

static bool is_any(const redux_params *p) { return p->symbol_bits != 8 || p->code_bits > 32; }

static uint64_t slot_cap_for(const redux_params *p, uint32_t block_size)
{
    const uint64_t freq_max = (1ull << p->freq_bits) - 1;
    if (is_any(p)) {
        // General parameters: a symbol of frequency >= 1 out of count <= min(freq_max, K + n)
        // costs at most ceil(log2 count) bits, + 1 for the truncation of codec.rs:59-60, + 1 spare.
        const uint64_t K = (1ull << p->symbol_bits) + 1;
        const uint64_t n = (uint64_t)block_size * 8 / p->symbol_bits + 1; // symbols incl. EOF
        uint32_t       lg = 0;
        while ((1ull << lg) < K + n)
            lg++;
        const uint64_t per = (lg < p->freq_bits ? lg : p->freq_bits) + 2;
        return n * per / 8 + p->code_bits / 8 + 64;
    }
    // Worst case of one block's stream.  While the model never freezes inside a block the
    // adaptive code length is <= 8 bits/symbol + O(256 log N) and the integer truncation of
    // codec.rs:59-60 loses < 1 bit/symbol: 9 bits/symbol.  Once frozen (count == freq_max) a
    // symbol of frequency 1 costs up to freq_bits + 1 bits.
    const uint64_t n        = block_size;
    const bool     freezes  = 257ull + n > freq_max;
    const uint64_t bits     = freezes ? n * (p->freq_bits + 2) : n * 9;
    return bits / 8 + 1024;
}

static Geometry geometry(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    Geometry g;
    memset(&g, 0, sizeof g);
    g.nblocks = in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
    const uint64_t cap = slot_cap_for(p, block_size);
    g.slot_cap   = cap > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)cap;
    // Slot stride: a whole number of 128-byte lines, and an ODD one.  All lanes write their
    // slots at about the same relative offset, so a stride that is a multiple of 2^k lines
    // folds the concurrently written lines onto 1/2^k of the L2 sets and evicts them
    // half-written (measured: 2.6x the stream bytes written to HBM at a stride of 584 lines).
    g.slot_bytes = align_up((uint64_t)g.slot_cap + 32, 128);
    if (((g.slot_bytes / 128) & 1) == 0)
        g.slot_bytes += 128;
    const uint64_t freq_max = (1ull << p->freq_bits) - 1;
    g.nfreeze = (uint32_t)(freq_max - 257);
    const uint64_t maxlen = in_len < block_size ? in_len : block_size;
    g.rc_n  = (uint32_t)((maxlen < g.nfreeze ? maxlen : g.nfreeze) + 1);
    g.u16   = block_size <= 65536;
    g.fixup = (257ull + (uint64_t)(g.rc_n - 1)) >= (1ull << 17);
    g.rc_n += 32; // slack: both coders load their reciprocals a group / a chunk ahead without clamping
    g.any = is_any(p);
    if (g.any) { // no reciprocal table; one tree of 2^symbol_bits + 2 u32 per block
        g.rc_n       = 0;
        g.u16        = false;
        g.fixup      = true;
        g.tree_bytes = align_up(((1ull << p->symbol_bits) + 2) * 4, 256);
    }
    g.off_rc    = 0;
    g.off_sizes = align_up(g.off_rc + (uint64_t)g.rc_n * 8, 256);
    g.off_mode  = align_up(g.off_sizes + g.nblocks * 4, 256); // one word: 0 linear slots, != 0 row-major group areas
    g.off_slots = g.off_mode + 256 + kClaimWords * 4; // mode word, then k_encode_pair's role book
    // whole groups of 64 slots (a row-major group area is 64 slots big) + 1 spare slot for the dead lanes of linear mode
    g.off_trees = align_up(g.off_slots + ((g.nblocks + 63) / 64 * 64 + 1) * g.slot_bytes + (g.nblocks + 63) / 64 * 128, 256);
    g.total     = g.off_trees + g.nblocks * g.tree_bytes;
    return g;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "redux_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),       \
                    __FILE__, __LINE__);                                                               \
            return REDUX_IO_ERROR;                                                                     \
        }                                                                                              \
    } while (0)

} // namespace redux

using namespace redux;

extern "C" {

const char *redux_version(void) { return "redux_hip 0.1.0 gfx950"; }

int redux_params_check(uint32_t symbol, uint32_t frequency, uint32_t code) /* model/mod.rs:64 */
{
    if (symbol < 1 || frequency < symbol + 2 || code < frequency + 2 || 64 < code + frequency)
        return REDUX_INVALID_INPUT;
    return REDUX_OK;
}

int redux_device_supports(const redux_params *p) { return check_params(p); }

uint64_t redux_block_count(uint64_t in_len, uint32_t block_size)
{
    if (block_size == 0)
        return 0;
    return in_len == 0 ? 1 : (in_len + block_size - 1) / block_size;
}

uint64_t redux_encode_slot_bytes(const redux_params *p, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    return geometry(p, block_size, block_size).slot_cap;
}

uint64_t redux_encode_bound(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    const Geometry g = geometry(p, in_len, block_size);
    return g.nblocks * (uint64_t)g.slot_cap;
}

uint64_t redux_encode_workspace_bytes(const redux_params *p, uint64_t in_len, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    return geometry(p, in_len, block_size).total;
}

int redux_encode_slots_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                           void *d_block_status, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_workspace || !d_block_status || (in_len && !d_in))
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry(p, in_len, block_size);
    if (workspace_bytes < g.total)
        return REDUX_OUTPUT_TOO_SMALL;
    if (((uintptr_t)d_workspace) & 255)
        return REDUX_INVALID_INPUT;
    hipStream_t s  = (hipStream_t)stream;
    uint8_t    *ws = (uint8_t *)d_workspace;

    HIP_TRY(hipMemsetAsync(ws + g.off_mode, 0, 256 + kClaimWords * 4, s)); // linear slots unless the pair kernel runs (below); empty role book
    if (g.any) {
        AnyEncArgs aa;
        aa.in         = (const uint8_t *)d_in;
        aa.in_len     = in_len;
        aa.nblocks    = g.nblocks;
        aa.slots      = ws + g.off_slots;
        aa.slot_bytes = g.slot_bytes;
        aa.sizes      = (uint32_t *)(ws + g.off_sizes);
        aa.status     = (int32_t *)d_block_status;
        aa.trees      = (uint32_t *)(ws + g.off_trees);
        aa.tree_words = g.tree_bytes / 4;
        aa.block_size = block_size;
        aa.slot_cap   = g.slot_cap;
        aa.sb = p->symbol_bits; aa.fb = p->freq_bits; aa.cb = p->code_bits;
        k_encode_any<<<(uint32_t)((g.nblocks + 63) / 64), 64, 0, s>>>(aa);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }

    k_fill_rc<<<(g.rc_n + 255) / 256, 256, 0, s>>>((double *)(ws + g.off_rc), g.rc_n);

    EncArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_len     = in_len;
    a.nblocks    = g.nblocks;
    a.slots      = ws + g.off_slots;
    a.slot_bytes = g.slot_bytes;
    a.sizes      = (uint32_t *)(ws + g.off_sizes);
    a.status     = (int32_t *)d_block_status;
    a.rc         = (const double *)(ws + g.off_rc);
    a.block_size = block_size;
    a.slot_cap   = g.slot_cap;
    a.nfreeze    = g.nfreeze;
    a.code_bits  = p->code_bits;
    a.aligned16  = ((((uintptr_t)d_in) & 15) == 0 && (block_size & 15) == 0) ? 1 : 0;
    a.claims     = (uint32_t *)(ws + g.off_mode + 256);
    // 64 blocks per wave while 64 slots / 64 blocks stay within a 32-bit lane offset;
    // otherwise (giant blocks, whole-stream mode) one block per wave.
    a.lanes = (64ull * g.slot_bytes < (1ull << 32) && 64ull * block_size < (1ull << 32)) ? 64u : 1u;
    const uint32_t grid = (uint32_t)((g.nblocks + a.lanes - 1) / a.lanes);
    // REDUX_ENCODE_KERNEL=single pins the one-wave kernel (A/B timing only)
    const char *force = getenv("REDUX_ENCODE_KERNEL");
    const bool  pair  = g.u16 && a.aligned16 && a.lanes == 64 && !(force && !strcmp(force, "single"));
    // (a u16 tree means blocks of <= 65536 symbols, so count < 2^17: the pair kernel never needs FIXUP)
#if REDUX_ROWS
    if (pair && !g.fixup)
        HIP_TRY(hipMemsetAsync(ws + g.off_mode, 1, 4, s));
#endif
    if (pair && !g.fixup && p->code_bits == 32)
        k_encode_pair<false, true><<<grid, 128, 0, s>>>(a);
    else if (pair && !g.fixup)
        k_encode_pair<false, false><<<grid, 128, 0, s>>>(a);
    else if (g.u16 && !g.fixup)
        k_encode<true, false><<<grid, 64, 0, s>>>(a);
    else if (g.u16)
        k_encode<true, true><<<grid, 64, 0, s>>>(a);
    else
        k_encode<false, true><<<grid, 64, 0, s>>>(a);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_compact_slots_dev(const redux_params *p, uint64_t in_len, uint32_t block_size, void *d_out,
                            uint64_t out_cap, void *d_out_offsets, void *d_block_status, void *d_summary,
                            void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_workspace || !d_block_status || !d_out_offsets || !d_out)
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry(p, in_len, block_size);
    if (workspace_bytes < g.total)
        return REDUX_OUTPUT_TOO_SMALL;
    hipStream_t s  = (hipStream_t)stream;
    uint8_t    *ws = (uint8_t *)d_workspace;

    ScanArgs sa;
    sa.sizes   = (const uint32_t *)(ws + g.off_sizes);
    sa.status  = (const int32_t *)d_block_status;
    sa.offsets = (uint64_t *)d_out_offsets;
    sa.summary = (int32_t *)d_summary;
    sa.nblocks = g.nblocks;
    k_scan_sizes<<<1, 1024, 0, s>>>(sa);

    CompactArgs ca;
    ca.slots      = ws + g.off_slots;
    ca.slot_bytes = g.slot_bytes;
    ca.offsets    = (const uint64_t *)d_out_offsets;
    ca.out        = (uint8_t *)d_out;
    ca.out_cap    = out_cap;
    ca.status     = (int32_t *)d_block_status;
    ca.summary    = (int32_t *)d_summary;
    ca.nblocks    = g.nblocks;
    ca.mode       = (const uint32_t *)(ws + g.off_mode);
    ca.cap_rows   = (uint32_t)(g.slot_bytes / 4);
    k_compact<<<(uint32_t)g.nblocks, 256, 0, s>>>(ca);
#if REDUX_ROWS // the mode word decides on the device which of the two does the work
    if (!g.any && g.u16) {
        const uint32_t tiles = (ca.cap_rows + kTileRows - 1) / kTileRows + 1;
        k_compact_rows<<<(uint32_t)((g.nblocks + 63) / 64) * tiles, 256, 0, s>>>(ca);
    }
#endif
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_encode_blocks_dev(const redux_params *p, const void *d_in, uint64_t in_len, uint32_t block_size,
                            void *d_out, uint64_t out_cap, void *d_out_offsets, void *d_block_status,
                            void *d_summary, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    int st = redux_encode_slots_dev(p, d_in, in_len, block_size, d_block_status, d_workspace, workspace_bytes, stream);
    if (st != REDUX_OK)
        return st;
    return redux_compact_slots_dev(p, in_len, block_size, d_out, out_cap, d_out_offsets, d_block_status, d_summary,
                                   d_workspace, workspace_bytes, stream);
}

int redux_encode_blocks(const redux_params *p, const uint8_t *in, uint64_t in_len, uint32_t block_size,
                        uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, int32_t *block_status)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !out || !out_offsets || (in_len && !in))
        return REDUX_INVALID_INPUT;
    const Geometry g = geometry(p, in_len, block_size);
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    uint64_t *d_off = nullptr;
    int32_t  *d_st = nullptr, *d_sum = nullptr;
    int       rc = REDUX_OK;
    int32_t   summary[2] = {0, 0};
#define TRY_GOTO(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "redux_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));              \
            rc = REDUX_IO_ERROR;                                                                       \
            goto done;                                                                                 \
        }                                                                                              \
    } while (0)
    TRY_GOTO(hipMalloc((void **)&d_in, in_len ? in_len : 16));
    TRY_GOTO(hipMalloc((void **)&d_out, out_cap ? out_cap : 16));
    TRY_GOTO(hipMalloc((void **)&d_ws, g.total));
    TRY_GOTO(hipMalloc((void **)&d_off, (g.nblocks + 1) * 8));
    TRY_GOTO(hipMalloc((void **)&d_st, g.nblocks * 4));
    TRY_GOTO(hipMalloc((void **)&d_sum, 8));
    if (in_len)
        TRY_GOTO(hipMemcpy(d_in, in, in_len, hipMemcpyHostToDevice));
    TRY_GOTO(hipMemset(d_sum, 0, 8));
    rc = redux_encode_blocks_dev(p, d_in, in_len, block_size, d_out, out_cap, d_off, d_st, d_sum, d_ws, g.total, nullptr);
    if (rc != REDUX_OK)
        goto done;
    TRY_GOTO(hipDeviceSynchronize());
    TRY_GOTO(hipMemcpy(out_offsets, d_off, (g.nblocks + 1) * 8, hipMemcpyDeviceToHost));
    TRY_GOTO(hipMemcpy(summary, d_sum, 8, hipMemcpyDeviceToHost));
    if (block_status)
        TRY_GOTO(hipMemcpy(block_status, d_st, g.nblocks * 4, hipMemcpyDeviceToHost));
    if (out_offsets[g.nblocks] <= out_cap)
        TRY_GOTO(hipMemcpy(out, d_out, out_offsets[g.nblocks], hipMemcpyDeviceToHost));
    rc = summary[0];
done:
    hipFree(d_in); hipFree(d_out); hipFree(d_ws); hipFree(d_off); hipFree(d_st); hipFree(d_sum);
    return rc;
#undef TRY_GOTO
}

int redux_compress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                   uint64_t *bytes_in, uint64_t *bytes_out) /* src/lib.rs:102-109 */
{
    if (in_len > 0xFFFFFF00ull)
        return REDUX_UNSUPPORTED;
    uint64_t  offs[2] = {0, 0};
    int32_t   st      = 0;
    const int rc = redux_encode_blocks(p, in, in_len, in_len ? (uint32_t)in_len : 1u, out, out_cap, offs, &st);
    if (rc == REDUX_OK) {
        if (bytes_in)
            *bytes_in = in_len;
        if (bytes_out)
            *bytes_out = offs[1];
    }
    return rc;
}

uint64_t redux_decode_workspace_bytes(const redux_params *p, uint64_t nblocks, uint32_t block_size)
{
    if (check_params(p) != REDUX_OK || block_size == 0)
        return 0;
    const Geometry g = geometry(p, block_size, block_size);
    if (g.any)
        return (nblocks ? nblocks : 1) * g.tree_bytes;
    return align_up((uint64_t)g.rc_n * 8, 256);
}

// d_in_used (optional, u64[nblocks]): bytes of each stream the reader fetched; only
// redux_decompress asks for it (the (u64, u64) of src/lib.rs:119)
static int decode_blocks_dev_impl(const redux_params *p, const void *d_in, const void *d_in_offsets, uint64_t nblocks,
                                  uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_sizes,
                                  void *d_block_status, void *d_summary, void *d_workspace,
                                  uint64_t workspace_bytes, void *stream, void *d_in_used)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !d_in_offsets || !d_out_sizes || !d_block_status || !d_workspace)
        return REDUX_INVALID_INPUT;
    if (nblocks == 0)
        return REDUX_OK;
    if (out_cap < nblocks * (uint64_t)block_size)
        return REDUX_OUTPUT_TOO_SMALL;
    const Geometry g = geometry(p, block_size, block_size);
    if (workspace_bytes < redux_decode_workspace_bytes(p, nblocks, block_size))
        return REDUX_OUTPUT_TOO_SMALL;
    hipStream_t s = (hipStream_t)stream;
    if (g.any) {
        AnyDecArgs aa;
        aa.in         = (const uint8_t *)d_in;
        aa.in_offsets = (const uint64_t *)d_in_offsets;
        aa.nblocks    = nblocks;
        aa.out        = (uint8_t *)d_out;
        aa.out_sizes  = (uint32_t *)d_out_sizes;
        aa.status     = (int32_t *)d_block_status;
        aa.in_used    = (uint64_t *)d_in_used;
        aa.trees      = (uint32_t *)d_workspace;
        aa.tree_words = g.tree_bytes / 4;
        aa.block_size = block_size;
        aa.sb = p->symbol_bits; aa.fb = p->freq_bits; aa.cb = p->code_bits;
        k_decode_any<<<(uint32_t)((nblocks + 63) / 64), 64, 0, s>>>(aa);
        if (d_summary)
            k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
        HIP_TRY(hipGetLastError());
        return REDUX_OK;
    }
    k_fill_rc<<<(g.rc_n + 255) / 256, 256, 0, s>>>((double *)d_workspace, g.rc_n);
    DecArgs a;
    a.in         = (const uint8_t *)d_in;
    a.in_offsets = (const uint64_t *)d_in_offsets;
    a.nblocks    = nblocks;
    a.out        = (uint8_t *)d_out;
    a.out_sizes  = (uint32_t *)d_out_sizes;
    a.status     = (int32_t *)d_block_status;
    a.rc         = (const double *)d_workspace;
    a.block_size = block_size;
    a.nfreeze    = g.nfreeze;
    a.code_bits  = p->code_bits;
    a.aligned4   = ((((uintptr_t)d_out) & 3) == 0 && (block_size & 3) == 0) ? 1 : 0;
    if (a.aligned4 && (((uintptr_t)d_out) & 15) == 0 && (block_size & 15) == 0)
        a.aligned4 = 2; // 16-byte aligned blocks: the lock-step decoder stages four dwords per store
    a.in_used    = (uint64_t *)d_in_used;
    const uint32_t grid = (uint32_t)((nblocks + 63) / 64);
    const char *force = getenv("REDUX_DECODE_KERNEL"); // "generic" pins k_decode (A/B timing only)
    if (g.u16 && !g.fixup && !force && p->code_bits == 32)
        k_decode_lock<true><<<grid, 64, 0, s>>>(a);
    else if (g.u16 && !g.fixup && !force)
        k_decode_lock<false><<<grid, 64, 0, s>>>(a);
    else if (g.u16 && !g.fixup)
        k_decode<true, false><<<grid, 64, 0, s>>>(a);
    else if (g.u16)
        k_decode<true, true><<<grid, 64, 0, s>>>(a);
    else
        k_decode<false, true><<<grid, 64, 0, s>>>(a);
    if (d_summary)
        k_summarize<<<64, 256, 0, s>>>((const int32_t *)d_block_status, nblocks, (int32_t *)d_summary);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

#ifdef REDUX_DEC_CENSUS
extern "C" int redux_debug_dec_census(uint32_t *out4096)
{
    return (int)hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_dec_hw), 4096 * 4);
}
#endif
#ifdef REDUX_DEC_STAMPS
extern "C" int redux_debug_dec_stamps(uint64_t *out8)
{
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dec_ts), 64);
}
#endif

// Diagnostic (tests only): max |v_rcp_f64(x) * x - 1| over the integers lo..hi, on the device.
int redux_debug_rcp_check(uint64_t lo, uint64_t hi, double *max_err)
{
    unsigned long long *d = nullptr, h = 0;
    HIP_TRY(hipMalloc(&d, 8));
    HIP_TRY(hipMemset(d, 0, 8));
    k_rcp_check<<<4096, 256>>>(lo, hi, d);
    HIP_TRY(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    memcpy(max_err, &h, 8);
    return REDUX_OK;
}

int redux_decode_blocks_dev(const redux_params *p, const void *d_in, const void *d_in_offsets, uint64_t nblocks,
                            uint32_t block_size, void *d_out, uint64_t out_cap, void *d_out_sizes,
                            void *d_block_status, void *d_summary, void *d_workspace, uint64_t workspace_bytes,
                            void *stream)
{
    return decode_blocks_dev_impl(p, d_in, d_in_offsets, nblocks, block_size, d_out, out_cap, d_out_sizes,
                                  d_block_status, d_summary, d_workspace, workspace_bytes, stream, nullptr);
}

static int decode_blocks_host(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                              uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes,
                              int32_t *block_status, uint64_t *in_used)
{
    int st = check_params(p);
    if (st != REDUX_OK)
        return st;
    if (block_size == 0 || !in_offsets || !out_sizes || (nblocks && !out))
        return REDUX_INVALID_INPUT;
    if (nblocks == 0)
        return REDUX_OK;
    if (out_cap < nblocks * (uint64_t)block_size)
        return REDUX_OUTPUT_TOO_SMALL;
    const uint64_t in_len = in_offsets[nblocks];
    const uint64_t wsb    = redux_decode_workspace_bytes(p, nblocks, block_size);
    uint8_t  *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_sz = nullptr;
    uint64_t *d_used = nullptr;
    int32_t  *d_st = nullptr, *d_sum = nullptr;
    int       rc = REDUX_OK;
    int32_t   summary[2] = {0, 0};
#define TRY_GOTO(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "redux_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));              \
            rc = REDUX_IO_ERROR;                                                                       \
            goto done;                                                                                 \
        }                                                                                              \
    } while (0)
    TRY_GOTO(hipMalloc((void **)&d_in, in_len + 16));
    TRY_GOTO(hipMalloc((void **)&d_out, nblocks * (uint64_t)block_size));
    TRY_GOTO(hipMalloc((void **)&d_ws, wsb));
    TRY_GOTO(hipMalloc((void **)&d_off, (nblocks + 1) * 8));
    TRY_GOTO(hipMalloc((void **)&d_sz, nblocks * 4));
    TRY_GOTO(hipMalloc((void **)&d_st, nblocks * 4));
    TRY_GOTO(hipMalloc((void **)&d_sum, 8));
    if (in_used)
        TRY_GOTO(hipMalloc((void **)&d_used, nblocks * 8));
    if (in_len)
        TRY_GOTO(hipMemcpy(d_in, in, in_len, hipMemcpyHostToDevice));
    TRY_GOTO(hipMemcpy(d_off, in_offsets, (nblocks + 1) * 8, hipMemcpyHostToDevice));
    TRY_GOTO(hipMemset(d_sum, 0, 8));
    rc = decode_blocks_dev_impl(p, d_in, d_off, nblocks, block_size, d_out, nblocks * (uint64_t)block_size, d_sz, d_st,
                                d_sum, d_ws, wsb, nullptr, d_used);
    if (rc != REDUX_OK)
        goto done;
    TRY_GOTO(hipDeviceSynchronize());
    TRY_GOTO(hipMemcpy(out_sizes, d_sz, nblocks * 4, hipMemcpyDeviceToHost));
    if (in_used)
        TRY_GOTO(hipMemcpy(in_used, d_used, nblocks * 8, hipMemcpyDeviceToHost));
    TRY_GOTO(hipMemcpy(summary, d_sum, 8, hipMemcpyDeviceToHost));
    if (block_status)
        TRY_GOTO(hipMemcpy(block_status, d_st, nblocks * 4, hipMemcpyDeviceToHost));
    TRY_GOTO(hipMemcpy(out, d_out, nblocks * (uint64_t)block_size, hipMemcpyDeviceToHost));
    rc = summary[0];
done:
    hipFree(d_in); hipFree(d_out); hipFree(d_ws); hipFree(d_off); hipFree(d_sz); hipFree(d_st); hipFree(d_sum); hipFree(d_used);
    return rc;
#undef TRY_GOTO
}

int redux_decode_blocks(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                        uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes,
                        int32_t *block_status)
{
    return decode_blocks_host(p, in, in_offsets, nblocks, block_size, out, out_cap, out_sizes, block_status, nullptr);
}

int redux_decompress(const redux_params *p, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                     uint64_t *bytes_in, uint64_t *bytes_out) /* src/lib.rs:113-120 */
{
    if (out_cap == 0 || out_cap > 0xFFFFFF00ull)
        out_cap = out_cap ? 0xFFFFFF00ull : 1;
    uint64_t  offs[2] = {0, in_len};
    uint32_t  sz      = 0;
    int32_t   st      = 0;
    uint64_t  used    = 0;
    const int rc = decode_blocks_host(p, in, offs, 1, (uint32_t)out_cap, out, out_cap, &sz, &st, &used);
    if (rc == REDUX_OK) {
        if (bytes_out)
            *bytes_out = sz;
        // input.get_count() (lib.rs:119): the bytes the BitReader fetched -- the decoder reads
        // exactly the bits the encoder wrote, so trailing bytes after the stream are not counted
        if (bytes_in)
            *bytes_in = used;
    }
    return rc;
}

int redux_gen_iid_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream)
{
    if (len == 0)
        return REDUX_OK;
    k_gen_iid<<<2048, 256, 0, (hipStream_t)stream>>>((uint8_t *)d_out, len, first_byte, seed);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

int redux_gen_zipf_dev(void *d_out, uint64_t len, uint64_t first_byte, uint64_t seed, void *stream)
{
    if (len == 0)
        return REDUX_OK;
    k_gen_zipf<<<2048, 256, 0, (hipStream_t)stream>>>((uint8_t *)d_out, len, first_byte, seed);
    HIP_TRY(hipGetLastError());
    return REDUX_OK;
}

const uint32_t *redux_zipf_thresholds(void) { return h_zipf; }

} // extern "C"
