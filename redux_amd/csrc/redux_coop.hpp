// redux_hip -- small launches: the model of a block computed by 64 lanes, its interval chain by one.
//
//   k_coop_model<WINB>         one WAVE per block: lane j runs AdaptiveTreeModel over the j-th 64th of the block (of the
//                              block's current window: WINB), starting from the counts of everything before it, and leaves
//                              every symbol's (low, high) in the workspace
//   k_coop_chain<CB32, FIXUP, LINEAR, WIN>
//                              one LANE per block, 64 blocks per workgroup of TWO waves: the interval half of
//                              compress_symbol (chain wave) and its bit-writer half (emit wave) over those pairs
//
// Why: with 64 blocks per wave a launch of n blocks keeps n / 64 SIMDs busy; below ~4096 blocks most of the chip idles
// while every block still pays the full serial price, 65,536 symbols x (model wave's issue time) = 10 ms.  But what
// get_frequency_range(s) returns for symbol i (adaptive_tree.rs:63-92) depends only on the COUNTS of symbols 0 .. i-1,
// not on the coder's state: counts are prefix statistics of the input and parallelise exactly.  Lane j histograms its
// segment, an exclusive scan across the lanes gives every segment its starting counts, each lane builds the Fenwick form
// of those and then runs the ordinary query + update over its 1/64th.  Only codec.rs:55-60's chain (low, high depend on
// the previous symbol's) stays serial: 27 instructions per symbol on a wave that does nothing else (4.9 ms per 64 KiB
// block against 9.6 ms; one block of any length -- redux_compress, the literal redux::compress -- 3 -> 9.5-13 MB/s).
//
// Blocks of up to 64 KiB (launches of at most kCoopMaxBlocks slots whose pairs fit kCoopMaxPairBytes): whole blocks, the
// pairs symbol-major per group of 64 blocks, pairs[(group * (block_size + 1 + slack) + i) * 64 + lane]: the chain wave reads
// one contiguous row per symbol.  Blocks ABOVE 64 KiB (launches of at most kCoopMaxLargeBlocks): window by window -- one
// k_fill_rc_from + k_coop_model<true> + k_coop_chain<.., WIN> per window of at most kCoopWindowMax symbols of every block
// (EncArgs::win0, winlen; the loop is in redux_hip.hip) --, the pairs block-major, pairs[block * pitch + i - win0]; coder
// state and symbol counts of a block travel in the workspace (cstate, cbase), so the pairs area and the reciprocal table
// hold one window whatever the block length.  A model that freezes inside a block included; results are the same bytes as
// every other kernel's (tests/test_gpu_parity.py: the corpus, batch, whole-stream, hand-made-table and windows tests).
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"
#include "redux_encode.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace redux {

constexpr uint64_t kCoopMaxBlocks = 2048; // slots, idle table entries included (the pairs take 8 bytes per input byte of workspace: 1 GiB at 2048 x 64 KiB)
constexpr uint64_t kCoopMaxPairBytes = 3ull << 29; // ... and never more than 1.5 GiB (blocks of up to 64 KiB: whole blocks of pairs)
constexpr uint32_t kCoopMinBlock  = 1024; // shorter blocks: the per-block set-up (scan, tree build) outweighs the model
constexpr uint32_t kCoopSlack     = 64;   // symbols of slack behind a group's pairs (the chain wave prefetches unclamped)
// Blocks above 64 KiB are coded in windows (EncArgs::win0): the pairs of one window of all blocks take at most this much ...
#ifndef REDUX_COOP_WINDOW_MIB // (A/B builds set these)
#define REDUX_COOP_WINDOW_MIB 2816
#endif
#ifndef REDUX_COOP_MAX_LARGE_BLOCKS
#define REDUX_COOP_MAX_LARGE_BLOCKS 24576
#endif
constexpr uint64_t kCoopWindowBytes = (uint64_t)REDUX_COOP_WINDOW_MIB << 20;
constexpr uint64_t kCoopMaxLargeBlocks = REDUX_COOP_MAX_LARGE_BLOCKS; // slots of a launch of blocks above 64 KiB
constexpr uint64_t kCoopWindowMax   = 65504;      // ... and a window is at most this many symbols: what a u16 tree node counts (k_coop_model)

// pairs of a block coded in windows: one window + slack, an even number of pairs (16-byte loads)
__host__ __device__ __forceinline__ uint64_t coop_block_pitch(uint32_t winlen) { return ((uint64_t)winlen + kCoopSlack + 1) & ~1ull; }

// inclusive sum over the wave's lanes 0 .. lane.  DPP: Hillis-Steele inside each row of 16 lanes (row_shr:1, 2, 4, 8; a lane
// whose source falls outside its row adds 0), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31 into
// rows 2 and 3 (row_bcast:31): six VALU additions.  (__shfl_up is a ds_bpermute -- an LDS round trip -- per step: six dependent
// ones per scan, 255 scans per block and window, on a wave that has nothing else to run: ~60 of the ~100 us a block-window
// took, which is what bounded launches of many large blocks once they were coded in windows.)
__device__ __forceinline__ uint32_t coop_wave_scan(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// WINB: the block is coded in windows (blocks above 64 KiB).  The lanes' trees still hold u16 nodes -- they count the symbols
// of THIS window only, at most kCoopWindowMax of them --, and what the windows before it counted comes from a table of the
// block in LDS: bcum[s] = the symbols < s among them (cbase: the counts themselves, carried in the workspace).  32 KiB + 1 KiB
// per block: four blocks per CU in flight, as for blocks of up to 64 KiB (u32 nodes: two).
constexpr uint32_t kCoopModelDwords = Tree<true>::kDwords + 260; // the lanes' trees + bcum
// (the body of the kernel: k_coop_step runs it next to the chain of the window before; ONE wave, lds = kCoopModelDwords)
template <bool WINB>
__device__ __forceinline__ void coop_model_body(const EncArgs &a, uint2 *pairs, const uint64_t ent, uint32_t *lds)
{
    constexpr bool U16 = true;
    typedef Tree<U16> TreeT;
    uint32_t *const bcum = lds + TreeT::kDwords; // WINB: 257 entries
    const uint32_t lane = threadIdx.x;
    // ent: the slot -- lane ent & 63 of chain wave ent >> 6
    const uint8_t *src;
    uint32_t       len;
    if (a.table) {
        const redux_block e = a.table[ent];
        if (e.index == kIdleEntry)
            return;
        src = a.in + e.offset;
        len = e.length;
    } else {
        if (ent >= a.nblocks)
            return;
        const uint64_t rem = a.in_len - ent * a.block_size;
        src = a.in + ent * a.block_size;
        len = rem < a.block_size ? (uint32_t)rem : a.block_size;
    }
    // this launch's window of the block: symbols [w0, w0 + lenw)
    const uint32_t w0 = WINB ? a.win0 : 0u;
    if (len <= w0)
        return;
    const uint32_t lenw = len - w0 < a.winlen ? len - w0 : a.winlen;
    for (uint32_t i = lane; i < TreeT::kDwords / 4; i += 64)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
    TreeT T;
    T.init(lds, lane);
    __syncthreads();

    // this lane's symbols [b0, b1), read as the aligned 16-byte pieces of memory that contain them (the bytes of a piece
    // outside the block are in the same page as bytes inside it)
    const uint32_t  seg = (lenw + 63) / 64;
    const uint32_t  r0 = lane * seg < lenw ? lane * seg : lenw, r1 = r0 + seg < lenw ? r0 + seg : lenw;
    const uint32_t  b0 = w0 + r0, b1 = w0 + r1;
    const uintptr_t A0 = (uintptr_t)src + b0, A1 = (uintptr_t)src + b1, C0 = A0 & ~(uintptr_t)15;
    const uint32_t  npieces = b1 > b0 ? (uint32_t)((A1 - C0 + 15) >> 4) : 0u;
    const uint32_t  maxpieces = __builtin_amdgcn_readfirstlane(wave_max(npieces));
    auto piece = [&](uint32_t k) { return *reinterpret_cast<const uint4 *>(C0 + 16 * (uintptr_t)(k < npieces ? k : 0)); };

    // ---- 1. counts of this segment: row s + 1 of the lane's column (row 0 takes symbol 255, whose count no prefix needs)
    {
        uint4 nx = npieces ? piece(0) : make_uint4(0, 0, 0, 0);
        for (uint32_t k = 0; k < maxpieces; k++) {
            const uint4 cur = nx;
            if (k + 1 < npieces)
                nx = piece(k + 1);
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uintptr_t ad = C0 + 16 * (uintptr_t)k + i;
                const uint32_t  s  = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                if (k < npieces && ad >= A0 && ad < A1 && (uint32_t)(ad - (uintptr_t)src) < a.nfreeze) // (the model stops counting at the freeze, adaptive_tree.rs:84)
                    T.add((((s + 1u) & 255u) << TreeT::kShift) | T.L, T.inc);
            }
        }
    }
    __syncthreads();
    // ---- 2. row r: counts of symbol r - 1 in the segments BEFORE this lane's (exclusive scan over the lanes) -- and, when
    //         the block is coded in windows, in the windows before this one: cbase row r of the block, lane l holding
    //         rows 4l .. 4l + 3 of it while the scan runs, updated with this window's totals for the next one
    uint4 *basep = WINB ? reinterpret_cast<uint4 *>(a.cbase + ent * 256) + lane : nullptr;
    uint4  bq    = (WINB && w0) ? *basep : make_uint4(0, 0, 0, 0);
    if (WINB) { // bcum[r] = rows 1 .. r of the base counts (row r = symbol r - 1), r = 0 .. 255; bcum[256] = 0: cum(256) is derived
        const uint32_t p0 = lane ? bq.x : 0u, p1 = p0 + bq.y, p2 = p1 + bq.z, p3 = p2 + bq.w;
        const uint32_t ex = coop_wave_scan(p3, lane) - p3;
        reinterpret_cast<uint4 *>(bcum)[lane] = make_uint4(ex + p0, ex + p1, ex + p2, ex + p3);
        if (lane == 0)
            bcum[256] = 0;
    }
    for (uint32_t r4 = 0; r4 < 64; r4++) { // four rows per turn: their LDS reads, scans and writes overlap (a lone wave hides nothing by itself)
        const uint32_t bw[4] = {bq.x, bq.y, bq.z, bq.w};
        uint32_t       nw[4] = {bq.x, bq.y, bq.z, bq.w};
        uint32_t       v[4], incl[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            v[k] = T.node(((4 * r4 + k) << TreeT::kShift) | T.L); // (row 0 is read and scanned too; nothing uses it)
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            incl[k] = coop_wave_scan(v[k], lane);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t r = 4 * r4 + k;
            if (r == 0)
                continue;
            // (r4 is wave-uniform: v_readlane, not a cross-lane LDS operation)
            char *cell = reinterpret_cast<char *>(lds) + ((r << TreeT::kShift) | T.L);
            *reinterpret_cast<uint16_t *>(cell + 2 * (lane >> 5)) = (uint16_t)(incl[k] - v[k]);
            if (WINB) { // this window's count of the symbol joins the base counts for the next window
                const uint32_t br  = (uint32_t)__builtin_amdgcn_readlane((int)bw[k], (int)r4);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl[k], 63);
                nw[k] = lane == r4 ? br + tot : nw[k];
            }
        }
        bq = make_uint4(nw[0], nw[1], nw[2], nw[3]);
    }
    if (WINB)
        *basep = bq;
    __syncthreads();
    // ---- 3. the Fenwick form in place (node i also covers node i - lowbit(i) + ... : adaptive_tree.rs:43-59): d[i] is
    //         the number of increments node i has received after the symbols before this segment
    //         Level by level -- the nodes (2k + 1) << b are complete once the levels below b have been pushed up, and
    //         independent of each other --, eight nodes in flight at a time (in index order every read waits for the atomic
    //         before it: 255 dependent LDS round trips)
#pragma unroll
    for (int b = 0; b < 7; b++) {
        const uint32_t n = 128u >> b; // nodes of this level
        for (uint32_t k0 = 0; k0 < n; k0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                const uint32_t k = k0 + q, i = (2 * k + 1) << b;
                v[q] = k < n ? T.node((i << TreeT::kShift) | T.L) : 0u;
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                const uint32_t k = k0 + q, j = (2 * k + 2) << b;
                if (k < n && j < 256)
                    T.add((j << TreeT::kShift) | T.L, v[q] << T.hsh);
            }
        }
    }
    // ---- 4. query + update over the segment (adaptive_tree.rs:63-92), pairs out
    // blocks of up to 64 KiB: symbol-major rows of 64 lanes per group; blocks coded in windows: block-major, a
    // block's window contiguous -- a lane then completes the 128-byte lines it writes by itself (sixteen consecutive stores)
    // instead of sharing each line with fifteen other workgroups on other XCDs, whose L2s each wrote their 8 bytes of it back
    uint2         *pg = WINB ? pairs + ent * coop_block_pitch(a.winlen) : pairs + ((ent >> 6) * ((uint64_t)a.winlen + kCoopSlack)) * 64 + (ent & 63);
    const uint32_t pw = WINB ? 1u : 64u;
    {
        uint4 nx = npieces ? piece(0) : make_uint4(0, 0, 0, 0);
        for (uint32_t k = 0; k < maxpieces; k++) {
            const uint4 cur = nx;
            if (k + 1 < npieces)
                nx = piece(k + 1);
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll 4
            for (int i = 0; i < 16; i++) {
                const uintptr_t ad = C0 + 16 * (uintptr_t)k + i;
                const uint32_t  s  = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                if (k < npieces && ad >= A0 && ad < A1) {
                    const uint32_t q   = (uint32_t)(ad - (uintptr_t)src); // symbols before this one
                    const uint32_t nup = q < a.nfreeze ? q : a.nfreeze;     // updates before this one
                    uint32_t       lo, hi;
                    // (the update of a block's last symbol is unobservable and skipped: u16 nodes, Tree)
                    T.template get_frequency<true>(s, nup, q < a.nfreeze && q + 1 != len, lo, hi);
                    if (WINB) { // (two adjacent entries)
                        lo += bcum[s];
                        hi += bcum[s + 1];
                    }
                    pg[(uint64_t)(q - w0) * pw] = make_uint2(lo, hi);
                }
            }
        }
    }
}

template <bool WINB>
__global__ void __launch_bounds__(64) k_coop_model(EncArgs a, uint2 *pairs)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[kCoopModelDwords];
    coop_model_body<WINB>(a, pairs, blockIdx.x, lds);
}

// ======================================================================================
// The chain.  compress_symbol (codec.rs:55-89) has two halves that only talk one way: the interval (low, high -> the k
// bits low and high now share, the j E3 steps, the next low and high) never looks at the bit writer, and the bit writer
// (put_bit with its pending count, codec.rs:39-46, and bitio/mod.rs:148-181) needs only (those k bits, k, j).  So the
// workgroup is two waves on two SIMDs with an LDS ring between them:
//   wave 0, CHAIN: pairs from the workspace (requested two chunks ahead) -> narrowing + closed-form renormalisation
//                  (redux_coder.hpp, encode_symbol) -> one message (top k bits, k | j << 8) per symbol;
//   wave 1, EMIT : messages -> pending-bit bookkeeping, 64-bit accumulator, 4-byte stores into the row-major group area
//                  (what k_encode_pair's coder wave does after its narrowing), the EOF tail and the sizes.
// A lone wave issues one instruction per 4-5 cycles whatever it depends on (profiles/r01_ubench/lone_wave_gfx950.txt),
// so what counts is the instruction count of the longer half: ~25 and ~30 per symbol against the ~50 of both together.
// The ring holds two periods of kPeriod symbols (ds_write_b128 / ds_read_b128 of two symbols, as k_encode_pair's ring);
// one s_barrier per period hands a half over.  Both waves derive the same schedule from wave-uniform values: chunks of
// 16 symbols below main_end are lock-step with no lane predicate; from there to the longest block's EOF every symbol is
// predicated per lane (a lane past its EOF sends the empty message k = j = 0, which the bit writer ignores).
// ======================================================================================
constexpr uint32_t kPeriod    = 32;                        // symbols per hand-off
constexpr uint32_t kCoopRing  = 2 * kPeriod * 64 * 8;      // two periods of uint2[kPeriod][64]

struct ChainState {
    uint32_t low, ih, r1; // low and ~high left-aligned (ih with a stray bit 31, see encode_symbol_spec), r1 = high - low
};

// the narrowing half of encode_symbol_spec: all lanes live, data symbols only
template <bool CB32, bool FIXUP>
__device__ __forceinline__ uint2 chain_step(ChainState &X, uint32_t lo, uint32_t hi, uint32_t c, double rc, uint32_t sh_)
{
    const uint32_t sh = CB32 ? 0u : sh_;
    const uint32_t R1 = X.r1 >> sh;
    const double   Y  = __builtin_fma((double)R1, rc, rc);
    asm volatile("" : "+v"(hi)); // (the high half of an 8-byte load: keeps ISel from converting it as u64 >> 32, +1 v_add_f64; see coder_chunk)
    const uint32_t nlow   = X.low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
    const uint32_t nihigh = 0u - (X.low + (scale_div<FIXUP, true>(R1, Y, hi, c) << sh));
    const uint32_t x      = ~(nlow ^ nihigh);
    // (32-bit codes with count < 2^17: low != high after every symbol, encode_symbol_spec)
    constexpr bool kNonZero = CB32 && !FIXUP;
    const uint32_t k      = kNonZero ? (uint32_t)__builtin_clz(x) : (x ? (uint32_t)__builtin_clz(x) : 32u);
    const uint64_t sl     = (uint64_t)nlow << k;
    const uint32_t ih2    = kNonZero ? nihigh << k : (uint32_t)((uint64_t)nihigh << k);
    const uint32_t low2   = (uint32_t)sl;
    const uint32_t nt     = ((~(low2 & ih2)) << 1) | 1u;
    const uint32_t j      = (uint32_t)__builtin_clz(nt);
    const uint32_t L      = low2 << j;
    X.ih  = ih2 << j;
    X.r1  = ~(L + X.ih);
    X.low = L & 0x7FFFFFFFu;
    return make_uint2((uint32_t)(sl >> 32), k | (j << 8));
}

// the same for any lane state: act = this lane codes a symbol at this step, eof = it is the EOF symbol (codec.rs:91-99:
// high unchanged, encode_symbol); returns the message, shifts = k + j
template <bool CB32, bool FIXUP>
__device__ __forceinline__ uint2 chain_step_any(ChainState &X, uint32_t lo, uint32_t hi, uint32_t c, double rc, uint32_t sh,
                                                bool act, bool eof, uint32_t &shifts)
{
    const uint32_t ihm = X.ih & 0x7FFFFFFFu;
    const uint32_t R1  = (~(ihm + X.low)) >> sh;
    const double   Y   = __builtin_fma((double)R1, rc, rc);
    const uint32_t nlow   = X.low + (scale_div<FIXUP>(R1, Y, lo, c) << sh);
    const uint32_t nihigh = eof ? ihm : ~(X.low + (scale_div<FIXUP, true>(R1, Y, eof ? 1u : hi, c) << sh) - 1u);
    const uint32_t x      = ~(nlow ^ nihigh);
    const uint32_t k      = x ? (uint32_t)__builtin_clz(x) : 32u;
    const uint64_t sl     = (uint64_t)nlow << k;
    const uint32_t low2   = (uint32_t)sl;
    const uint32_t ih2    = (uint32_t)((uint64_t)nihigh << k);
    const uint32_t t      = (low2 & ih2) << 1;
    const uint32_t j      = (uint32_t)__builtin_clz(~t);
    shifts = k + j;
    if (act) {
        X.low = (low2 << j) & 0x7FFFFFFFu;
        X.ih  = (ih2 << j) & 0x7FFFFFFFu;
        X.r1  = ~(X.ih + X.low);
    }
    return act ? make_uint2((uint32_t)(sl >> 32), k | (j << 8)) : make_uint2(0, 0);
}

// the bit-writer half of encode_symbol_spec (redux_coder.hpp): returns the length of the append
template <int ST>
__device__ __forceinline__ uint32_t emit_spec(EncState &S, uint32_t &nbm, uint32_t topk, uint32_t kj, uint8_t *wbase)
{
    const uint32_t k = kj & 0xFFu, j = kj >> 8;
    const uint32_t P   = S.pend;
    const uint32_t Pz  = k ? P : 0u;
    const uint32_t km1 = k - 1u;
    S.pend             = P - Pz + j;
    const uint32_t m   = k + Pz;
    uint32_t       run;
    asm("v_bfm_b32 %0, %1, %2" : "=v"(run) : "v"(Pz), "v"(km1));
    const uint64_t sa = S.acc << (m & 63u);
    S.acc = (sa & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)sa + topk + run);
    const uint32_t nb = nbm + m;
    if ((int32_t)nb >= 0) {
        *reinterpret_cast<uint32_t *>(wbase + S.off) = stream_dword<ST>((uint32_t)(S.acc >> (nb & 63u)));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(S.off) : "i"(stride_of<ST>) : "memory");
    }
    nbm = nb | 0xFFFFFFE0u;
    return m;
}

// the bit-writer half of encode_symbol: any pending count, every store checked against `limit`
template <int ST>
__device__ __forceinline__ void emit_careful(EncState &S, uint32_t topk, uint32_t kj, uint8_t *wbase, uint32_t limit)
{
    const uint32_t k = kj & 0xFFu, j = kj >> 8;
    const uint32_t P  = S.pend;
    const uint32_t Pz = k ? P : 0u;
    S.pend            = P - Pz + j;
    if (k + Pz <= 32) {
        put_bits<ST>(S, topk + (((1u << (Pz & 31u)) - 1u) << ((k - 1u) & 31u)), k + Pz, wbase, limit);
    } else {
        const uint32_t b = topk >> (k - 1);
        put_bits<ST>(S, b, 1, wbase, limit);
        put_run<ST>(S, b ^ 1u, P, wbase, limit);
        put_bits<ST>(S, topk & ((1u << (k - 1)) - 1u), k - 1, wbase, limit);
    }
}

// LINEAR: the launch has fewer than 64 (large) blocks and its slots are linear -- a lane's dwords contiguous in its own
// slot -- instead of one row-major area of 64 slots: the workspace of such a launch (ONE block of any length above all:
// redux_compress) then holds nblocks slots, not 64.
// WIN: the launch codes one window of blocks that are coded window by window (every launch of blocks above 64 KiB); a lane
// without a symbol in the window then sends EMPTY messages instead of coding a copy of a neighbour's symbols -- its slot may
// hold a block that ended in an earlier window -- and stores nothing at all, so linear slots need no spare one.
constexpr uint32_t kCoopChainDwords = kCoopRing / 4 + 128; // the ring + fin
// (the body of the kernel; bid: the workgroup's group of 64 blocks; lds = kCoopChainDwords, 16-byte aligned)
template <bool CB32, bool FIXUP, bool LINEAR, bool WIN>
__device__ __forceinline__ void coop_chain_body(const EncArgs &a, const uint2 *pairs, const uint32_t bid, uint32_t *lds)
{
    static_assert(WIN || !LINEAR, "linear slots: blocks above 64 KiB, coded in windows");
    constexpr int ST = LINEAR ? (4 | kSwapped) : kPairStride;
    uint2 *const ring = reinterpret_cast<uint2 *>(lds);
    uint2 *const fin  = ring + kCoopRing / 8; // [64] (low after the EOF symbol, its shifts): what encode_finish needs from the chain
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk0 = (uint64_t)bid * 64;
    const uint64_t blk  = blk0 + lane;
    const bool     has  = blk < a.nblocks && !(a.table && a.table[blk].index == kIdleEntry);
    const uint8_t *wsrc;
    const EncLane  EL  = enc_lane(a, blk0, blk, lane, has, wsrc);
    // This launch codes the window [w0, w0 + wl) of every block (a.win0, a.winlen: redux_encode.hpp; one window of
    // block_size + 1 symbols when the pairs of whole blocks fit the workspace).  Positions below are window-relative: a lane
    // is live while its block has not ended before the window (its EOF symbol is coded in the window that holds symbol
    // number `len`), it has `nd` data symbols here, acts on positions 0 .. lastq, and eofq is its EOF symbol's position.
    const uint32_t w0   = WIN ? a.win0 : 0u, wl = a.winlen;
    const bool     live = has && EL.len >= w0;
    const uint32_t lend = EL.len - w0;
    const uint32_t nd   = lend < wl ? lend : wl;
    const uint32_t eofq = lend < wl ? lend : 0xFFFFFFFFu;
    const uint32_t lastq = lend < wl ? lend : wl - 1u;
    const uint32_t minlen = __builtin_amdgcn_readfirstlane(wave_min(live ? nd : 0xFFFFFFFFu));
    const uint32_t maxlast = __builtin_amdgcn_readfirstlane(wave_max(live ? lastq : 0u));
    const uint32_t sh     = 32 - a.code_bits;
    const rc_ptr   rc     = (rc_ptr)a.rc;
    const uint32_t nfreeze = a.nfreeze;
    const uint32_t nfz_w  = nfreeze > w0 ? nfreeze - w0 : 0u; // the freeze point, window-relative
    const uint64_t lives  = __builtin_amdgcn_ballot_w64(live);
    if (lives == 0)
        return;
    uint32_t main_end = 0;
    if (minlen > 16)
        main_end = (minlen - 1) & ~15u;
    const uint32_t nperiods = (maxlast + 1 + kPeriod - 1) / kPeriod; // positions 0 .. maxlast (the longest block's last one)
    uint32_t      *stp = (WIN && a.cstate && live) ? a.cstate + blk * 8 : nullptr; // this block's state between windows
    const uint32_t lmask = (!WIN || live) ? 0xFFFFFFFFu : 0u;
    auto slot = [&](uint32_t i) { return ring + ((i >> 1) * 128u + lane * 2u); }; // symbols i (even) and i + 1 of this lane, i < 2 kPeriod

    if (wave == 0) {
        // ---------------- chain wave ----------------
        // a lane without a block runs the lock-step part on a copy of the wave's first live block (valid pairs: its own
        // column of the workspace was never written) and sends empty messages after it
        const uint32_t col  = live ? lane : (uint32_t)__builtin_ctzll(lives);
        // the pairs: rows of 64 lanes per symbol and group (blocks of up to 64 KiB), or block-major (WIN: k_coop_model)
        const uint2 *pg = WIN ? pairs + (blk0 + col) * coop_block_pitch(a.winlen)
                              : pairs + (uint64_t)bid * ((uint64_t)a.winlen + kCoopSlack) * 64 + col;
        auto load16 = [&](uint2 (&d)[16], uint32_t p) { // (up to 47 symbols past the last chunk: inside the slack)
            if (WIN) { // a chunk is one 128-byte line of the lane's block (the pitch is even, p a multiple of 16)
                const uint4 *q = reinterpret_cast<const uint4 *>(pg + p);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint4 t = q[i];
                    d[2 * i]      = make_uint2(t.x, t.y);
                    d[2 * i + 1]  = make_uint2(t.z, t.w);
                }
            } else { // whole groups: the row offsets are immediates of the sixteen loads
                const uint2 *q = pg + (uint64_t)p * 64;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    d[i] = q[i * 64];
            }
        };
        ChainState X;
        X.low = 0; X.ih = 0; X.r1 = 0xFFFFFFFFu;
        if (stp && w0) {
            X.low = stp[0]; X.ih = stp[1]; X.r1 = stp[2];
        }
        const double rcF = a.rc_frozen;
        // Chunks of 16 symbols; the pairs of chunk c + 2 are requested when chunk c starts (three register buffers whose
        // roles rotate: the loop is unrolled by three so that no copy -- which would wait for the newest loads -- moves
        // them).  Every chunk is loaded without a lane predicate (a short block's column is garbage behind its end, the
        // slack keeps the addresses inside the area); what a lane does with a symbol is decided below.
        const uint32_t nchunks = 2 * nperiods;
        uint2 buf[3][16];
        load16(buf[0], 0);
        load16(buf[1], 16);
        // One chunk.  U: which of the three buffers holds it (a constant of the unrolled code).  HOT: the chunk is known to
        // lie below the shortest block's end AND below the freeze point -- sixteen lock-step symbols of an adapting model,
        // their counts an induction variable and their reciprocals two wide scalar loads, no branch; otherwise everything is
        // decided here: behind the freeze point the count and reciprocal are scalar selects per symbol (a lone wave pays an
        // issue slot for scalar instructions too, hence the separate hot copy), and from the shortest block's last chunk on
        // what a lane does with a symbol is its own matter.
        auto chunk = [&](auto u_tag, auto hot_tag, const uint32_t c) {
            constexpr int  U   = decltype(u_tag)::value;
            constexpr bool HOT = decltype(hot_tag)::value;
            const uint32_t p = 16 * c, ro = (c & 3u) * 16u;
            uint2 (&cur)[16] = buf[U];
            load16(buf[(U + 2) % 3], p + 32);
            // one wait for the whole chunk: the 32 loads issued since (this chunk's and the previous one's) may stay
            // in flight (vmcnt(32); the lgkmcnt / expcnt fields all ones = no wait)
            __builtin_amdgcn_s_waitcnt(0x8F70);
            if (HOT || p + 16 <= main_end) {
                double r[16]; // (the table covers the window + slack: rc[i] belongs to count 257 + w0 + i)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    r[i] = rc[p + i];
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    uint2 m[2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const bool     upd = HOT || p + i + e < nfz_w;
                        const uint32_t n   = upd ? w0 + p + i + e : nfreeze;
                        m[e] = chain_step<CB32, FIXUP>(X, cur[i + e].x, cur[i + e].y, 257u + n, upd ? r[i + e] : rcF, sh);
                    }
                    if (WIN)
                        *reinterpret_cast<uint4 *>(slot(ro + i)) = make_uint4(m[0].x & lmask, m[0].y & lmask, m[1].x & lmask, m[1].y & lmask);
                    else
                        *reinterpret_cast<uint4 *>(slot(ro + i)) = make_uint4(m[0].x, m[0].y, m[1].x, m[1].y);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    uint2 m[2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const uint32_t q   = p + i + e;
                        const bool     act = live && q <= lastq, eof = q == eofq;
                        const bool     upd = q < nfz_w;
                        const uint32_t qc  = upd ? w0 + q : nfreeze; // updates before symbol q
                        const uint32_t qr  = q < wl + 32u ? q : wl + 32u; // (the table ends 64 entries behind the window)
                        uint32_t       shifts;
                        m[e] = chain_step_any<CB32, FIXUP>(X, eof ? 256u + qc : cur[i + e].x, cur[i + e].y, 257u + qc, upd ? rc[qr] : rcF, sh, act, eof, shifts);
                        if (act && eof)
                            fin[lane] = make_uint2(X.low, shifts);
                    }
                    *reinterpret_cast<uint4 *>(slot(ro + i)) = make_uint4(m[0].x, m[0].y, m[1].x, m[1].y);
                }
            }
            if (c & 1u)
                pair_barrier();
        };
        typedef std::integral_constant<int, 0> U0;
        typedef std::integral_constant<int, 1> U1;
        typedef std::integral_constant<int, 2> U2;
        const uint32_t hot_end = main_end < nfz_w ? main_end : nfz_w; // symbols that are lock-step AND adaptive
        uint32_t       c0      = 0;
        for (; 16 * (c0 + 3) <= hot_end; c0 += 3) { // whole triples of hot chunks: one straight run of code
            chunk(U0(), std::true_type(), c0);
            chunk(U1(), std::true_type(), c0 + 1);
            chunk(U2(), std::true_type(), c0 + 2);
        }
        for (; c0 < nchunks; c0 += 3) {
            chunk(U0(), std::false_type(), c0);
            if (c0 + 1 < nchunks)
                chunk(U1(), std::false_type(), c0 + 1);
            if (c0 + 2 < nchunks)
                chunk(U2(), std::false_type(), c0 + 2);
        }
        if (stp && eofq == 0xFFFFFFFFu) { // the block goes on in the next window
            stp[0] = X.low; stp[1] = X.ih; stp[2] = X.r1;
        }
        return;
    }

    // ---------------- emit wave ----------------
    uint8_t       *wdst  = LINEAR ? a.slots : a.slots + (uint64_t)bid * (64 * a.slot_bytes + 128);
    const uint32_t off0  = LINEAR ? (live ? lane : 0u) * (uint32_t)a.slot_bytes : lane * 4u; // (LINEAR: a lane that is not live never stores)
    const uint32_t limit = LINEAR ? off0 + (a.slot_cap & ~3u) : off0 + (a.slot_cap / 4u) * 256u;
    constexpr uint32_t kChunkBudget = (16 * 4 + 32) * (stride_of<ST> / 4);
    EncState S;
    enc_init(S, off0);
    if (stp && w0) {
        S.pend = stp[3]; S.nb = stp[4]; S.off = stp[5];
        S.acc  = ((uint64_t)stp[7] << 32) | stp[6];
    }
    for (uint32_t t = 0; t < nperiods; t++) {
        pair_barrier();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t p = t * kPeriod + 16 * h, ro = (t & 1) * kPeriod + 16 * h;
            uint2 msg[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const uint4 two = *reinterpret_cast<const uint4 *>(slot(ro + i));
                msg[i]     = make_uint2(two.x, two.y);
                msg[i + 1] = make_uint2(two.z, two.w);
            }
            // (a lane without a symbol at some step gets the empty message, which changes nothing: every chunk takes the
            // straight-line path while the slots have room; a block's EOF tail is written after the group of eight that holds it)
            const bool fast = __builtin_amdgcn_ballot_w64(S.off + kChunkBudget > limit) == 0;
            if (fast) {
                // eight symbols straight-line; the rare append of more than 32 bits only raises a flag, and the eight
                // are then redone from the saved state with the general routine (as coder_chunk, redux_encode.hpp)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const EncState S0  = S;
                    uint32_t       nbm = S.nb - 32u, mx = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t m = emit_spec<ST>(S, nbm, msg[8 * g + i].x, msg[8 * g + i].y, wdst);
                        mx = m > mx ? m : mx;
                    }
                    S.nb = nbm + 32u;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > 32u) != 0, 0)) {
                        S = S0;
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            emit_careful<ST>(S, msg[8 * g + i].x, msg[8 * g + i].y, wdst, 0xFFFFFFFFu);
                    }
                    const uint32_t pg = p + 8 * g;
                    if (pg + 8 > main_end && live && eofq >= pg && eofq - pg < 8u) { // this lane's EOF symbol was among the eight
                        const uint2 f = fin[lane];
                        S.low         = f.x;
                        const uint32_t size = encode_finish<ST>(S, f.y, a.code_bits, off0, wdst, limit);
                        a.sizes[EL.ob]  = size;
                        a.status[EL.ob] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
                    }
                }
            } else {
#pragma unroll 1
                for (uint32_t i = 0; i < 16; i++) {
                    uint2 mg = msg[0];
#pragma unroll
                    for (int k = 1; k < 16; k++)
                        mg = i == (uint32_t)k ? msg[k] : mg;
                    const uint32_t q = p + i;
                    if (q < main_end || (live && q <= lastq)) // (below main_end the other lanes get empty messages)
                        emit_careful<ST>(S, mg.x, mg.y, wdst, limit);
                    if (live && q == eofq) {
                        const uint2 f = fin[lane];
                        S.low         = f.x;
                        const uint32_t size = encode_finish<ST>(S, f.y, a.code_bits, off0, wdst, limit);
                        a.sizes[EL.ob]  = size;
                        a.status[EL.ob] = size > a.slot_cap ? REDUX_OUTPUT_TOO_SMALL : REDUX_OK;
                    }
                }
            }
        }
    }
    if (stp && eofq == 0xFFFFFFFFu) { // the block goes on in the next window
        stp[3] = S.pend; stp[4] = S.nb; stp[5] = S.off;
        stp[6] = (uint32_t)S.acc; stp[7] = (uint32_t)(S.acc >> 32);
    }
}

template <bool CB32, bool FIXUP, bool LINEAR = false, bool WIN = LINEAR>
__global__ void __launch_bounds__(128) k_coop_chain(EncArgs a, const uint2 *pairs)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[kCoopChainDwords];
    coop_chain_body<CB32, FIXUP, LINEAR, WIN>(a, pairs, blockIdx.x, lds);
}

// One window's chain NEXT TO the model of the window after it (blocks coded in windows): workgroups [0, cgrid) are chain
// workgroups of window w over the pairs the launch before left in pc, the others run the model of window w + 1 -- one wave
// each, the second returns at once -- into the other pairs buffer.  Nothing in the launch depends on anything else in it; the
// chain workgroups have the low numbers, so they are dispatched first.  Both roles take ~33 KiB of LDS: four workgroups per
// CU in any mix.
template <bool CB32, bool FIXUP, bool LINEAR>
__global__ void __launch_bounds__(128) k_coop_step(EncArgs ac, EncArgs am, const uint2 *pc, uint2 *pm, uint32_t cgrid)
{
    constexpr uint32_t kDwords = kCoopChainDwords > kCoopModelDwords ? kCoopChainDwords : kCoopModelDwords;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kDwords];
    if (blockIdx.x < cgrid) {
        coop_chain_body<CB32, FIXUP, LINEAR, true>(ac, pc, blockIdx.x, lds);
    } else {
        if (threadIdx.x >= 64)
            return;
        coop_model_body<true>(am, pm, blockIdx.x - cgrid, lds);
    }
}

} // namespace redux
