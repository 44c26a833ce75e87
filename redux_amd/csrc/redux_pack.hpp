// redux_pack.hpp -- what turns per-block slots into the dense stream (gfx950 only).
//
//   k_summarize      first failing status + number of failing blocks
//   k_scan_sizes     sizes -> offsets (exclusive scan, one workgroup) + status summary;
//                    k_scan_sizes_coalesced for whole chunks of 4096 blocks (the headline shape)
//   k_compact        slot b [0, size_b) -> out + offsets[b], 16-byte stores with byte realignment
//   k_compact_rows   the same from row-major group areas (what k_encode_pair leaves)
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "redux_coder.hpp"
#include "redux_encode.hpp"

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

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

static inline bool scan_is_coalesced(const ScanArgs &a)
{
    return a.nblocks % 4096 == 0 && a.nblocks / 4096 <= 16 &&
           ((((uintptr_t)a.sizes) | ((uintptr_t)a.status) | ((uintptr_t)a.offsets)) & 15) == 0;
}

// sizes -> offsets for whole chunks of 4096 blocks, at most sixteen of them (the 65,536-block
// configuration): see scan_is_coalesced() for when the host picks this kernel.
__global__ void __launch_bounds__(1024) k_scan_sizes_coalesced(ScanArgs a)
{
    __shared__ uint32_t bad_cnt;
    __shared__ uint64_t bad_first; // (index << 8) | status, minimised
    __shared__ uint64_t wtot[2][16];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        bad_cnt   = 0;
        bad_first = ~0ull;
    }
    // Thread t takes quad t of every chunk, so a wave's loads and stores are contiguous (1 KiB of
    // sizes, 2 KiB of offsets per instruction); all size loads are issued up front, then one wave
    // scan and one barrier per chunk.  (k_scan_sizes' per-thread contiguous ranges cost 64
    // different lines per load, and its two 64-register arrays spill: 113 us against this.)
    const uint32_t nch  = (uint32_t)(a.nblocks / 4096);
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint4   *s4   = reinterpret_cast<const uint4 *>(a.sizes);
    const int4    *t4   = reinterpret_cast<const int4 *>(a.status);
    // statuses first, packed to a byte each as they arrive (a status is 0..5), then the sizes: 16 + 64
    // registers live in the loop instead of 128, and no load inside it
    uint32_t stp[16];
    {
        int4 stv[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            stv[i] = (uint32_t)i < nch ? t4[i * 1024 + tid] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; i++)
            stp[i] = ((uint32_t)stv[i].x & 0xFFu) | (((uint32_t)stv[i].y & 0xFFu) << 8) | (((uint32_t)stv[i].z & 0xFFu) << 16) |
                     ((uint32_t)stv[i].w << 24);
    }
    uint4 sz[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        sz[i] = (uint32_t)i < nch ? s4[i * 1024 + tid] : make_uint4(0, 0, 0, 0);
    uint64_t running = 0;
    uint32_t nb      = 0;
    uint64_t fb      = ~0ull;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if ((uint32_t)i >= nch) // (uniform)
            continue;
        // inclusive scan over the wave in 32 bits: 256 sizes of one wave stay below 2^32 for any
        // input that fits in memory; wave totals and everything above them are 64-bit
        const uint32_t qs   = sz[i].x + sz[i].y + sz[i].z + sz[i].w;
        uint32_t       incl = qs;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            incl += lane >= (uint32_t)o ? v : 0u;
        }
        if (lane == 63)
            wtot[i & 1][wave] = incl;
        __syncthreads(); // one barrier per chunk: wtot is double-buffered, and chunk i+2's writers have passed chunk i+1's barrier
        uint64_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint64_t v = wtot[i & 1][w];
            before += (uint32_t)w < wave ? v : 0;
            total += v;
        }
        const uint64_t q  = (uint64_t)i * 1024 + tid;
        const uint64_t o0 = running + before + (incl - qs);
        const uint64_t o1 = o0 + sz[i].x, o2 = o1 + sz[i].y, o3 = o2 + sz[i].z;
        ulonglong2    *op = reinterpret_cast<ulonglong2 *>(a.offsets + 4 * q);
        op[0]             = make_ulonglong2(o0, o1);
        op[1]             = make_ulonglong2(o2, o3);
        running += total;
        if (stp[i] != 0) { // (rare) some block of this quad failed
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t st = (stp[i] >> (8 * k)) & 0xFFu;
                if (st != REDUX_OK) {
                    nb++;
                    const uint64_t cand = ((4 * q + k) << 8) | st;
                    fb                  = cand < fb ? cand : fb;
                }
            }
        }
    }
    if (nb) {
        atomicAdd(&bad_cnt, nb);
        atomicMin((unsigned long long *)&bad_first, (unsigned long long)fb);
    }
    __syncthreads();
    if (tid == 0) {
        a.offsets[a.nblocks] = running;
        if (a.summary) {
            a.summary[0] = bad_cnt ? (int32_t)(bad_first & 0xFF) : REDUX_OK;
            a.summary[1] = (int32_t)bad_cnt;
        }
    }
}

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
    const uint32_t *mode;     // bit 0: row-major group areas (k_compact_rows) instead of linear slots (k_compact);
                              // bit 1: every aligned dword of a slot is byte-reversed (kSwapped, redux_coder.hpp)
    uint32_t        cap_rows; // rows of a group area
    const redux_block *table; // block table (null: slot b holds block b): slot b holds the block numbered table[b].index
};

__global__ void __launch_bounds__(256) k_compact(CompactArgs a)
{
    const uint64_t b = blockIdx.x;
    const uint32_t mode = *a.mode;
    if (b >= a.nblocks || (mode & 1u))
        return;
    const uint32_t bx  = (mode & 2u) ? 3u : 0u;                   // slot byte that holds stream byte i: i ^ bx
    const uint32_t sel = (mode & 2u) ? 0x00010203u : 0x03020100u; // v_perm selector that restores stream order
    const uint64_t lb = a.table ? a.table[b].index : b;
    if (lb == kIdleEntry) // an idle table entry: nothing was coded in this slot
        return;
    const uint64_t o0 = a.offsets[lb], o1 = a.offsets[lb + 1];
    const uint32_t tid = threadIdx.x;
    if (o1 > a.out_cap) { // the dense buffer is too small for this block: report, never write
        if (tid == 0) {
            if (a.status[lb] == REDUX_OK)
                a.status[lb] = REDUX_OUTPUT_TOO_SMALL;
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
        dst[tid] = src[tid ^ bx];
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
            v[k] = __builtin_amdgcn_perm(0u, dq == 0 ? d[k] : dq == 1 ? d[k + 1] : dq == 2 ? d[k + 2] : d[k + 3], sel);
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
        dst[done + tid] = src[(done + tid) ^ bx];
}

// Row-major group areas: row r of group g holds dword r of its 64 streams
// (slots + g * 64 * slot_bytes + 256 r + 4 l).  One workgroup gathers a tile of 64 rows: the
// rows are read whole (coalesced) into LDS, then every stream's 64 dwords of the tile leave as
// one 256-byte run of ALIGNED dwords of the dense output: output dword j of a stream that starts
// at byte offset sh (0..3) inside its first aligned dword is the byte-funnel of source dwords
// j-1 and j.  Only a stream's first and last output dword can be partial: those go bytewise.
constexpr uint32_t kTileRows = 64;
__global__ void __launch_bounds__(256) k_compact_rows(CompactArgs a)
{
    const uint32_t mode = *a.mode;
    if ((mode & 1u) == 0)
        return;
    const uint32_t sel = (mode & 2u) ? 0x00010203u : 0x03020100u; // v_perm selector that restores stream order (kSwapped)
    __shared__ uint32_t tile[(kTileRows + 1) * 65]; // +1 leading row (source dword j-1); pitch 65: conflict-free column reads
    __shared__ uint64_t s_dst[64];                  // aligned dword that holds each stream's first byte (0: skip the stream)
    __shared__ uint32_t s_n[64], s_sh[64];
    __shared__ uint32_t s_maxj;
    const uint32_t tiles = (a.cap_rows + kTileRows - 1) / kTileRows + 1;
    // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2.  A
    // stream's consecutive 256-byte runs come from consecutive tiles of its group and share cache lines at their
    // ends: all tiles of a group go to ONE XCD (blockIdx = 8 * (8-group block * tiles + tile) + group's slot), so the
    // partial lines merge in that L2 instead of leaving two L2s as two partial writes.
    const uint32_t y   = blockIdx.x >> 3;
    const uint64_t g   = (uint64_t)(y / tiles) * 8 + (blockIdx.x & 7u);
    const uint32_t r0  = (y % tiles) * kTileRows;
    if (g * 64 >= a.nblocks)
        return;
    const uint32_t tid   = threadIdx.x;
    if (tid == 0)
        s_maxj = 0;
    __syncthreads();
    if (tid < 64) { // where does each stream go, and how many output dwords does the longest one need?
        const uint64_t b = g * 64 + tid;
        uint64_t       d = 0;
        uint32_t       n = 0, sh = 0;
        const uint64_t lb = b < a.nblocks ? (a.table ? a.table[b].index : b) : kIdleEntry;
        if (lb != kIdleEntry) {
            const uint64_t o0 = a.offsets[lb], o1 = a.offsets[lb + 1];
            if (o1 <= a.out_cap) {
                n  = (uint32_t)(o1 - o0);
                sh = (uint32_t)((uintptr_t)(a.out + o0) & 3);
                d  = (uint64_t)(uintptr_t)(a.out + o0) - sh;
                atomicMax(&s_maxj, (sh + n + 3) >> 2);
            } else if (r0 == 0) { // the dense buffer is too small for this block: report, never write
                if (a.status[lb] == REDUX_OK)
                    a.status[lb] = REDUX_OUTPUT_TOO_SMALL;
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
            src[c] = __builtin_amdgcn_perm(0u, tile[(jq + c) * 65 + l], sel); // source dwords j0-1 .. j0+3, in stream order
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

} // namespace redux
