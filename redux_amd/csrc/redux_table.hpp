// redux_table.hpp -- the block table of the `_v_dev` calls is CALLER data in device memory: checked before any coder kernel
// reads it (gfx950 only).
//
// The reference's surface cannot be made to write out of bounds -- BitWriter::write_bits returns Err (bitio/mod.rs:148-198),
// a Vec grows --, so neither may a table: the coder kernels take entry.offset, entry.length and entry.index as given.  One
// thread per entry, tens of microseconds: an entry must be idle (index REDUX_BLOCK_IDLE) or name a block below nblocks that
// no other entry names, with length <= block_size, offset + length inside the buffer and, where the caller promised
// REDUX_V_ALIGNED16, a 16-byte aligned offset.  The kernels then read a COPY of the table in the workspace in which every
// entry that failed is idle; a block that no valid entry codes keeps size 0 and status InvalidInput, and the call's
// summary says InvalidInput.
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

struct TableCheckArgs {
    const redux_block *in;      // the caller's table
    redux_block       *out;     // its checked copy
    uint64_t           nentries, nblocks;
    uint64_t           bytes;   // size of the buffer entry.offset points into
    uint32_t           block_size, aligned16;
    uint32_t          *seen;    // bitmap of block numbers + one word behind it: entries that failed
    uint32_t          *sizes;   // per block: 0 until its coder lane says otherwise
    int32_t           *status;  // per block: InvalidInput until its coder lane says otherwise
};
__host__ __device__ static inline uint64_t table_seen_words(uint64_t nblocks) { return (nblocks + 31) / 32 + 1; }

__global__ void k_table_prepare(TableCheckArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nblocks) {
        a.sizes[i]  = 0;
        a.status[i] = REDUX_INVALID_INPUT;
    }
    if (i < table_seen_words(a.nblocks))
        a.seen[i] = 0;
}

__global__ void k_table_check(TableCheckArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nentries)
        return;
    redux_block e = a.in[i];
    if (e.index != REDUX_BLOCK_IDLE) {
        bool ok = e.index < a.nblocks && e.length <= a.block_size && e.offset <= a.bytes && e.length <= a.bytes - e.offset &&
                  (!a.aligned16 || (e.offset & 15) == 0);
        if (ok) { // the first entry to claim a block number keeps it
            const uint32_t bit = 1u << (e.index & 31u);
            ok                 = (atomicOr(&a.seen[e.index >> 5], bit) & bit) == 0;
        }
        if (!ok) {
            atomicAdd(&a.seen[table_seen_words(a.nblocks) - 1], 1u);
            e.offset = 0;
            e.length = 0;
            e.index  = REDUX_BLOCK_IDLE;
        }
    }
    a.out[i] = e;
}

// last kernel of a `_v_dev` call: entries that failed the check make the call's summary InvalidInput
__global__ void k_table_verdict(const uint32_t *failed, int32_t *summary)
{
    const uint32_t n = *failed;
    if (n && summary) {
        atomicCAS(&summary[0], REDUX_OK, REDUX_INVALID_INPUT);
        atomicAdd(&summary[1], (int32_t)n);
    }
}

} // namespace redux
