// redux_synth.hpp -- synthetic workloads generated straight into HBM (bench.py, tests).
//
//   k_gen_iid    uniform bytes, splitmix64 of the byte index (reproducible per position)
//   k_gen_zipf   Zipf(1.2) bytes through a 256-entry threshold table (zipf_table.inc)
//
// Included by redux_hip.hip (one translation unit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace redux {

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

} // namespace redux
