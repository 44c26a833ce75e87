// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the DECODERS' access pattern: every lane reads its own
// stream 16 bytes at a time (64 lanes of a wave = 64 streams STRIDE bytes apart) and writes its own block 16 or 2 x 16 bytes at
// a time.  Known byte counts; run each kernel under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes,
// tools/prof_fetchsize.sh) and compare.  k_stream: the coalesced control (FETCH_SIZE reads half the bytes there,
// MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// coalesced: thread t reads 16 bytes at 16 t, grid-stride
__global__ void k_stream(const u32x4 *in, u32x4 *out, uint64_t n16)
{
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        acc += in[i];
    if (acc.x == 0x12345678u)
        out[0] = acc;
}

// per-lane streams: lane of global index g reads 16-byte chunks 0 .. nchunks-1 of the stream at g * stride, one chunk per
// `gap` dependent iterations of filler work (so that a wave's chunks arrive as separate requests, like the decoder's)
template <int STORE> // 0: no store, 1: one 16-byte store per chunk, 2: two adjacent 16-byte stores per two chunks
__global__ void __launch_bounds__(64) k_lanes(const uint8_t *in, uint8_t *out, uint64_t stride, uint32_t nchunks, uint32_t gap)
{
    const uint64_t g   = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const u32x4   *src = reinterpret_cast<const u32x4 *>(in + g * stride);
    u32x4         *dst = reinterpret_cast<u32x4 *>(out + g * stride);
    u32x4          acc = {0, 0, 0, 0}, prev = {0, 0, 0, 0};
    for (uint32_t c = 0; c < nchunks; c++) {
        const u32x4 v = src[c];
        acc += v;
        for (uint32_t k = 0; k < gap; k++) // filler: a dependent chain on the loaded value
            acc.x = acc.x * 1664525u + 1013904223u;
        if (STORE == 1)
            dst[c] = acc;
        if (STORE == 2) {
            if (c & 1) {
                dst[c - 1] = prev;
                dst[c]     = acc;
            } else
                prev = acc;
        }
    }
    if (STORE == 0 && acc.x == 0x12345678u)
        dst[0] = acc;
}

int main(int argc, char **argv) // (return values of the set-up calls are not checked: a failed allocation shows as zero counters)
{
    const uint64_t stride = 65536, nlanes = 65536; // 4 GiB, the decoder's shape
    const uint32_t nchunks = argc > 1 ? atoi(argv[1]) : 4096, gap = argc > 2 ? atoi(argv[2]) : 64;
    uint8_t *in, *out;
    if (hipMalloc(&in, stride * nlanes) != hipSuccess || hipMalloc(&out, stride * nlanes) != hipSuccess)
        return 1;
    hipMemset(in, 1, stride * nlanes);
    hipMemset(out, 0, stride * nlanes);
    hipDeviceSynchronize();
    k_stream<<<4096, 256>>>((const u32x4 *)in, (u32x4 *)out, stride * nlanes / 16);
    k_lanes<0><<<nlanes / 64, 64>>>(in, out, stride, nchunks, gap);
    k_lanes<1><<<nlanes / 64, 64>>>(in, out, stride, nchunks, gap);
    k_lanes<2><<<nlanes / 64, 64>>>(in, out, stride, nchunks, gap);
    hipDeviceSynchronize();
    printf("k_stream reads %.0f KiB; k_lanes<*> read %.0f KiB each; k_lanes<1>, <2> write %.0f KiB each (chunks %u, gap %u)\n",
           stride * nlanes / 1024.0, (double)nlanes * nchunks * 16 / 1024.0, (double)nlanes * nchunks * 16 / 1024.0, nchunks, gap);
    return 0;
}
