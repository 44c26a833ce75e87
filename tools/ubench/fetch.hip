// Does a lone wave's issue rate depend on the SIZE of the loop body (instruction fetch)?
// Straight-line bodies of N dependent or independent VALU instructions, 4-byte (VOP2) and
// 8-byte (VOP3) encodings, one 64-thread workgroup per CU.  Prints ticks per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define S4(x) x x x x
#define S16(x) S4(S4(x))
#define S64(x) S4(S16(x))
#define S256(x) S4(S64(x))
#define S1024(x) S4(S256(x))

#define VOP2D "v_add_u32 %0, %0, %2\n\t"
#define VOP3D "v_add3_u32 %0, %0, %2, 1\n\t"
#define VOP2I "v_add_u32 %0, %2, %3\n\tv_add_u32 %1, %2, %3\n\t"
#define VOP3I "v_add3_u32 %0, %2, %3, 1\n\tv_add3_u32 %1, %2, %3, 1\n\t"

template <int T>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed, unsigned long long *clk, int iters)
{
    uint32_t x = seed + threadIdx.x, w = 1, y = seed * 3 + 1, z = threadIdx.x;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (T == 0) asm volatile(S16(VOP2D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 1) asm volatile(S256(VOP2D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 2) asm volatile(S1024(VOP2D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 3) asm volatile(S16(VOP3D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 4) asm volatile(S256(VOP3D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 5) asm volatile(S1024(VOP3D) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 6) asm volatile(S16(VOP2I) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 7) asm volatile(S256(VOP2I) S256(VOP2I) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 8) asm volatile(S16(VOP3I) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
        if (T == 9) asm volatile(S256(VOP3I) S256(VOP3I) : "+v"(x), "+v"(w) : "v"(y), "v"(z));
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x + w;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

struct Test { const char *name; int n; void (*fn)(uint32_t *, uint32_t, unsigned long long *, int); };

int main()
{
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&clk, 8);
    Test tests[] = {
        {"VOP2 dependent, body 16", 16, k<0>}, {"VOP2 dependent, body 256", 256, k<1>}, {"VOP2 dependent, body 1024", 1024, k<2>},
        {"VOP3 dependent, body 16", 16, k<3>}, {"VOP3 dependent, body 256", 256, k<4>}, {"VOP3 dependent, body 1024", 1024, k<5>},
        {"VOP2 independent, body 32", 32, k<6>}, {"VOP2 independent, body 1024", 1024, k<7>},
        {"VOP3 independent, body 32", 32, k<8>}, {"VOP3 independent, body 1024", 1024, k<9>},
    };
    for (auto &t : tests)
        for (int grid : {1, 1024, 2048}) { // idle chip / one wave per SIMD everywhere / two per SIMD
            const int iters = 65536 / t.n;
            t.fn<<<grid, 64>>>(out, 12345, clk, iters);
            hipDeviceSynchronize();
            t.fn<<<grid, 64>>>(out, 12345, clk, iters);
            hipDeviceSynchronize();
            unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("%-32s grid %4d: %7.2f ticks/instr\n", t.name, grid, (double)c / ((double)iters * t.n));
        }
    return 0;
}
