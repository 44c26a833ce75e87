// Does a VALU instruction cost less when only part of the wave64 is active (EXEC masks out quarters)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define S4(x) x x x x
#define S16(x) S4(S4(x))
#define S256(x) S16(S16(x))
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed, unsigned long long *clk, int active)
{
    uint32_t x = seed + threadIdx.x, y = seed * 3 + 1;
    double   d = 1.0 + threadIdx.x, e = 1.0000001;
    unsigned long long c0 = 0, c1 = 0;
    if ((int)threadIdx.x < active) {
        c0 = clock64();
        for (int it = 0; it < 64; it++) {
            asm volatile(S256("v_add_u32 %0, %0, %1\n\t") : "+v"(x) : "v"(y));
            asm volatile(S256("v_fma_f64 %0, %0, %1, %1\n\t") : "+v"(d) : "v"(e));
        }
        c1 = clock64();
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + (uint32_t)d;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
int main()
{
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&clk, 8);
    for (int active : {64, 48, 32, 16, 8, 1}) {
        k<<<1024, 64>>>(out, 12345, clk, active); hipDeviceSynchronize();
        k<<<1024, 64>>>(out, 12345, clk, active); hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("active lanes %2d: %6.2f cycles per instruction (v_add + v_fma_f64 mix)\n", active, (double)c / (64.0 * 512));
    }
    return 0;
}
