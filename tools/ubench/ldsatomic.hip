// LDS atomics / reads with PER-LANE ROWS (what a per-lane frequency tree does): does a wave-wide ds_add_u32 whose
// lanes hit different rows (each lane its own bank) cost more than one whose lanes hit one row?
// One 64-thread workgroup with 32 KiB of LDS per launch slot -> 4 waves per CU, one per SIMD, like k_decode_lock.
// hipcc --offload-arch=gfx950 -O2 -o ldsatomic tools/ubench/ldsatomic.hip && ./ldsatomic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define S4(x) x x x x
#define S16(x) S4(S4(x))
#define S64(x) S4(S16(x))

// PAT 0: every lane row 0 (address = 4*lane); 1: random row per lane, 256-B rows, own dword column;
//     2: random row per lane, 128-B rows, lanes l and l+32 share a dword (encoder layout); 3: random row, 512-B rows, own 8-byte column
template <int OP, int PAT>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed, unsigned long long *clk, int iters)
{
    __shared__ uint32_t lds[8192]; // 32 KiB
    for (uint32_t i = threadIdx.x; i < 8192; i += 64) lds[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    uint32_t h = (lane * 2654435761u + seed) >> 8;
    uint32_t a[4];
    for (int j = 0; j < 4; j++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 10);
        if (PAT == 0) a[j] = lane * 4;
        if (PAT == 1) a[j] = ((row & 127) << 8) | (lane * 4);
        if (PAT == 2) a[j] = ((row & 255) << 7) | ((lane & 31) * 4);
        if (PAT == 3) a[j] = ((row & 63) << 9) | (lane * 8);
    }
    uint32_t one = 1, r0 = 0;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#define A(body) asm volatile(body "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(one) : "memory", "v40", "v41", "v42", "v43", "v44", "v45")
        if (OP == 0) A(S16("ds_add_u32 %1, %5\n\tds_add_u32 %2, %5\n\tds_add_u32 %3, %5\n\tds_add_u32 %4, %5\n\t"));
        if (OP == 1) A(S16("ds_add_rtn_u32 v40, %1, %5\n\tds_add_rtn_u32 v41, %2, %5\n\tds_add_rtn_u32 v42, %3, %5\n\tds_add_rtn_u32 v43, %4, %5\n\t"));
        if (OP == 2) A(S16("ds_read_b32 v40, %1\n\tds_read_b32 v41, %2\n\tds_read_b32 v42, %3\n\tds_read_b32 v43, %4\n\t"));
        if (OP == 3) A(S16("ds_write_b32 %1, %5\n\tds_write_b32 %2, %5\n\tds_write_b32 %3, %5\n\tds_write_b32 %4, %5\n\t"));
        if (OP == 4) A(S16("ds_add_u64 %1, v[44:45]\n\tds_add_u64 %2, v[44:45]\n\tds_add_u64 %3, v[44:45]\n\tds_add_u64 %4, v[44:45]\n\t"));
        if (OP == 5) A(S16("ds_read_b64 v[40:41], %1\n\tds_read_b64 v[42:43], %2\n\tds_read_b64 v[40:41], %3\n\tds_read_b64 v[42:43], %4\n\t"));
        if (OP == 6) A(S16("ds_read_b128 v[40:43], %1\n\tds_read_b128 v[40:43], %2\n\tds_read_b128 v[40:43], %3\n\tds_read_b128 v[40:43], %4\n\t"));
        // 5 atomics, then 40 VALU, then a dependent read: the decoder's "update, then next step's first probe"
        if (OP == 7) A(S4("ds_add_u32 %1, %5\n\tds_add_u32 %2, %5\n\tds_add_u32 %3, %5\n\tds_add_u32 %4, %5\n\tds_add_u32 %1, %5\n\t"
                          S16("v_add_u32 %0, %0, %5\n\t") S16("v_add_u32 %0, %0, %5\n\t") "v_add_u32 %0, %0, %5\n\t"
                          "ds_read_b32 v40, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v40\n\t"));
        if (OP == 8) A(S4(S16("v_add_u32 %0, %0, %5\n\t") S16("v_add_u32 %0, %0, %5\n\t") "v_add_u32 %0, %0, %5\n\t"
                          "ds_read_b32 v40, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v40\n\t"));
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = r0 + lds[lane];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
struct Test { const char *name; int n; void (*fn)(uint32_t *, uint32_t, unsigned long long *, int); };
int main()
{
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&clk, 8);
#define ROW(op, name, n) {name " | one row", n, k<op, 0>}, {name " | row per lane, own dword", n, k<op, 1>}, {name " | row per lane, lanes l/l+32 share a dword", n, k<op, 2>}, {name " | row per lane, own 8-byte column", n, k<op, 3>}
    Test tests[] = {ROW(0, "ds_add_u32", 64), ROW(1, "ds_add_rtn_u32", 64), ROW(2, "ds_read_b32", 64), ROW(3, "ds_write_b32", 64),
                    {"ds_add_u64 | one row (8-byte column)", 64, k<4, 3>}, {"ds_add_u64 | row per lane, own 8-byte column", 64, k<4, 3>},
                    {"ds_read_b64 | row per lane, own 8-byte column", 64, k<5, 3>}, {"ds_read_b128 | one row", 64, k<6, 0>},
                    {"[5 ds_add + 33 v_add + dependent ds_read] per group | row per lane", 4, k<7, 1>},
                    {"[33 v_add + dependent ds_read] per group | row per lane", 4, k<8, 1>}};
    for (auto &t : tests) {
        const int iters = 256;
        for (int rep = 0; rep < 2; rep++) { t.fn<<<1024, 64>>>(out, 12345, clk, iters); hipDeviceSynchronize(); }
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-78s %8.2f cycles per %s\n", t.name, (double)c / ((double)iters * t.n), t.n == 4 ? "group" : "instr");
    }
    return 0;
}
