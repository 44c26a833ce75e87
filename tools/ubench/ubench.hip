// Micro-benchmarks of the instructions the coder's hot loops are made of (gfx950).
// One 64-thread workgroup with 32 KiB of LDS per wave (4 waves/CU = 1 per SIMD) or
// 128-thread workgroups with 40 KiB (8 waves/CU = 2 per SIMD), like the real kernels.
// Prints cycles per wave-instruction assuming the clock reported by hipDeviceProp.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
#define ITER 4096

template <int T>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    extern __shared__ uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *my = lds; (void)wave;   // all waves of the workgroup share one 32 KiB region
    for (uint32_t i = lane; i < 8192; i += 64) my[i] = i * seed;
    __syncthreads();
    uint32_t a[8], x[8];
    const uint32_t col = (T == 4) ? (lane & 31) * 4 : lane * 4;        // T4: two lanes per dword (l, l+32)
    const uint32_t col2 = (lane >> 1) * 4;                             // 2-way conflict layout
#pragma unroll
    for (int b = 0; b < 8; b++) { a[b] = ((seed * (b + 3) + lane * 7 + b * 11) & 127) * 128 + ((T == 15) ? col2 : (col & 127)); x[b] = seed + b + lane; }
    double d0 = seed * 1.5, d1 = 0.001 * lane, d2 = 3.0;
    uint64_t q = seed * 77ull + lane;
    for (int it = 0; it < ITER; it++) {
        if (T == 1) {
#pragma unroll
            for (int b = 0; b < 8; b++) x[b] += *(volatile uint32_t *)((char *)my + a[b]);
        } else if (T == 2 || T == 15) {
#pragma unroll
            for (int b = 0; b < 8; b++) __hip_atomic_fetch_add((uint32_t *)((char *)my + a[b]), x[b] & 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (T == 3 || T == 4) {
#pragma unroll
            for (int b = 0; b < 8; b++) x[b] += __hip_atomic_fetch_add((uint32_t *)((char *)my + a[b]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (T == 5) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[b]) : "s"(seed), "v"(a[b]));
        } else if (T == 6) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x[b]) : "v"(a[b]), "v"(a[(b + 1) & 7]));
        } else if (T == 7) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 8) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[b]) : "v"(a[b]), "v"(a[(b + 1) & 7]));
        } else if (T == 9) {
#pragma unroll
            for (int b = 0; b < 8; b++) asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %1, vcc" : "+v"(x[b]) : "v"(a[b]), "v"(a[(b + 1) & 7]) : "vcc");
        } else if (T == 10) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double t0, t1; uint32_t u;
                asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t0) : "v"(x[r]));
                asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(t1) : "v"(t0), "v"(d1));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t0) : "v"(t1), "v"(d2));
                asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u) : "v"(t0));
                x[r + 4] ^= u;
            }
        } else if (T == 11) {
#pragma unroll
            for (int b = 0; b < 16; b++) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q) : "v"(a[b & 7]));
        } else if (T == 12) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 13) {
#pragma unroll
            for (int b = 0; b < 8; b++) *(volatile uint64_t *)((char *)my + ((a[b] & ~7u))) = ((uint64_t)x[b] << 32) | x[b];
        } else if (T == 14) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x[b]));
        } else if (T == 16) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 17) { // dependent chain of adds
#pragma unroll
            for (int b = 0; b < 16; b++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(a[b & 7]));
        } else if (T == 18) { // dependent f64 chain
#pragma unroll
            for (int b = 0; b < 8; b++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));
        } else if (T == 19) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
} else if (T == 30) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 31) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 32) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 33) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 34) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 35) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 36) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 37) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 38) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 39) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 40) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_not_b32 %0, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 41) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_mov_b32 %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 42) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 43) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 44) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 45) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 46) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 47) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[b]) : "v"(a[b]) : "vcc");
        } else if (T == 48) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : "+v"(x[b]) : "v"(a[b]) : "vcc");
        } else if (T == 49) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 50) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 51) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 52) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_sub_u32 %0, 32, %0" : "+v"(x[b]) : "v"(a[b]));
        } else if (T == 53) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int b = 0; b < 8; b++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[b]) : "v"(a[b]) : "vcc");
} else if (T == 60) {
#pragma unroll
            for (int b = 0; b < 8; b++) { double t; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t) : "v"(x[b])); asm volatile("" :: "v"(t)); }
        } else if (T == 61) {
#pragma unroll
            for (int b = 0; b < 8; b++) { double t; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(d1), "v"(d2)); asm volatile("" :: "v"(t)); }
        } else if (T == 62) {
#pragma unroll
            for (int b = 0; b < 8; b++) { double t; asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(t) : "v"(d1), "v"(d2)); asm volatile("" :: "v"(t)); }
        } else if (T == 63) {
#pragma unroll
            for (int b = 0; b < 8; b++) { uint32_t t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(d1)); asm volatile("" :: "v"(t)); }
        } else if (T == 64) {
#pragma unroll
            for (int b = 0; b < 8; b++) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(x[b]), "v"(a[b]), "v"(q) : "vcc"); asm volatile("" :: "v"(t)); }
        } else if (T == 65) {
#pragma unroll
            for (int b = 0; b < 8; b++) { uint64_t t; asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(t) : "v"(a[b]), "v"(q)); asm volatile("" :: "v"(t)); }
        } else if (T == 66) {
#pragma unroll
            for (int b = 0; b < 8; b++) { float t; asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(t) : "v"(x[b])); asm volatile("" :: "v"(t)); }
        } else if (T == 67) {
#pragma unroll
            for (int b = 0; b < 8; b++) { uint2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a[b] & ~7u)); asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(t)); }
        } else if (T == 68) {
#pragma unroll
            for (int b = 0; b < 8; b++) { asm volatile("ds_write_b64 %0, %1" :: "v"(a[b] & ~7u), "v"(q)); }
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (T == 69) {
#pragma unroll
            for (int b = 0; b < 8; b++) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a[b])); asm volatile("" :: "v"(t)); }
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (T == 70) { // bare barrier
#pragma unroll
            for (int b = 0; b < 8; b++) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (T == 71) { // barrier + LDS hand-off: write my slot, barrier, read the other wave's slot
#pragma unroll
            for (int b = 0; b < 8; b++) {
                my[((wave * 64 + lane) & 255) + 256 * (b & 1)] = x[b];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                x[b] += my[(((wave + 1) * 64 + lane) & 255) + 256 * (b & 1)];
            }
        } else if (T == 72) { // 40 independent VALU between barriers (a "2-symbol phase" of work)
#pragma unroll
            for (int b = 0; b < 8; b++) {
#pragma unroll
                for (int r = 0; r < 5; r++)
#pragma unroll
                    for (int c = 0; c < 8; c++) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[c]) : "s"(seed), "v"(a[c]));
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else if (T == 20) { // SALU
#pragma unroll
            for (int b = 0; b < 16; b++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(seed));
        }
    }
    uint32_t r = (uint32_t)q + (uint32_t)d0 + seed;
#pragma unroll
    for (int b = 0; b < 8; b++) r += x[b];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (blockIdx.x == 500 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int T>
static void run(const char *name, int nper, int threads, size_t lds_bytes, double ghz)
{
    uint32_t *out; unsigned long long *clk, hclk[2];
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
    hipFuncSetAttribute((const void *)k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int grid = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<T><<<grid, threads, lds_bytes>>>(out, 12345, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<grid, threads, lds_bytes>>>(out, 12345, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    double real_ghz = (double)hclk[0] / (double)hclk[1] * 0.1; (void)ghz;
    double cyc = ms * 1e-3 * real_ghz * 1e9 / ITER;
    printf("%-34s w/SIMD=%d %7.3f ms  clk %.2f GHz  %7.1f cyc/iter  %6.2f cyc/instr/wave  %5.2f cyc/instr/SIMD\n", name, threads / 64, ms, real_ghz, cyc, cyc / nper, cyc / nper / (threads / 64));
    hipFree(out);
}

__global__ void kclk(unsigned long long *o) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    while (wall_clock64() - w0 < 2000000ull) { }   // 20 ms at 100 MHz
    o[0] = clock64() - c0; o[1] = wall_clock64() - w0;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned long long *dc, hc[2]; hipMalloc(&dc, 16);
    kclk<<<1, 64>>>(dc); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    double ghz = (double)hc[0] / (double)hc[1] * 0.1;
    printf("measured shader clock (idle chip): %.3f GHz; hipDeviceProp clockRate %.2f GHz\n", ghz, p.clockRate / 1e6);
    printf("device %s clock %.2f GHz CUs %d\n", p.name, ghz, p.multiProcessorCount);
    for (int cfg = 0; cfg < 4; cfg++) {
        int th = cfg == 0 ? 64 : cfg == 1 ? 128 : cfg == 2 ? 192 : 256; size_t l = 40960; // 4 workgroups per CU by LDS
        printf("---- %d wave(s) per workgroup x 4 workgroups per CU = %d waves/SIMD\n", th / 64, th / 64);
        run<0>("empty loop", 1, th, l, ghz);
        run<1>("8x ds_read_b32", 8, th, l, ghz);
        run<2>("8x ds_add_u32", 8, th, l, ghz);
        run<15>("8x ds_add_u32 2-way conflict", 8, th, l, ghz);
        run<3>("8x ds_add_rtn_u32", 8, th, l, ghz);
        run<4>("8x ds_add_rtn_u32 (l,l+32 share)", 8, th, l, ghz);
        run<13>("8x ds_write_b64", 8, th, l, ghz);
        run<16>("16x v_add_u32 indep", 16, th, l, ghz);
        run<17>("16x v_add_u32 dependent", 16, th, l, ghz);
        run<5>("16x v_and_or_b32", 16, th, l, ghz);
        run<6>("16x v_dot2_u32_u16", 16, th, l, ghz);
        run<7>("16x v_pk_lshrrev_b16", 16, th, l, ghz);
        run<8>("16x v_perm_b32", 16, th, l, ghz);
        run<9>("8x v_cmp+v_cndmask", 16, th, l, ghz);
        run<10>("4x (cvt,fma,mul,cvt) f64", 16, th, l, ghz);
        run<18>("8x v_fma_f64 dependent", 8, th, l, ghz);
        run<11>("16x v_lshlrev_b64 dependent", 16, th, l, ghz);
        run<12>("16x v_mul_u32_u24", 16, th, l, ghz);
        run<19>("16x v_mul_lo_u32", 16, th, l, ghz);
        run<14>("16x v_ffbh_u32", 16, th, l, ghz);
        run<30>("16x v_and_b32 (VOP2)", 16, th, l, ghz);
        run<31>("16x v_or_b32 (VOP2)", 16, th, l, ghz);
        run<32>("16x v_xor_b32 (VOP2)", 16, th, l, ghz);
        run<33>("16x v_lshlrev_b32 (VOP2)", 16, th, l, ghz);
        run<34>("16x v_lshrrev_b32 (VOP2)", 16, th, l, ghz);
        run<35>("16x v_sub_u32 (VOP2)", 16, th, l, ghz);
        run<36>("16x v_min_u32 (VOP2)", 16, th, l, ghz);
        run<37>("16x v_bfe_u32 (VOP3)", 16, th, l, ghz);
        run<38>("16x v_lshl_add_u32 (VOP3)", 16, th, l, ghz);
        run<39>("16x v_add3_u32 (VOP3)", 16, th, l, ghz);
        run<40>("16x v_not_b32 (VOP1)", 16, th, l, ghz);
        run<41>("16x v_mov_b32 (VOP1)", 16, th, l, ghz);
        run<42>("16x v_and_b32 SDWA byte sel", 16, th, l, ghz);
        run<43>("16x v_add_u32 + literal", 16, th, l, ghz);
        run<44>("16x v_bfi_b32 (VOP3)", 16, th, l, ghz);
        run<45>("16x v_alignbit_b32 (VOP3)", 16, th, l, ghz);
        run<46>("16x v_mul_hi_u32 (VOP3)", 16, th, l, ghz);
        run<47>("16x v_cndmask_b32 (VOP2, vcc)", 16, th, l, ghz);
        run<48>("16x v_cmp_gt_u32 (VOPC)", 16, th, l, ghz);
        run<49>("16x v_and_or_b32 all-VGPR", 16, th, l, ghz);
        run<50>("16x v_bfm_b32 (VOP3)", 16, th, l, ghz);
        run<51>("16x v_add_u32 DPP", 16, th, l, ghz);
        run<52>("16x v_sub_u32 inline const", 16, th, l, ghz);
        run<53>("16x v_addc_co_u32", 16, th, l, ghz);
        run<60>("8x v_cvt_f64_u32 indep", 8, th, l, ghz);
        run<61>("8x v_mul_f64 indep", 8, th, l, ghz);
        run<62>("8x v_fma_f64 indep", 8, th, l, ghz);
        run<63>("8x v_cvt_u32_f64 indep", 8, th, l, ghz);
        run<64>("8x v_mad_u64_u32 indep", 8, th, l, ghz);
        run<65>("8x v_lshlrev_b64 indep", 8, th, l, ghz);
        run<66>("8x v_cvt_f32_u32 indep", 8, th, l, ghz);
        run<67>("8x ds_read_b64 (latency each)", 8, th, l, ghz);
        run<68>("8x ds_write_b64 throughput", 8, th, l, ghz);
        run<69>("8x ds_read_b32 throughput", 8, th, l, ghz);
        run<70>("8x s_barrier", 8, th, l, ghz);
        run<71>("8x (lds write, barrier, lds read)", 8, th, l, ghz);
        run<72>("8x (40 VALU + barrier)", 8, th, l, ghz);
        run<20>("16x s_add_u32 dependent", 16, th, l, ghz);
    }
    return 0;
}
