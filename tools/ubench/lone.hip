// Issue cost of instruction kinds for a LONE wave (one 64-thread workgroup per CU): straight-line
// bodies of 256 instructions in ONE asm statement (no compiler-inserted s_nop), looped.
// "dep" = every instruction depends on the previous one; "ind" = two alternating chains.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define S4(x) x x x x
#define S16(x) S4(S4(x))
#define S64(x) S4(S16(x))
#define S128(x) S64(x) S64(x)
#define S256(x) S4(S64(x))

template <int T>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed, unsigned long long *clk, int iters)
{
    __shared__ uint32_t lds[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += 64) lds[i] = (i * 4u * seed) & 0x7FFC;
    __syncthreads();
    uint32_t x = (seed + threadIdx.x * 4) & 0xFFC, w = threadIdx.x * 4, y = seed * 3 + 1, z = threadIdx.x * 4;
    double   d = 1.0 + threadIdx.x * 0.001, e = 1.0000001;
    uint64_t q = seed * 77ull + threadIdx.x;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#define A(body) asm volatile(body : "+v"(x), "+v"(w), "+v"(d), "+v"(q) : "v"(y), "v"(z), "v"(e) : "memory", "vcc", "s20", "s21", "v40", "v41", "v42", "v43")
        if (T == 0) A(S256("v_add_u32 %0, %0, %4\n\t"));
        if (T == 1) A(S256("v_lshlrev_b32 %0, 1, %0\n\t"));
        if (T == 2) A(S256("v_add3_u32 %0, %0, %4, 1\n\t"));
        if (T == 3) A(S256("v_max_u32 %0, %0, %4\n\t"));
        if (T == 4) A(S128("v_cmp_gt_i32 vcc, 0, %0\n\tv_cndmask_b32 %0, %4, %5, vcc\n\t"));
        if (T == 5) A(S128("v_cmp_gt_i32 s[20:21], 0, %0\n\tv_cndmask_b32 %0, %4, %5, s[20:21]\n\t"));
        if (T == 6) A(S256("v_alignbit_b32 %0, %0, %4, 31\n\t"));
        if (T == 7) A(S256("v_fma_f64 %2, %2, %6, %6\n\t"));
        if (T == 8) A(S256("v_rcp_f64 %2, %2\n\t"));
        if (T == 9) A(S128("v_cvt_f64_u32 %2, %0\n\tv_cvt_u32_f64 %0, %2\n\t"));
        if (T == 10) A(S256("v_lshlrev_b64 %3, 1, %3\n\t"));
        if (T == 11) A(S256("v_ffbh_u32 %0, %0\n\t"));
        if (T == 12) A(S256("v_bfe_u32 %0, %0, 1, 9\n\t"));
        if (T == 13) A(S256("v_and_or_b32 %0, %0, %4, %5\n\t"));
        if (T == 14) A(S256("v_perm_b32 %0, %0, %4, %5\n\t"));
        if (T == 15) A(S256("ds_read_b32 v40, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t");                       // independent reads, same address
        if (T == 16) A(S256("ds_read2st64_b32 v[40:41], %1 offset0:2 offset1:4\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 17) A(S256("ds_add_u32 %1, %4\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 18) A(S256("ds_write_b32 %1, %4\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 19) A(S256("ds_add_rtn_u32 v40, %1, %4\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 20) A(S256("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t"));                              // pointer chase
        if (T == 21) A(S256("s_add_u32 s20, s20, 1\n\t"));
        if (T == 22) A(S128("v_add_u32 %0, %0, %4\n\ts_add_u32 s20, s20, 1\n\t"));                           // VALU/SALU interleave (per pair)
        if (T == 23) A(S128("v_add_u32 %0, %0, %4\n\tds_read_b32 v40, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t");  // VALU/LDS interleave (per pair)
        if (T == 24) A(S128("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\t"));                            // two chains
        if (T == 25) A(S256("ds_read_b64 v[40:41], %1\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 26) A(S256("ds_read_b128 v[40:43], %1\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 27) A(S256("v_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"));
        if (T == 28) A(S256("v_mul_f64 %2, %2, %6\n\t"));
        if (T == 29) A(S256("v_dot2_u32_u16 %0, %4, %5, %0\n\t"));
        if (T == 30) A(S256("ds_write_b128 %1, v[40:43]\n\t") "s_waitcnt lgkmcnt(0)\n\t");
        if (T == 31) A(S256("v_min3_u32 %0, %0, %4, %5\n\t"));
        if (T == 35) A(S256("v_cvt_f64_i32 %2, %0\n\t"));
        if (T == 36) A(S256("v_cvt_f64_u32 %2, %0\n\t"));
        if (T == 37) A(S256("v_cvt_u32_f64 %0, %2\n\t"));
        if (T == 38) A(S256("v_lshrrev_b64 %3, 1, %3\n\t"));
        if (T == 39) A(S256("v_bfm_b32 %0, %0, %4\n\t"));
        if (T == 40) A(S256("v_xnor_b32 %0, %0, %4\n\t"));
        if (T == 41) A(S256("v_subrev_co_u32 %0, s[20:21], 1, %0\n\t"));
        if (T == 42) A(S256("v_add_f64 %2, %2, %6\n\t"));
        if (T == 43) A(S256("v_cvt_i32_f64 %0, %2\n\t"));
        if (T == 44) A(S256("v_mul_u32_u24 %0, %0, %4\n\t"));
        if (T == 45) A(S256("v_mad_u32_u24 %0, %0, %4, %5\n\t"));
        if (T == 46) A(S256("v_lshl_add_u32 %0, %0, 3, %4\n\t"));
        if (T == 32) A(S128("v_add_u32 %0, %0, %4\n\ts_waitcnt lgkmcnt(0)\n\t"));   // per pair: does a satisfied s_waitcnt cost an issue slot?
        if (T == 33) A(S128("v_add_u32 %0, %0, %4\n\ts_nop 0\n\t"));
        if (T == 34) A(S128("v_add_u32 %0, %0, %4\n\ts_waitcnt vmcnt(0)\n\t"));
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x + w + (uint32_t)d + (uint32_t)q;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

struct Test { const char *name; int n; void (*fn)(uint32_t *, uint32_t, unsigned long long *, int); };

int main()
{
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&clk, 8);
    Test tests[] = {
        {"v_add_u32 dep", 256, k<0>}, {"v_add_u32 two chains", 256, k<24>}, {"v_lshlrev_b32 dep", 256, k<1>}, {"v_add3_u32 dep", 256, k<2>},
        {"v_max_u32 dep", 256, k<3>}, {"v_min3_u32 dep", 256, k<31>}, {"v_cmp->vcc->v_cndmask dep (per instr)", 256, k<4>},
        {"v_cmp->sgpr->v_cndmask dep (per instr)", 256, k<5>}, {"v_alignbit dep", 256, k<6>}, {"v_bfe_u32 dep", 256, k<12>},
        {"v_and_or_b32 dep", 256, k<13>}, {"v_perm_b32 dep", 256, k<14>}, {"v_ffbh_u32 dep", 256, k<11>}, {"v_dot2_u32_u16 dep", 256, k<29>},
        {"v_fma_f64 dep", 256, k<7>}, {"v_mul_f64 dep", 256, k<28>}, {"v_rcp_f64 dep", 256, k<8>}, {"cvt_f64_u32+cvt_u32_f64 dep (per instr)", 256, k<9>},
        {"v_lshlrev_b64 dep", 256, k<10>}, {"v_mad_u64_u32 dep", 256, k<27>},
        {"ds_read_b32 x256 then wait", 256, k<15>}, {"ds_read2st64_b32 x256 then wait", 256, k<16>}, {"ds_read_b64 x256 then wait", 256, k<25>},
        {"ds_read_b128 x256 then wait", 256, k<26>}, {"ds_add_u32 x256 then wait", 256, k<17>}, {"ds_add_rtn_u32 x256 then wait", 256, k<19>},
        {"ds_write_b32 x256 then wait", 256, k<18>}, {"ds_write_b128 x256 then wait", 256, k<30>}, {"ds_read_b32 pointer chase", 256, k<20>},
        {"v_cvt_f64_i32", 256, k<35>}, {"v_cvt_f64_u32", 256, k<36>}, {"v_cvt_u32_f64", 256, k<37>}, {"v_cvt_i32_f64", 256, k<43>}, {"v_lshrrev_b64", 256, k<38>},
        {"v_bfm_b32", 256, k<39>}, {"v_xnor_b32", 256, k<40>}, {"v_subrev_co_u32 (sgpr carry)", 256, k<41>}, {"v_add_f64", 256, k<42>},
        {"v_mul_u32_u24", 256, k<44>}, {"v_mad_u32_u24", 256, k<45>}, {"v_lshl_add_u32", 256, k<46>},
        {"v_add + s_waitcnt lgkmcnt(0) (per PAIR)", 128, k<32>}, {"v_add + s_nop 0 (per PAIR)", 128, k<33>}, {"v_add + s_waitcnt vmcnt(0) (per PAIR)", 128, k<34>},
        {"s_add_u32 dep", 256, k<21>}, {"v_add + s_add interleaved (per instr)", 256, k<22>}, {"v_add + ds_read interleaved (per instr)", 256, k<23>},
    };
    for (auto &t : tests) {
        const int iters = 64;
        t.fn<<<1024, 64>>>(out, 12345, clk, iters);
        hipDeviceSynchronize();
        t.fn<<<1024, 64>>>(out, 12345, clk, iters);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.2f cycles/instr\n", t.name, (double)c / ((double)iters * t.n));
    }
    return 0;
}
