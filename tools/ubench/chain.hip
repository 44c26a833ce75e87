// Dependent-chain latencies on a LONE wave (one 64-thread workgroup per CU, gfx950): what a
// decoder lane is made of.  Prints s_memtime ticks (core clock) per instruction of each chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 2048
#define REP16(x) x x x x x x x x x x x x x x x x

template <int T>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    __shared__ uint32_t lds[8192];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 8192; i += 64) lds[i] = (i * 4u * seed) & 0x7FFC; // every dword holds a valid byte address
    __syncthreads();
    uint32_t x = seed + lane, y = seed * 3 + 1, z = lane * 4;
    double   d = 1.0 + lane * 0.001, e = 1.0000001, f = 0.5;
    uint64_t q = seed * 77ull + lane;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < ITER; it++) {
        if (T == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (T == 1) { REP16(asm volatile("v_add3_u32 %0, %0, %1, 1" : "+v"(x) : "v"(y));) }
        if (T == 2) { REP16(asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (T == 3) { REP16(asm volatile("v_cmp_gt_i32 vcc, 0, %0\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(x) : "v"(y), "v"(z) : "vcc");) }
        if (T == 4) { REP16(asm volatile("v_cmp_gt_i32 s[20:21], 0, %0\n\tv_cndmask_b32 %0, %1, %2, s[20:21]" : "+v"(x) : "v"(y), "v"(z) : "s20", "s21");) }
        if (T == 5) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(y));) }
        if (T == 6) { REP16(asm volatile("v_and_b32 %0, 0xffff, %0\n\tv_add3_u32 %0, %0, %1, 1" : "+v"(x) : "v"(y));) }
        if (T == 7) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(e), "v"(f));) }
        if (T == 8) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (T == 9) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (T == 10) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(d));) }
        if (T == 11) { REP16(asm volatile("v_cvt_f64_u32 %1, %0\n\tv_cvt_u32_f64 %0, %1" : "+v"(x), "+v"(d));) }
        if (T == 12) { REP16(asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q));) }
        if (T == 13) { REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) :: "memory");) }
        if (T == 14) { REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(z) : "memory");) }
        if (T == 15) { // 7 reads in flight (4 instructions), then wait: the shape of a descent round
            REP16(asm volatile("ds_read2st64_b32 v[40:41], %0 offset0:2 offset1:4\n\tds_read2st64_b32 v[42:43], %0 offset0:6 offset1:8\n\t"
                               "ds_read2st64_b32 v[44:45], %0 offset0:10 offset1:12\n\tds_read_b32 %0, %0 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                               : "+v"(x) :: "memory", "v40", "v41", "v42", "v43", "v44", "v45");)
        }
        if (T == 16) { // value chain: cvt, add, rcp, fma, fma, mul, mul, cvt, cvt, fma, cmp, addc
            REP16(asm volatile("v_cvt_f64_u32 v[40:41], %0\n\tv_add_f64 v[40:41], v[40:41], 1.0\n\tv_rcp_f64 v[42:43], v[40:41]\n\t"
                               "v_fma_f64 v[44:45], -v[40:41], v[42:43], 1.0\n\tv_fma_f64 v[42:43], v[44:45], v[42:43], v[42:43]\n\t"
                               "v_mul_f64 v[42:43], v[42:43], %1\n\tv_mul_f64 v[42:43], v[42:43], %2\n\tv_cvt_u32_f64 %0, v[42:43]\n\t"
                               "v_cvt_f64_u32 v[42:43], %0\n\tv_fma_f64 v[42:43], -v[42:43], v[40:41], %2\n\tv_cmp_ge_f64 vcc, v[42:43], v[40:41]\n\t"
                               "v_addc_co_u32 %0, vcc, 0, %0, vcc"
                               : "+v"(x) : "v"(e), "v"(f) : "vcc", "v40", "v41", "v42", "v43", "v44", "v45");)
        }
        if (T == 17) { REP16(asm volatile("ds_add_u32 %0, %1" :: "v"(z), "v"(y) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (T == 18) { REP16(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(e), "v"(f));) } // independent f64 issue rate
        if (T == 19) { REP16(asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(z), "v"(y));) } // independent VOP2 issue rate
        if (T == 20) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_cvt_u32_f32 %0, %0" : "+v"(x));) }
        if (T == 21) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(y), "v"(z) : "vcc");) }
        if (T == 22) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 64 + lane] = x + (uint32_t)d + (uint32_t)q;
    if (lane == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

struct Test { const char *name; int per_rep; void (*fn)(uint32_t *, uint32_t, unsigned long long *); };
#define T(n, per, name) {name, per, k<n>}

int main()
{
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 64 * 4); hipMalloc(&clk, 8);
    Test tests[] = {
        T(0, 1, "v_add_u32 dependent"), T(19, 1, "v_add_u32 independent"), T(1, 1, "v_add3_u32 dependent"), T(2, 1, "v_max_u32 dependent"),
        T(3, 2, "v_cmp -> vcc -> v_cndmask (per instr)"), T(4, 2, "v_cmp -> sgpr pair -> v_cndmask (per instr)"),
        T(5, 1, "v_alignbit dependent"), T(6, 2, "v_and + v_add3 (per instr)"),
        T(7, 1, "v_fma_f64 dependent"), T(8, 1, "v_mul_f64 dependent"), T(18, 1, "v_mul_f64 independent"), T(9, 1, "v_add_f64 dependent"),
        T(10, 1, "v_rcp_f64 dependent"), T(11, 2, "v_cvt_f64_u32 + v_cvt_u32_f64 (per instr)"), T(12, 1, "v_lshlrev_b64 dependent"),
        T(13, 1, "ds_read_b32 pointer chase (round trip)"), T(14, 1, "ds_read_b32 + v_add (round trip + 1)"),
        T(15, 1, "descent round: 4 ds_read (7 dwords) + wait"), T(16, 12, "value chain (per instr, 12 instrs)"),
        T(17, 1, "ds_add_u32 (issue, drained per 16)"), T(20, 3, "cvt_f32_u32+rcp_f32+cvt_u32_f32 (per instr)"),
        T(21, 1, "v_mad_u64_u32 dependent"), T(22, 1, "v_mul_hi_u32 dependent"),
    };
    for (auto &t : tests) {
        for (int grid : {1, 1024}) { // lone wave on an idle chip / 4 waves per CU like the real kernel
            t.fn<<<grid, 64>>>(out, 12345, clk);
            hipDeviceSynchronize();
            t.fn<<<grid, 64>>>(out, 12345, clk);
            hipDeviceSynchronize();
            unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("%-52s grid %4d: %7.2f ticks/instr\n", t.name, grid, (double)c / ((double)ITER * 16 * t.per_rep));
        }
    }
    return 0;
}
