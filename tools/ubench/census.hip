// Where do the two waves of each 128-thread / 40 KiB-LDS workgroup land?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) census(uint32_t *out)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that all 1024 are placed concurrently
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000ull) { }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2]     = hwid;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main()
{
    const int G = 1024;
    uint32_t *d; hipMalloc(&d, G * 4 * 4);
    census<<<G, 128, 40960 - 512>>>(d);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(G * 4); hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    // per (xcc, se, sh, cu, simd): how many wave-0 (model) and wave-1 (coder) waves
    std::map<uint32_t, std::pair<int,int>> cnt;
    for (int b = 0; b < G; b++) for (int w = 0; w < 2; w++) {
        uint32_t hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xF;
        uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        uint32_t key = (xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd;
        if (w == 0) cnt[key].first++; else cnt[key].second++;
    }
    std::map<std::pair<int,int>, int> hist;
    for (auto &kv : cnt) hist[kv.second]++;
    printf("distinct (xcc,se,sh,cu,simd) slots used: %zu\n", cnt.size());
    for (auto &kv : hist) printf("  SIMDs holding %d wave-0 and %d wave-1: %d\n", kv.first.first, kv.first.second, kv.second);
    for (int b = 0; b < 6; b++) printf("block %d: w0 hw=%08x xcc=%x | w1 hw=%08x xcc=%x\n", b, h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    return 0;
}
