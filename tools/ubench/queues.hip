// How many kernels from different HIP streams of one process really run at once (GPU_MAX_HW_QUEUES):
// K streams, one 5 ms single-workgroup spin kernel each; wall time / 5 ms = serialisation factor.
// hipcc --offload-arch=gfx950 -O2 -o queues tools/ubench/queues.hip && ./queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const long long five_ms = 500000; // wall_clock64 ticks at 100 MHz
    spin<<<1, 64>>>(1000, nullptr);
    hipDeviceSynchronize();
    for (int pri = 0; pri < 2; pri++)
        for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
            std::vector<hipStream_t> s(K);
            int lo, hi;
            hipDeviceGetStreamPriorityRange(&lo, &hi);
            for (int i = 0; i < K; i++) {
                if (pri) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, (i & 1) ? hi : lo);
                else hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            }
            hipDeviceSynchronize();
            const double t = now();
            for (int i = 0; i < K; i++) spin<<<1, 64, 0, s[i]>>>(five_ms, nullptr);
            hipDeviceSynchronize();
            printf("%s %2d streams: %.2f ms (%.1f x one kernel)\n", pri ? "mixed priorities" : "default priority", K, (now() - t) * 1e3, (now() - t) * 1e3 / 5.0);
            for (auto x : s) hipStreamDestroy(x);
        }
    return 0;
}
