// Host <-> device transfer costs that shape the host-pointer ABI (redux_host.hpp):
// pinned H2D / D2H rate, both directions at once, hipHostRegister of caller memory, and how fast
// N CPU threads copy pageable memory into a pinned staging buffer.
// hipcc --offload-arch=gfx950 -O2 -o pcie tools/ubench/pcie.hip -lpthread && ./pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pcopy(char *d, const char *s, size_t n, int T)
{
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) {
        size_t a = n * t / T, b = n * (t + 1) / T;
        th.emplace_back([=] { memcpy(d + a, s + a, b - a); });
    }
    for (auto &x : th) x.join();
}
int main()
{
    const size_t N = 1ull << 30;
    char *d0, *d1, *hp0, *hp1;
    CK(hipMalloc(&d0, N)); CK(hipMalloc(&d1, N));
    double t = now(); CK(hipHostMalloc(&hp0, N, hipHostMallocDefault)); CK(hipHostMalloc(&hp1, N, hipHostMallocDefault));
    printf("hipHostMalloc 2 x 1 GiB: %.1f ms\n", (now() - t) * 1e3);
    char *pg = (char *)malloc(N), *pg2 = (char *)malloc(N);
    memset(pg, 1, N); memset(pg2, 2, N); memset(hp0, 3, N); memset(hp1, 4, N);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 2; rep++) {
        t = now(); CK(hipMemcpyAsync(d0, hp0, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        printf("H2D pinned 1 GiB: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpyAsync(hp1, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
        printf("D2H pinned 1 GiB: %.2f ms = %.1f GB/s\n", (now() - t) * 1e3, N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpyAsync(d0, hp0, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(hp1, d1, N, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        printf("H2D + D2H at once, 1 GiB each: %.2f ms = %.1f GB/s each way\n", (now() - t) * 1e3, N / (now() - t) / 1e9);
    }
    for (size_t piece : {4ull << 20, 16ull << 20, 64ull << 20}) {
        t = now();
        for (size_t o = 0; o < N; o += piece) CK(hipMemcpyAsync(d0 + o, hp0 + o, piece, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("H2D pinned in %zu MiB pieces: %.1f GB/s\n", piece >> 20, N / (now() - t) / 1e9);
    }
    t = now(); CK(hipMemcpy(d0, pg, N, hipMemcpyHostToDevice)); printf("H2D pageable (runtime staging): %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pg2, d1, N, hipMemcpyDeviceToHost)); printf("D2H pageable (runtime staging): %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipHostRegister(pg, N, hipHostRegisterDefault)); double tr = now() - t;
    t = now(); CK(hipMemcpyAsync(d0, pg, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double tc = now() - t;
    t = now(); CK(hipHostUnregister(pg)); double tu = now() - t;
    printf("hipHostRegister 1 GiB: %.1f ms, H2D from it: %.1f GB/s, unregister %.1f ms\n", tr * 1e3, N / tc / 1e9, tu * 1e3);
    for (int T : {1, 2, 4, 8, 12, 16}) {
        pcopy(hp0, pg, N, T);
        t = now(); pcopy(hp0, pg, N, T); double a = now() - t;
        t = now(); pcopy(pg2, hp1, N, T); double b = now() - t;
        printf("CPU copy, %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s\n", T, N / a / 1e9, N / b / 1e9);
    }
    return 0;
}
