#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned dyn[];
__global__ void __launch_bounds__(512) big(unsigned *o) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = dyn[511 - threadIdx.x]; }
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    unsigned *o; hipMalloc(&o, 4096);
    for (size_t bytes : {65536ul, 98304ul, 131072ul, 163840ul}) {
        hipError_t e = hipFuncSetAttribute((const void *)big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        big<<<1, 512, bytes>>>(o);
        hipError_t e2 = hipDeviceSynchronize(); hipError_t e3 = hipGetLastError();
        printf("dynamic LDS %zu per 512-thread workgroup: attr %s, run %s %s\n", bytes, hipGetErrorString(e), hipGetErrorString(e2), hipGetErrorString(e3));
    }
    return 0;
}
