// Does a kernel launch with ALL 160 KB of a CU's LDS as dynamic shared memory succeed on gfx950?  (tools/ubench, exploration)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x] = s[40959] + s[0];
}
int main() {
    float* o;
    hipMalloc(&o, 4096);
    for (int bytes : {158 * 1024, 160 * 1024 - 256, 160 * 1024}) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), bytes, 0, o);
        hipError_t e2 = hipGetLastError(), e3 = hipDeviceSynchronize();
        float h = 0;
        hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost);
        printf("%d bytes: attr %d launch %d sync %d value %.0f\n", bytes, (int)e1, (int)e2, (int)e3, h);
    }
    return 0;
}
