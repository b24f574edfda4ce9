#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    // per-lane address: column (l & 15) of a row-major [.][16] image, lane group l >> 4 starts 4 rows further
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * (l & 15) + (l >> 4) * 64));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[2048], o[256];
    for (int i = 0; i < 2048; ++i) h[i] = i;
    unsigned short *di, *dd; hipMalloc(&di, sizeof h); hipMalloc(&dd, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dd);
    hipMemcpy(o, dd, sizeof o, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf("\n"); }
    return 0;
}
