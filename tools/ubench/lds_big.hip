#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void big_static(double *out) {
    __shared__ double buf[16384];            // 128 KB
    for (int k = threadIdx.x; k < 16384; k += blockDim.x) buf[k] = k;
    __syncthreads();
    double s = 0;
    for (int k = threadIdx.x; k < 16384; k += blockDim.x) s += buf[16383 - k];
    out[threadIdx.x] = s;
}
extern __shared__ double dyn[];
__global__ void big_dynamic(double *out, int n) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) dyn[k] = k;
    __syncthreads();
    double s = 0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) s += dyn[n - 1 - k];
    out[threadIdx.x] = s;
}
int main() {
    double *d; hipMalloc(&d, 256 * 8);
    double h[256];
    hipLaunchKernelGGL(big_static, dim3(1), dim3(256), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("static 128 KB: %s, out[0] = %.0f (want %.0f)\n", hipGetErrorString(e), h[0], 64.0 * 16383 - 256.0 * (63 * 64 / 2));
    const int n = 18432;                      // 144 KB
    hipError_t a = hipFuncSetAttribute((const void *)big_dynamic, hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
    hipLaunchKernelGGL(big_dynamic, dim3(1), dim3(256), n * 8, 0, d, n);
    hipError_t l = hipGetLastError();
    e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("dynamic 144 KB: attr %s, launch %s, sync %s, out[0] = %.0f\n", hipGetErrorString(a), hipGetErrorString(l), hipGetErrorString(e), h[0]);
    return 0;
}
