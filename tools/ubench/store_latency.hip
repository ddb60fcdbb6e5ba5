// Micro-benchmark (diagnostic): how long one wave waits for its global stores to retire (s_waitcnt vmcnt(0)),
// and for a load issued behind them -- the cost of every "drain" in the fill kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long *out, double *buf, int n_iter, int stores_per_iter, size_t stride) {
    double v = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    double *p = buf + threadIdx.x * 3;
    const double *end = buf + ((size_t)1 << 26) - 4096;        // stay inside the buffer whatever the stride
    for (int i = 0; i < n_iter; ++i) {
        for (int s = 0; s < stores_per_iter; ++s) {
            asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
            p += stride;
            if (p >= end) p = buf + threadIdx.x * 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    // store then dependent load (sc1) of an unrelated line
    double acc = 0;
    for (int i = 0; i < n_iter; ++i) {
        asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
        double x;
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(buf + threadIdx.x) : "memory");
        acc += x; p += stride;
        if (p >= end) p = buf + threadIdx.x * 3;
    }
    long long t2 = __builtin_readcyclecounter();
    // load alone
    for (int i = 0; i < n_iter; ++i) {
        double x;
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(buf + threadIdx.x + 64 * (i & 63)) : "memory");
        acc += x;
    }
    long long t3 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
    buf[threadIdx.x] = acc;
}
int main() {
    long long *o; double *buf;
    const size_t n = 1 << 26;
    hipMalloc(&o, 64); hipMalloc(&buf, n * 8);
    hipMemset(buf, 0, n * 8);
    const int n_iter = 2000;
    for (int spi : {1, 3, 8}) {
        for (size_t stride : {(size_t)192, (size_t)8192}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, buf, n_iter, spi, stride);
            hipDeviceSynchronize();
            long long h[3];
            hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
            printf("stores/iter %d stride %zu doubles: drain %.0f ticks/iter, store+load %.0f, load alone %.0f\n", spi, stride,
                   (double)h[0] / n_iter, (double)h[1] / n_iter, (double)h[2] / n_iter);
            fflush(stdout);
        }
    }
    return 0;
}
