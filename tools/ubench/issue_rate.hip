// Micro-benchmark (diagnostic, not part of the library): cycles per instruction of ONE wave on an
// otherwise idle CU, for the instruction kinds the fill kernel is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void k(long long *out, double *sink, int n_iter) {
    double a = threadIdx.x * 1e-3, b = 1.5, c = 2.5, e = 3.5;
    int x = threadIdx.x, y = 7;
    long long t[16];
    int s = 3;
    // 0: dependent v_add_f64
    t[0] = __builtin_readcyclecounter();
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    t[1] = __builtin_readcyclecounter();
    // 1: independent v_add_f64 (4 chains)
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(b));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(e) : "v"(b));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(b) : "v"(b));
        }
    }
    t[2] = __builtin_readcyclecounter();
    // 2: dependent v_add_u32
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    t[3] = __builtin_readcyclecounter();
    // 3: dependent s_add_i32
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
    }
    t[4] = __builtin_readcyclecounter();
    // 4: v_cmp_gt_f64 + 2 v_cndmask (the max-select idiom), dependent
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r)
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_add_f64 %0, %0, %3"
                         : "+v"(a), "+v"(x) : "v"(c), "v"(b), "v"(y) : "vcc");
    }
    t[5] = __builtin_readcyclecounter();
    // 5: alternating SALU / VALU (independent)
    for (int i = 0; i < n_iter; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 2; ++r) {
            asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
        }
    }
    t[6] = __builtin_readcyclecounter();
    t[7] = t[8] = t[9] = t[6];
    if (threadIdx.x == 0) for (int q = 0; q < 9; ++q) out[q + 16 * blockIdx.x] = t[q + 1] - t[q];
    sink[threadIdx.x] = a + b + c + e + x + s;
}
int main() {
    long long *o; double *sink;
    hipMalloc(&o, 16 * 8 * 4); hipMalloc(&sink, 4096);
    const int n_iter = 200;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, o, sink, n_iter);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        const char *names[9] = {"dep v_add_f64", "indep v_add_f64", "dep v_add_u32", "dep s_add_i32", "cmp+2cndmask+add f64 (4 instr)", "alt salu/valu", "taken branch + 2 valu (4 instr)", "exec-mask region (4 instr)", "readfirstlane+cmp+branch+valu (4 instr)"};
        const double per[9] = {1, 1, 1, 1, 4, 1, 4, 4, 4};
        printf("waves in the workgroup: %d\n", waves);
        for (int q = 0; q < 6; ++q) printf("  %-42s %.2f cycles/instr\n", names[q], (double)h[q] / (n_iter * (REP / per[q]) * per[q]));
        fflush(stdout);
    }
    return 0;
}
