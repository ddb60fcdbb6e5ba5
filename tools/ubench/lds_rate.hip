// Micro-benchmark (diagnostic, not part of the library): what the instruction kinds of dp_pipe.hip's hot loop cost a wave
// when 1, 2 or 4 waves of a workgroup (one per SIMD) run the same stream at the same time -- which of them go through a
// unit the SIMDs share.  Every loop is one asm statement (nothing for the compiler to fold), timed with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

extern __shared__ char lds[];

#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1(k) do { long long t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); res[k] = t1_ - t0; } while (0)

#define BEGIN_BLOCK if (mask & (1u << blk)) { T0(); it = n_iter;
#define END_BLOCK(k) T1(k); } ++blk;
#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)

__global__ void k(long long *out, double *sink, int n_iter, double *gbuf, unsigned mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) ((int *)lds)[i] = i;
    __syncthreads();
    long long t0, res[16];
    for (int q = 0; q < 16; ++q) res[q] = 0;
    double a = lane * 1e-3, b = 1.5, c = 2.5, e = 3.5;
    typedef __attribute__((address_space(3))) char lc;
    const unsigned base = (unsigned)(unsigned long long)(lc *)lds;
    const unsigned a24 = base + wave * 8192 + lane * 24;                 // the ring's layout: 24 bytes per lane
    const unsigned a16 = base + 65536 + ((lane * 37) & 255) * 16;        // table-like: 16-byte entries, scattered
    const unsigned a16s = base + 98304 + lane * 16;                      // site records: 16-byte stride
    const unsigned aflag = base + 131072;
    double *g = gbuf + (size_t)blockIdx.x * 65536 + wave * 16384;
    unsigned long long gp = (unsigned long long)(g + lane * 3);
    int it, blk = 0;
    // 0: 32 independent v_add_f64
    BEGIN_BLOCK
    asm volatile("s_mov_b32 s44, %5\n\t1:\n\t" R8("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t")
                 "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b" : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(1.25), "s"(it) : "scc", "s44");
    END_BLOCK(0)
    // 1: 32 s_add_u32 (SALU)
    BEGIN_BLOCK
    { int s1 = 1, s2 = 2;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("s_add_u32 %0, %0, 3\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b" : "+s"(s1), "+s"(s2) : "s"(it) : "scc", "s44"); a += s1 + s2; }
    END_BLOCK(1)
    // 2: 32 ds_read2_b64 (24-byte lane stride) + one wait
    BEGIN_BLOCK
    { double x0, x1;
      asm volatile("s_mov_b32 s44, %3\n\t1:\n\t" R32("ds_read2_b64 v[100:103], %2 offset1:1\n\t") "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\tv_mov_b64 %0, v[100:101]\n\tv_mov_b64 %1, v[102:103]"
                   : "=v"(x0), "=v"(x1) : "v"(a24), "s"(it) : "scc", "s44", "v100", "v101", "v102", "v103", "memory"); a += x0 + x1; }
    END_BLOCK(2)
    // 3: 32 ds_write2_b64 (24-byte lane stride)
    BEGIN_BLOCK
    asm volatile("s_mov_b32 s44, %3\n\t1:\n\t" R32("ds_write2_b64 %0, %1, %2 offset1:1\n\t") "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b"
                 : : "v"(a24), "v"(b), "v"(c), "s"(it) : "scc", "s44", "memory");
    END_BLOCK(3)
    // 4: 32 ds_read_b128 (scattered 16-byte entries)
    BEGIN_BLOCK
    { double x0;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("ds_read_b128 v[100:103], %1\n\t") "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\tv_mov_b64 %0, v[100:101]"
                   : "=v"(x0) : "v"(a16), "s"(it) : "scc", "s44", "v100", "v101", "v102", "v103", "memory"); a += x0; }
    END_BLOCK(4)
    // 5: 32 ds_read_u16 (16-byte lane stride)
    BEGIN_BLOCK
    { int x0;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("ds_read_u16 v100, %1\n\t") "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\tv_mov_b32 %0, v100"
                   : "=v"(x0) : "v"(a16s), "s"(it) : "scc", "s44", "v100", "memory"); a += x0; }
    END_BLOCK(5)
    // 6: 32 ds_read_b32 of one address (a flag)
    BEGIN_BLOCK
    { int x0;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("ds_read_b32 v100, %1\n\t") "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\tv_mov_b32 %0, v100"
                   : "=v"(x0) : "v"(aflag), "s"(it) : "scc", "s44", "v100", "memory"); a += x0; }
    END_BLOCK(6)
    // 7: 32 x (ds_read2_b64 ; 4 independent v_add_f64): LDS reads between arithmetic, one wait per 32
    BEGIN_BLOCK
    { double x0;
      asm volatile("s_mov_b32 s44, %6\n\t1:\n\t" R32("ds_read2_b64 v[100:103], %5 offset1:1\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t")
                   "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\tv_mov_b64 %7, v[100:101]"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "v"(1.25), "v"(a24), "s"(it), "v"(x0) : "scc", "s44", "v100", "v101", "v102", "v103", "memory"); }
    END_BLOCK(7)
    // 8: LDS round trip: ds_read_b32 -> wait -> dependent v_add, 32 times (latency)
    BEGIN_BLOCK
    { int x0 = 0;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("ds_read_b32 v100, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v100\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b"
                   : "+v"(x0) : "v"(aflag), "s"(it) : "scc", "s44", "v100", "memory"); a += x0; }
    END_BLOCK(8)
    // 9: 32 v_mov_b32_dpp wave_shr:1
    BEGIN_BLOCK
    { int x0 = lane, x1 = 0;
      asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b"
                   : "+v"(x0), "+v"(x1) : "s"(it) : "scc", "s44"); a += x1; }
    END_BLOCK(9)
#ifdef WITH_STORE
    // 10: 32 x (global_store_dwordx4 of 24-byte cells ; s_waitcnt vmcnt(16))
    typedef double d2_t __attribute__((ext_vector_type(2)));
    d2_t bc2; bc2.x = b; bc2.y = c;
    BEGIN_BLOCK
    asm volatile("s_mov_b32 s44, %2\n\t1:\n\t" R32("global_store_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(16)\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b"
                 : : "v"(gp), "v"(bc2), "s"(it) : "scc", "s44", "memory");
    END_BLOCK(10)
#endif
#ifdef WITH_SLOAD
    // 11: s_load_dwordx8 -> wait, 32 times (scalar cache hit latency)
    BEGIN_BLOCK
    { unsigned long long sp = (unsigned long long)gbuf;
      asm volatile("s_mov_b32 s44, %1\n\t1:\n\t" R32("s_load_dwordx8 s[36:43], %0, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b"
                   : : "s"(sp), "s"(it) : "scc", "s44", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "memory"); }
    END_BLOCK(11)
#endif
    // 12: 32 x (s_cmp ; s_cbranch not taken ; v_add_f64)
    BEGIN_BLOCK
    asm volatile("s_mov_b32 s44, %1\n\t1:\n\t" R32("s_cmp_eq_u32 %1, 0x7fffffff\n\ts_cbranch_scc1 2f\n\tv_add_f64 %0, %0, %2\n\t") "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 1b\n\t2:"
                 : "+v"(a) : "s"(it), "v"(1.25) : "scc", "s44");
    END_BLOCK(12)
    if (lane == 0) for (int q = 0; q < 13; ++q) out[(blockIdx.x * 8 + wave) * 16 + q] = res[q];
    sink[threadIdx.x] = a + b + c + e;
}

int main(int argc, char **argv) {
    long long *o; double *sink, *gbuf;
    hipMalloc(&o, 16 * 8 * 8 * 4); hipMalloc(&sink, 65536); hipMalloc(&gbuf, 8 * 65536 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    const int n_iter = 100;
    const char *names[13] = {"v_add_f64 (independent)", "s_add_u32", "ds_read2_b64 (24 B stride)", "ds_write2_b64 (24 B stride)", "ds_read_b128 (scattered)",
                             "ds_read_u16 (16 B stride)", "ds_read_b32 (one address)", "ds_read2_b64 + 4 v_add_f64 (5 instr)", "LDS round trip (read, wait, use)",
                             "v_mov_b32_dpp wave_shr:1", "global_store_dwordx4 + vmcnt(16)", "s_load_dwordx8 round trip", "s_cmp + s_cbranch (not taken) + v_add_f64 (3 instr)"};
    const unsigned mask = argc > 1 ? (unsigned)strtoul(argv[1], 0, 0) : 0x1fffu;
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 160000, 0, o, sink, n_iter, gbuf, mask);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        long long h[16 * 8];
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves in the workgroup: %d   (s_memtime ticks per item, wave 0; an item is one instruction unless stated)\n", waves);
        for (int q = 0; q < 13; ++q) printf("  %-52s %.2f\n", names[q], (double)h[q] / (n_iter * 32.0));
        fflush(stdout);
    }
    return 0;
}
