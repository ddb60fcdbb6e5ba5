// dp_fb.hip -- forward/backward sum-product over the three alignment matrices, posteriors, path sampling.
//
// Counterpart of the reference's `compute_full_score` pass (src/main/basic_alignment.h:621-625: --full-probability,
// --sample-path, --mpost-posterior-plot-file): the forward sums accumulated beside the Viterbi maxima
// (src/main/viterbi_alignment.cpp:2049-2054, 2078-2083, 2106-2111, 2151-2155, 2182-2186, 2213-2217, 2249-2253, with the
// factors of :1376-1393 and the end corner :1440-1552), the backward pass (:329-341, 740-854, 975-1026, 1571-1662,
// 2259-2305), the posterior (:1029-1034) and sample_new_path (:1193-1322, 1666-2025, 2309-2446).
//
// The reference multiplies raw probabilities, which under- and overflows beyond a few hundred columns; here every
// quantity is a logarithm (product = sum, sum = log-sum-exp in fp64).  Same cell layout as the Viterbi kernels
// (diagonal-major, [cell][X, Y, M]), same anti-diagonal wavefront: forward sweeps d = 0 .. nd-1 reading predecessor
// cells through the bwd edge lists, backward sweeps d = nd-1 .. 0 reading successor cells through the fwd edge lists
// (built on the host from the bwd CSR: a site's fwd list is its outgoing edges in creation order).  Operands in HBM/L2.
// Schedule: an alignment whose diagonals are narrow (a band) is one workgroup with a __syncthreads() per anti-diagonal;
// a wide one (round 4) is cut into 64 x 64 blocks that a grid of one-wave workgroups works through block anti-diagonal by
// block anti-diagonal (see pg_fb_forward_tiled below).  The sums here need to agree with the reference's to 1e-9, not bit
// for bit (log-sum-exp in place of its products), so unlike the Viterbi kernels nothing constrains the order of evaluation.
//
// Reference quirks kept (see oracle/oracle_fb.cpp, which restates the same rules on the CPU): full-probability terms
// use gap_ext for every gap (no end-gap extension) and the plain gap-open probability; edge weights enter matches only;
// the end corner visits the Y-close term of a non-first right edge once per left edge.  Cells outside the tunnel hold
// probability 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pagan_dp.h"
#include "dp_band.h"

struct PgFbJob {
    int Lx, Ly, nd, S;
    double l_ext, l_open, l_ng;              // logs of gap_ext, gap_open, non_gap
    const int *stL, *offL, *srcL; const float *lwL;      // bwd lists (log weights as the CSR carries them)
    const int *stR, *offR, *srcR; const float *lwR;
    const int *foffL, *fdstL; const float *flwL;         // fwd lists
    const int *foffR, *fdstR; const float *flwR;
    const double *ltab;                      // log((double) score[a + b*S])
    const int *imin, *imax; const long long *doff;
    long long cells;
    double *F, *B;                           // [cells][3] log forward / log backward
    int n_init; const long long *init_at; const double *init_val;   // initialise_array_corner_bwd
    double *totals;                          // [2]: log fwd_end, log bwd(M,0,0)
    // the tiled sweeps (round 5: one launch may carry the sweeps of SEVERAL pairs, grid = (most workgroups of any pair, pairs) --
    // pagan_fb_run_batch): the pair's own workgroup count and barrier words
    int groups, groups_b;                    // workgroups of this pair's forward / backward sweep (blockIdx.x beyond: nothing to do)
    int *sync;                               // [64] ints: forward barrier at [0..7], backward at [8..15], diagnostics behind
    int init_dmin;                           // the first cell diagonal that holds a cell of init_at (nd: none)
};

namespace {

#define FB_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (std::getenv("PAGAN_DP_VERBOSE")) std::fprintf(stderr, "pagan_fb: %s: %s\n", #expr, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? PAGAN_E_NOMEM : PAGAN_E_NODEVICE;                            \
        }                                                                                                    \
    } while (0)

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

// The sums' transcendental functions.  A log-sum-exp only ever asks for exp(x) with x <= 0 and for log(z) with
// 1 <= z <= 3 (the largest term factored out, at most two more), and it needs them to ~1e-16 ABSOLUTE: the library's
// exp / log1p -- correct to an ulp over their whole domain, special cases and all -- were ~3,000 of the instructions of a
// cell (the sweep's chain is latency of exactly these).  Here: exp by the usual reduction x = n ln2 + r and a degree-13
// Taylor polynomial in r (|r| <= 0.347: remainder 4e-18), log by z = 2^k m, m in (0.707, 1.415], s = (m - 1) / (m + 1),
// log m = 2 atanh s as a polynomial of degree 11 in s^2 (s^2 <= 0.0295: remainder 6e-19).  fma() explicitly: the file is
// compiled without contraction.
__device__ __forceinline__ double fb_exp_neg(double x) {
    x = fmax(x, -60.0);                                            // (e^-60 = 9e-27 beside a term of 1: nothing; and no -inf - -inf below)
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                             // 1/13!
    p = fma(p, r, 2.08767569878681e-09);  p = fma(p, r, 2.505210838544172e-08); p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.755731922398589e-06); p = fma(p, r, 2.48015873015873e-05);  p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03); p = fma(p, r, 8.333333333333333e-03); p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ double fb_log_1to3(double z) {
    const bool k1 = z > 1.4142135623730951, k2 = z > 2.8284271247461903;
    const double m = k2 ? 0.25 * z : (k1 ? 0.5 * z : z);
    const double kl = k2 ? 1.3862943611198906 : (k1 ? 0.6931471805599453 : 0.0);
    const double den = m + 1.0;
    double y = __builtin_amdgcn_rcp(den);
    y = fma(fma(-den, y, 1.0), y, y);
    y = fma(fma(-den, y, 1.0), y, y);
    const double num = m - 1.0;
    double sq = num * y;
    sq = fma(fma(-den, sq, num), y, sq);                           // one more correction of the quotient itself
    const double w = sq * sq;
    double p = 1.0 / 23.0;
    p = fma(p, w, 1.0 / 21.0); p = fma(p, w, 1.0 / 19.0); p = fma(p, w, 1.0 / 17.0); p = fma(p, w, 1.0 / 15.0);
    p = fma(p, w, 1.0 / 13.0); p = fma(p, w, 1.0 / 11.0); p = fma(p, w, 1.0 / 9.0);  p = fma(p, w, 1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0);  p = fma(p, w, 1.0 / 3.0);  p = fma(p, w, 1.0);
    return fma(2.0 * sq, p, kl);
}

// log(exp(a) + exp(b))
__device__ __forceinline__ double lse(double a, double b) {
    const double hi = fmax(a, b), lo = fmin(a, b);
    if (hi == ninf()) return hi;
    return hi + fb_log_1to3(1.0 + fb_exp_neg(lo - hi));
}

// log(exp(a) + exp(b) + exp(c)): the largest term is factored out, the other two cost an exp each
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double hi = fmax(a, fmax(b, c));
    if (hi == ninf()) return hi;
    // the two that are not the (first) largest
    const double u = a == hi ? b : a, v = (a == hi || b == hi) ? c : b;
    return hi + fb_log_1to3(1.0 + (fb_exp_neg(u - hi) + fb_exp_neg(v - hi)));
}

// The model's log score: from the LDS copy when the table fits (S * S <= 256), else from memory -- as two loads in two address
// spaces behind a uniform branch.  Written as `tab_lds ? M.ltab[k] : J.ltab[k]` the compiler selects the POINTER and emits one
// flat load, and a flat load is a vector memory operation: s_waitcnt vmcnt(0) behind it waited for the wave's stores of the step
// before, on every step of every sweep.
typedef const __attribute__((address_space(1))) double *fb_gcd;
// ALL_LDS: the launch holds only pairs whose table is in LDS, and the load from memory is not even compiled: with both paths in one
// kernel the LDS read still waited -- s_waitcnt vmcnt(0): the wave's stores of the step before -- because its destination register
// is the memory load's too, and the compiler cannot know that load was never issued.
template <bool ALL_LDS>
__device__ __forceinline__ double fb_score(bool tab_lds, const double *lds_tab, const double *mem_tab, int a, int b, int S) {
    if (ALL_LDS || tab_lds) return lds_tab[a + b * S];
    return ((fb_gcd)(unsigned long long)mem_tab)[a + (long long)b * S];
}

__device__ __forceinline__ long long cell_at(const PgFbJob &J, int p, int q) {
    if (p < 0 || q < 0 || p >= J.Lx || q >= J.Ly) return -1;
    const int d = p + q;
    const int mn = J.imin[d], mx = J.imax[d];
    return (p >= mn && p <= mx) ? J.doff[d] + (p - mn) : -1;
}

__device__ __forceinline__ double rd(const double *A, long long at, int s) { return at >= 0 ? A[3 * at + s] : ninf(); }

__global__ __launch_bounds__(1024) void pg_fb_forward(const PgFbJob *jobs) {
    const PgFbJob J = jobs[blockIdx.x];
    for (int d = 0; d < J.nd; ++d) {
        const int mn = J.imin[d], mx = J.imax[d];
        const long long off = J.doff[d];
        for (int i = mn + (int)threadIdx.x; i <= mx; i += (int)blockDim.x) {
            const int j = d - i;
            double fx = ninf(), fy = ninf(), fm = ninf();
            if (i == 0 && j == 0) {
                fm = 0.0;                                                          // fwd_score = 1, VA:730
            } else {
                if (i > 0)
                    for (int k = J.offL[i]; k < J.offL[i + 1]; ++k) {
                        const long long at = cell_at(J, J.srcL[k], j);
                        fx = lse(fx, lse3(rd(J.F, at, 0) + J.l_ext,                 // VA:2153
                                          rd(J.F, at, 1) + J.l_open,                // VA:2184 (gap_close = 1)
                                          rd(J.F, at, 2) + J.l_ng + J.l_open));     // VA:2215
                    }
                if (j > 0)
                    for (int k = J.offR[j]; k < J.offR[j + 1]; ++k) {
                        const long long at = cell_at(J, i, J.srcR[k]);
                        fy = lse(fy, lse3(rd(J.F, at, 1) + J.l_ext, rd(J.F, at, 0) + J.l_open, rd(J.F, at, 2) + J.l_ng + J.l_open));
                    }
                if (i > 0 && j > 0) {
                    const double sc = J.ltab[J.stL[i] + (long long)J.stR[j] * J.S];
                    const double mm = J.l_ng + J.l_ng + sc, xm = J.l_ng + sc;      // VA:1383-1391
                    for (int k1 = J.offL[i]; k1 < J.offL[i + 1]; ++k1)
                        for (int k2 = J.offR[j]; k2 < J.offR[j + 1]; ++k2) {
                            const long long at = cell_at(J, J.srcL[k1], J.srcR[k2]);
                            const double w = (double)J.lwL[k1] + (double)J.lwR[k2];
                            fm = lse(fm, lse3(rd(J.F, at, 2) + mm + w,              // VA:2051
                                              rd(J.F, at, 0) + xm + w,              // VA:2080
                                              rd(J.F, at, 1) + xm + w));            // VA:2108
                        }
                }
            }
            double *o = J.F + 3 * (off + (i - mn));
            o[0] = fx; o[1] = fy; o[2] = fm;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // end corner, VA:1440-1552
        double acc = ninf();
        const int l0 = J.offL[J.Lx], l1 = J.offL[J.Lx + 1], r0 = J.offR[J.Ly], r1 = J.offR[J.Ly + 1];
        auto mt = [&](int k1, int k2) { return rd(J.F, cell_at(J, J.srcL[k1], J.srcR[k2]), 2) + J.l_ng + (double)J.lwL[k1] + (double)J.lwR[k2]; };
        auto xc = [&](int k1) { return rd(J.F, cell_at(J, J.srcL[k1], J.Ly - 1), 0); };
        auto yc = [&](int k2) { return rd(J.F, cell_at(J, J.Lx - 1, J.srcR[k2]), 1); };
        if (l1 > l0 && r1 > r0) {
            acc = lse(acc, mt(l0, r0)); acc = lse(acc, xc(l0)); acc = lse(acc, yc(r0));
            for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(l0, k2)); acc = lse(acc, yc(k2)); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) {
                acc = lse(acc, mt(k1, r0)); acc = lse(acc, xc(k1));
                for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(k1, k2)); acc = lse(acc, yc(k2)); }
            }
        }
        J.totals[0] = acc;
    }
}

__global__ __launch_bounds__(1024) void pg_fb_backward(const PgFbJob *jobs) {
    const PgFbJob J = jobs[blockIdx.x];
    for (long long k = threadIdx.x; k < 3 * J.cells; k += blockDim.x) J.B[k] = ninf();
    __syncthreads();
    for (int k = threadIdx.x; k < J.n_init; k += blockDim.x) J.B[J.init_at[k]] = J.init_val[k];   // VA:740-854
    __syncthreads();
    for (int d = J.nd - 1; d >= 0; --d) {
        const int mn = J.imin[d], mx = J.imax[d];
        const long long off = J.doff[d];
        for (int i = mn + (int)threadIdx.x; i <= mx; i += (int)blockDim.x) {
            const int j = d - i;
            double *o = J.B + 3 * (off + (i - mn));
            double bx = o[0], by = o[1], bm = o[2];
            for (int k = J.foffL[i]; k < J.foffL[i + 1]; ++k) {                    // iterate_fwd_edges_for_gap, left site
                const int t = J.fdstL[k];
                if (t >= J.Lx) continue;                                           // VA:1580
                const double nx = rd(J.B, cell_at(J, t, j), 0);
                bx = lse(bx, nx + J.l_ext); by = lse(by, nx + J.l_open); bm = lse(bm, nx + J.l_ng + J.l_open);   // VA:2281-2303
            }
            for (int k = J.foffR[j]; k < J.foffR[j + 1]; ++k) {
                const int t = J.fdstR[k];
                if (t >= J.Ly) continue;
                const double ny = rd(J.B, cell_at(J, i, t), 1);
                by = lse(by, ny + J.l_ext); bx = lse(bx, ny + J.l_open); bm = lse(bm, ny + J.l_ng + J.l_open);
            }
            for (int k1 = J.foffL[i]; k1 < J.foffL[i + 1]; ++k1)                    // iterate_fwd_edges_for_match
                for (int k2 = J.foffR[j]; k2 < J.foffR[j + 1]; ++k2) {
                    const int t = J.fdstL[k1], u = J.fdstR[k2];
                    if (t >= J.Lx || u >= J.Ly) continue;
                    const double thru = rd(J.B, cell_at(J, t, u), 2) + J.ltab[J.stL[t] + (long long)J.stR[u] * J.S] +
                                        (double)J.flwL[k1] + (double)J.flwR[k2];   // VA:2269-2271
                    bx = lse(bx, thru + J.l_ng); by = lse(by, thru + J.l_ng); bm = lse(bm, thru + J.l_ng + J.l_ng);
                }
            o[0] = bx; o[1] = by; o[2] = bm;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) J.totals[1] = rd(J.B, cell_at(J, 0, 0), 2);
}

// ---- long tunnels between plain sequences: one workgroup, lane = row mod B, the last diagonals in an LDS ring ----
// (round 5, last session.)  The one-workgroup sweeps above pay a full __syncthreads() -- the wave's stores acknowledged -- and
// round trips to L2 for operands and list entries on every cell diagonal (2.7 us); the block schedule sweeps every cell diagonal
// of a tunnel twice, with 25 of 64 lanes at work, behind a counter barrier.  When every site of both graphs has exactly one edge,
// from the site before it (leaf sequences: what a tunnel of 2 x 100 kb is made of), a cell reads (i-1, j), (i, j-1) of the
// diagonal before and (i-1, j-1) of the one before that, and nothing else.  Thread x owns the row i = x (mod B) of the current
// diagonal (B = blockDim >= the widest diagonal: at most one such row), the last three diagonals live in LDS ([slot][state][x]; an
// idle thread writes -inf), an operand is a read of the neighbouring lane's slot -- valid if that row was in the band on that
// diagonal, which two compares against the diagonal's interval tell --, the rows' and the columns' records (state, log weight of
// the one edge) and the diagonals' intervals sit in LDS windows refilled every FB_RG_REFILL diagonals, and the step ends in
// s_waitcnt lgkmcnt(0) + s_barrier: the scores go to memory unwaited-for.  Graph pairs (any site with another edge
// list) and diagonals wider than 512 cells keep the block schedule.
#define FB_RG_COLS 1024          // columns / rows in the LDS windows (>= B + 2 * FB_RG_REFILL) up to B = 512; 2,048 for B = 1,024
#define FB_RG_REFILL 256
#define FB_RG_MAXB 1024         // rows of the widest diagonal a ring sweep takes
#define FB_RG_INIT 32           // assignments of initialise_array_corner_bwd a ring sweep holds (a pair with more takes the block schedule)
#define FB_RG_THREADS 1024       // B = 64 ... 256: three threads a row, one per state; B = 512: two (X and Y; M)
// Who a thread is.  A cell's three states are three independent log-sum-exps of the same size -- the step's latency is one wave's
// way through them --, so up to B = 256 a row has THREE threads, one per state (NSPLIT = 3, blockDim = 3 B), and with B = 512 two
// (NSPLIT = 2: X and Y in one, M in the other; 1,024 threads is what a workgroup holds): wave w works on part w % NSPLIT of the
// rows 64 (w / NSPLIT) ..., which puts the waves of a block of rows on different SIMDs (waves go to the SIMDs round-robin).
// NSPLIT is a template argument: with the parts as run-time flags the one-thread-per-row sweeps lost a third of their speed.
// x: the thread's row modulo B; do_x / do_y / do_m: its states; tid / nt: for the staging loops, which all threads share.
#define FB_RING_THREADS                                                                                       \
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;                                                   \
    const int B = nt / NSPLIT;                                                                                \
    const int part = (tid >> 6) % NSPLIT;                                                                     \
    const int x = ((tid >> 6) / NSPLIT) * 64 + (tid & 63);                                                    \
    const bool do_x = NSPLIT == 1 || part == 0, do_y = NSPLIT == 1 || (NSPLIT == 2 ? part == 0 : part == 1), \
               do_m = NSPLIT == 1 || part == NSPLIT - 1;
typedef __attribute__((address_space(1))) double *fb_gd;

__device__ __forceinline__ void fb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Nothing on a step's path is a load from memory (a vector load would wait behind the wave's stores -- memory operations
// complete in order --, and the compiler cannot make scalar loads of arrays it has only a pointer from memory to): the diagonals'
// intervals and offsets, the rows' and the columns' records are staged every FB_RG_REFILL diagonals for the next FB_RG_REFILL.
// MAXB = 512: 59 KB.  MAXB = 1024 (one thread a row; round 5: a workgroup's LDS is not bounded by 64 KB on gfx950 --
// tools/ubench/lds_big.hip): a ring of 74 KB, windows of 2,048 rows / columns, 113 KB in all.
template <int MAXB>
struct FbRingSmem {
    static constexpr int COLS = MAXB > 512 ? 2048 : FB_RG_COLS;
    double ring[3][3][MAXB];           // [diagonal slot][X, Y, M][thread]
    int c_st[COLS]; float c_lw[COLS];      // column j at j % FB_RG_COLS: state, log weight of the edge (j-1) -> j
    int r_st[COLS]; float r_lw[COLS];      // row i the same way
    int dmin[FB_RG_REFILL], dmax[FB_RG_REFILL]; long long doff[FB_RG_REFILL];     // diagonal d at d % FB_RG_REFILL
    double ltab[256];
    long long i_at[FB_RG_INIT]; double i_val[FB_RG_INIT];                         // initialise_array_corner_bwd's assignments (backward sweep)
};

template <bool ALL_LDS, int NSPLIT, int MAXB>
__global__ __launch_bounds__(FB_RG_THREADS) void pg_fb_forward_ring(const PgFbJob *jobs) {
    __shared__ FbRingSmem<MAXB> M;
    constexpr int CMASK = FbRingSmem<MAXB>::COLS - 1;
    const PgFbJob J = jobs[blockIdx.x];
    FB_RING_THREADS
    const int xm1 = (x - 1) & (B - 1);
    const double NI = ninf();
    const bool tab_lds = J.S * J.S <= 256;
    if (tab_lds) for (int k = tid; k < J.S * J.S; k += nt) M.ltab[k] = J.ltab[k];
    for (int q = 0; q < 9; ++q) (&M.ring[0][0][0])[q * MAXB + x] = NI;
    int mn1 = 0, mx1 = -1, mn2 = 0, mx2 = -1;            // the intervals of the diagonals d-1, d-2
    int s0 = 0, s1 = 2, s2 = 1;                          // ring slots of d, d-1, d-2
    int cols_hi = -1, rows_hi = -1;                      // columns <= cols_hi, rows <= rows_hi are in the windows
    const fb_gd F = (fb_gd)(unsigned long long)J.F;
    __syncthreads();
    for (int d = 0; d < J.nd; ++d) {
        if ((d & (FB_RG_REFILL - 1)) == 0) {
            // the next FB_RG_REFILL diagonals; their columns and rows: the largest of either grows by at most one a diagonal
            for (int k = tid; k < FB_RG_REFILL; k += nt) {
                const int dd = d + k;
                const bool in = dd < J.nd;
                M.dmin[k] = in ? J.imin[dd] : 0; M.dmax[k] = in ? J.imax[dd] : -1; M.doff[k] = in ? J.doff[dd] : 0;
            }
            const int mn_ = J.imin[d], mx_ = J.imax[d];
            const int want_c = min(J.Ly - 1, d - mn_ + FB_RG_REFILL), want_r = min(J.Lx - 1, mx_ + FB_RG_REFILL);
            for (int j = cols_hi + 1 + tid; j <= want_c; j += nt) {
                M.c_st[j & CMASK] = J.stR[j];
                M.c_lw[j & CMASK] = j > 0 ? J.lwR[j - 1] : 0.0f;       // (plain graph: site j's one edge is list entry j - 1)
            }
            for (int i = rows_hi + 1 + tid; i <= want_r; i += nt) {
                M.r_st[i & CMASK] = J.stL[i];
                M.r_lw[i & CMASK] = i > 0 ? J.lwL[i - 1] : 0.0f;
            }
            cols_hi = max(cols_hi, want_c); rows_hi = max(rows_hi, want_r);
            fb_lds_barrier();                                      // (the staged values went through registers into LDS: the loads are done)
        }
        const int mn = M.dmin[d & (FB_RG_REFILL - 1)], mx = M.dmax[d & (FB_RG_REFILL - 1)];
        const int i = mn + ((x - mn) & (B - 1));
        const bool active = i <= mx;
        double fx = NI, fy = NI, fm = NI;
        if (active) {
            const int j = d - i;
            if (i == 0 && j == 0) {
                fm = 0.0;                                                          // fwd_score = 1, VA:730
            } else {
                if (do_x && i > 0 && i - 1 >= mn1 && i - 1 <= mx1) {               // (i-1, j): VA:2153, 2184, 2215
                    const double ax = M.ring[s1][0][xm1], ay = M.ring[s1][1][xm1], am = M.ring[s1][2][xm1];
                    fx = lse3(ax + J.l_ext, ay + J.l_open, am + J.l_ng + J.l_open);
                }
                if (do_y && j > 0 && i >= mn1 && i <= mx1) {                       // (i, j-1)
                    const double px = M.ring[s1][0][x], py = M.ring[s1][1][x], pm = M.ring[s1][2][x];
                    fy = lse3(py + J.l_ext, px + J.l_open, pm + J.l_ng + J.l_open);
                }
                if (do_m && i > 0 && j > 0 && i - 1 >= mn2 && i - 1 <= mx2) {      // (i-1, j-1): VA:2051, 2080, 2108
                    const double cx = M.ring[s2][0][xm1], cy = M.ring[s2][1][xm1], cm = M.ring[s2][2][xm1];
                    const int str = M.r_st[i & CMASK], stc = M.c_st[j & CMASK];
                    const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, str, stc, J.S);
                    const double w = (double)M.r_lw[i & CMASK] + (double)M.c_lw[j & CMASK];
                    const double mm = J.l_ng + J.l_ng + sc + w, xm = J.l_ng + sc + w;   // VA:1383-1391
                    fm = lse3(cm + mm, cx + xm, cy + xm);
                }
            }
            const fb_gd o = F + 3 * (M.doff[d & (FB_RG_REFILL - 1)] + (i - mn));
            if (do_x) o[0] = fx;
            if (do_y) o[1] = fy;
            if (do_m) o[2] = fm;
        }
        if (do_x) M.ring[s0][0][x] = fx;
        if (do_y) M.ring[s0][1][x] = fy;
        if (do_m) M.ring[s0][2][x] = fm;
        mn2 = mn1; mx2 = mx1; mn1 = mn; mx1 = mx;
        { const int t = s2; s2 = s1; s1 = s0; s0 = t; }
        fb_lds_barrier();
    }
    __syncthreads();                                                               // (the end corner reads the scores from memory)
    if (threadIdx.x == 0) {
        // end corner, VA:1440-1552
        double acc = ninf();
        const int l0 = J.offL[J.Lx], l1 = J.offL[J.Lx + 1], r0 = J.offR[J.Ly], r1 = J.offR[J.Ly + 1];
        auto mt = [&](int k1, int k2) { return rd(J.F, cell_at(J, J.srcL[k1], J.srcR[k2]), 2) + J.l_ng + (double)J.lwL[k1] + (double)J.lwR[k2]; };
        auto xc = [&](int k1) { return rd(J.F, cell_at(J, J.srcL[k1], J.Ly - 1), 0); };
        auto yc = [&](int k2) { return rd(J.F, cell_at(J, J.Lx - 1, J.srcR[k2]), 1); };
        if (l1 > l0 && r1 > r0) {
            acc = lse(acc, mt(l0, r0)); acc = lse(acc, xc(l0)); acc = lse(acc, yc(r0));
            for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(l0, k2)); acc = lse(acc, yc(k2)); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) {
                acc = lse(acc, mt(k1, r0)); acc = lse(acc, xc(k1));
                for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(k1, k2)); acc = lse(acc, yc(k2)); }
            }
        }
        J.totals[0] = acc;
    }
}

// Mirrored: diagonals nd-1 .. 0; a cell reads (i+1, j), (i, j+1) of the diagonal after it and (i+1, j+1) of the one after that;
// the windows hold row t = i+1's and column u = j+1's records (state, weight of the edge from the site before).
// What initialise_array_corner_bwd assigns (a handful of cells on the last diagonals, VA:740-854) is laid over the -inf a cell
// starts from on the diagonals >= init_dmin, and there the sums are taken one by one as pg_fb_backward takes them.
template <bool ALL_LDS, int NSPLIT, int MAXB>
__global__ __launch_bounds__(FB_RG_THREADS) void pg_fb_backward_ring(const PgFbJob *jobs) {
    __shared__ FbRingSmem<MAXB> M;
    constexpr int CMASK = FbRingSmem<MAXB>::COLS - 1;
    const PgFbJob J = jobs[blockIdx.x];
    FB_RING_THREADS
    const int xp1 = (x + 1) & (B - 1);
    const double NI = ninf();
    const bool tab_lds = J.S * J.S <= 256;
    if (tab_lds) for (int k = tid; k < J.S * J.S; k += nt) M.ltab[k] = J.ltab[k];
    for (int q = 0; q < 9; ++q) (&M.ring[0][0][0])[q * MAXB + x] = NI;
    int mn1 = 0, mx1 = -1, mn2 = 0, mx2 = -1;            // the intervals of the diagonals d+1, d+2
    int s0 = 0, s1 = 2, s2 = 1;
    for (int k = tid; k < J.n_init && k < FB_RG_INIT; k += nt) { M.i_at[k] = J.init_at[k]; M.i_val[k] = J.init_val[k]; }   // (in LDS: no load from memory inside the sweep's loop but the staging's)
    int cols_lo = J.Ly, rows_lo = J.Lx;                  // columns >= cols_lo, rows >= rows_lo are in the windows
    const fb_gd Bm = (fb_gd)(unsigned long long)J.B;
    __syncthreads();
    for (int d = J.nd - 1; d >= 0; --d) {
        if (d == J.nd - 1 || (d & (FB_RG_REFILL - 1)) == FB_RG_REFILL - 1) {
            // the diagonals down to the next multiple of FB_RG_REFILL; columns u = j + 1 and rows t = i + 1: the smallest of either
            // falls by at most one a diagonal
            for (int k = tid; k < FB_RG_REFILL; k += nt) {
                const int dd = (d & ~(FB_RG_REFILL - 1)) + k;
                if (dd <= d) { M.dmin[k] = J.imin[dd]; M.dmax[k] = J.imax[dd]; M.doff[k] = J.doff[dd]; }
            }
            const int mn_ = J.imin[d], mx_ = J.imax[d];
            const int want_c = max(0, d - mx_ + 1 - FB_RG_REFILL), want_r = max(0, mn_ + 1 - FB_RG_REFILL);
            for (int u = cols_lo - 1 - tid; u >= want_c; u -= nt) {
                M.c_st[u & CMASK] = J.stR[u];
                M.c_lw[u & CMASK] = u > 0 ? J.lwR[u - 1] : 0.0f;
            }
            for (int t = rows_lo - 1 - tid; t >= want_r; t -= nt) {
                M.r_st[t & CMASK] = J.stL[t];
                M.r_lw[t & CMASK] = t > 0 ? J.lwL[t - 1] : 0.0f;
            }
            cols_lo = min(cols_lo, want_c); rows_lo = min(rows_lo, want_r);
            fb_lds_barrier();                                      // (the staged values went through registers into LDS: the loads are done)
        }
        const int mn = M.dmin[d & (FB_RG_REFILL - 1)], mx = M.dmax[d & (FB_RG_REFILL - 1)];
        const int i = mn + ((x - mn) & (B - 1));
        const bool active = i <= mx;
        double bx = NI, by = NI, bm = NI;
        if (active) {
            const int j = d - i;
            const bool has_a = i + 1 < J.Lx && i + 1 >= mn1 && i + 1 <= mx1;       // (i+1, j)
            const bool has_p = j + 1 < J.Ly && i >= mn1 && i <= mx1;               // (i, j+1)
            const bool has_c = i + 1 < J.Lx && j + 1 < J.Ly && i + 1 >= mn2 && i + 1 <= mx2;   // (i+1, j+1)
            const double nx = has_a ? M.ring[s1][0][xp1] : NI, ny = has_p ? M.ring[s1][1][x] : NI;
            double thru = NI;
            if (i + 1 < J.Lx && j + 1 < J.Ly) {
                const int str = M.r_st[(i + 1) & CMASK], stc = M.c_st[(j + 1) & CMASK];
                const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, str, stc, J.S);
                thru = (has_c ? M.ring[s2][2][xp1] : NI) + sc + (double)M.r_lw[(i + 1) & CMASK] + (double)M.c_lw[(j + 1) & CMASK];   // VA:2269-2271
            }
            const long long at = M.doff[d & (FB_RG_REFILL - 1)] + (i - mn);
            if (d >= J.init_dmin) {
                for (int k = 0; k < J.n_init; ++k) {
                    const long long w = M.i_at[k] - 3 * at;
                    if (w == 0) bx = M.i_val[k]; else if (w == 1) by = M.i_val[k]; else if (w == 2) bm = M.i_val[k];
                }
                // (one by one, as pg_fb_backward takes them; a state's sums do not look at the other states')
                if (i + 1 < J.Lx) { if (do_x) bx = lse(bx, nx + J.l_ext); if (do_y) by = lse(by, nx + J.l_open); if (do_m) bm = lse(bm, nx + J.l_ng + J.l_open); }   // VA:2281-2303
                if (j + 1 < J.Ly) { if (do_y) by = lse(by, ny + J.l_ext); if (do_x) bx = lse(bx, ny + J.l_open); if (do_m) bm = lse(bm, ny + J.l_ng + J.l_open); }
                if (i + 1 < J.Lx && j + 1 < J.Ly) { if (do_x) bx = lse(bx, thru + J.l_ng); if (do_y) by = lse(by, thru + J.l_ng); if (do_m) bm = lse(bm, thru + J.l_ng + J.l_ng); }
            } else {
                if (do_x) bx = lse3(nx + J.l_ext, ny + J.l_open, thru + J.l_ng);
                if (do_y) by = lse3(nx + J.l_open, ny + J.l_ext, thru + J.l_ng);
                if (do_m) bm = lse3(nx + J.l_ng + J.l_open, ny + J.l_ng + J.l_open, thru + J.l_ng + J.l_ng);
            }
            const fb_gd o = Bm + 3 * at;
            if (do_x) o[0] = bx;
            if (do_y) o[1] = by;
            if (do_m) o[2] = bm;
        }
        if (do_x) M.ring[s0][0][x] = bx;
        if (do_y) M.ring[s0][1][x] = by;
        if (do_m) M.ring[s0][2][x] = bm;
        mn2 = mn1; mx2 = mx1; mn1 = mn; mx1 = mx;
        { const int t = s2; s2 = s1; s1 = s0; s0 = t; }
        fb_lds_barrier();
    }
    __syncthreads();
    if (threadIdx.x == 0) J.totals[1] = rd(J.B, cell_at(J, 0, 0), 2);
}
#undef FB_RING_THREADS

// ---- wide alignments: 64 x 64 blocks on a block-anti-diagonal schedule ----
// Block (a, b) needs blocks (a', b') <= (a, b) only (bwd edges point to earlier sites), so the blocks of one block
// anti-diagonal are independent: workgroup g of the sweep's grid takes the blocks g, g + G, ... of block diagonal t, one
// wave each, and all workgroups meet at a counter barrier before t + 1 -- Lx/64 + Ly/64 barriers per sweep where a barrier
// per cell diagonal cost more than the cells (14 us per diagonal measured: written-through stores, loads from beyond the
// L2).  Inside a block the wave sweeps the block's own 127 anti-diagonals, lane r on row i0 + r: the last FB_RING of them
// live in LDS, and so do the row above / column left of the block (forward) or the row below / column right of it
// (backward); anything else -- an edge that reaches further -- is a load from L2/HBM.  The wave's own stores go out
// unwaited-for; only a read of a cell of its own block that has left the ring waits for them.
// Barrier (MI355X_MICROARCH.md, "Valid forms", producer / consumer with fences): the wave's s_waitcnt vmcnt(0), the
// workgroup barrier, lane 0's agent release + wait, its relaxed agent add; lane 0 polls (relaxed, agent), then agent
// acquire + wait, workgroup barrier, plain loads.
#define FB_T 64
#define FB_RING 12
#define FB_H 8                   // halo depth: rows above / columns left of a block (forward), below / right of it (backward) kept in LDS
#define FB_MAX_GROUPS 64
#define FB_SPIN_LIMIT (1 << 22)
typedef __attribute__((address_space(1))) int *fb_gi;

struct FbSmem {
    double ring[FB_RING][FB_T][3];
    // forward: ha[u][k] = cell (i0 - FB_H + u, j0 - FB_H + k), the FB_H rows above with the corner; hb[k][u] = cell (i0 + k, j0 - FB_H + u),
    // the FB_H columns to the left.  backward: ha[u][k] = cell (i0 + 64 + u, j0 + k), the rows below with the corner at k >= 64;
    // hb[k][u] = cell (i0 + k, j0 + 64 + u), the columns to the right
    double ha[FB_H][FB_T + FB_H][3];
    double hb[FB_T][FB_H][3];
    // what a step reads besides cells, staged per block: nothing of the common path is a load from memory (a load would
    // queue behind the wave's stores in flight: vector memory operations complete in order)
    int dmin[2 * FB_T], dmax[2 * FB_T];      // band interval of the block's diagonals
    long long doff[2 * FB_T];
    // columns j0 + k: first list entry, entries, the first two entries' other ends and log weights, state (backward: one more column)
    int c_off[FB_T + 1], c_n[FB_T + 1], c_e0[FB_T + 1], c_e1[FB_T + 1], c_st[FB_T + 1];
    float c_lw0[FB_T + 1], c_lw1[FB_T + 1];
    double ltab[256];                // the model's log scores when S * S <= 256
};

// sync[0] arrivals, sync[1] nonzero: a workgroup gave up (a logic error becomes a status instead of a hung GPU).
__device__ __forceinline__ bool fb_barrier(int *sync, int target) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        fb_gi g = (fb_gi)(unsigned long long)sync;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, good = 1;
        while (__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255) == 0 && (spins > FB_SPIN_LIMIT || __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(g + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// does block (a, b) hold a cell of the band?
__device__ __forceinline__ bool fb_block_live(const PgFbJob &J, int i0, int j0, int r) {
    bool any = false;
    for (int u = 0; u < 2; ++u) {
        const int d = i0 + j0 + r + 64 * u;
        if (d >= J.nd || r + 64 * u > 2 * FB_T - 2) continue;
        const int lo = max(max(J.imin[d], i0), d - (j0 + FB_T - 1)), hi = min(min(J.imax[d], i0 + FB_T - 1), d - j0);
        any = any || lo <= hi;
    }
    return __builtin_amdgcn_ballot_w64(any) != 0;
}

// The block rows a <= a' <= b that can hold a cell of block anti-diagonal t: the rows of the band's cells on its 127 cell
// diagonals (exact for any band: a minimum and a maximum over the diagonals' intervals).  A tunnel around the main diagonal of
// 2 x 100 kb has 1,563 blocks on its longest block anti-diagonal and a cell in two or three of them.
__device__ __forceinline__ void fb_live_rows(const PgFbJob &J, int t, int r, int &a_first, int &a_last) {
    int lo = 0x7fffffff, hi = -1;
    for (int u = 0; u < 2; ++u) {
        const int d = FB_T * t + r + 64 * u;
        if (d < J.nd && r + 64 * u <= 2 * FB_T - 2) {
            const int mn = J.imin[d], mx = J.imax[d];
            if (mx >= mn) { lo = min(lo, mn); hi = max(hi, mx); }
        }
    }
    for (int w = 32; w >= 1; w >>= 1) { lo = min(lo, __shfl_xor(lo, w, 64)); hi = max(hi, __shfl_xor(hi, w, 64)); }
    a_first = hi < 0 ? 1 : lo / FB_T;
    a_last = hi < 0 ? 0 : hi / FB_T;
}

// lane n takes lane n-1's v; lane 0 keeps `lane0`
__device__ __forceinline__ double fb_shr1(double v, double lane0) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// lane n takes lane n+1's v; lane 63 keeps `lane63`
__device__ __forceinline__ double fb_shl1(double v, double lane63) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane63), lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane63), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// LDS traffic of the one wave is in order; this keeps the compiler from moving it and waits for nothing in memory
__device__ __forceinline__ void fb_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void fb_stage_diagonals(const PgFbJob &J, FbSmem &M, int dbase, int r) {
    for (int k = r; k < 2 * FB_T; k += 64) {
        const int d = dbase + k;
        const bool in = d < J.nd;
        M.dmin[k] = in ? J.imin[d] : 0; M.dmax[k] = in ? J.imax[d] : -1; M.doff[k] = in ? J.doff[d] : 0;
    }
}

template <bool ALL_LDS>
__global__ __launch_bounds__(64) void pg_fb_forward_tiled(const PgFbJob *jobs) {
    __shared__ FbSmem M;
    const PgFbJob J = jobs[blockIdx.y];
    const int G = J.groups, r = (int)threadIdx.x;
    if ((int)blockIdx.x >= G) return;                              // (a launch is as wide as its widest pair)
    int *sync = J.sync;
    const int nbr = (J.Lx + FB_T - 1) / FB_T, nbc = (J.Ly + FB_T - 1) / FB_T;
    const double NI = ninf();
    const bool tab_lds = J.S * J.S <= 256;
    if (tab_lds) for (int k = r; k < J.S * J.S; k += 64) M.ltab[k] = J.ltab[k];
    for (int t = 0; t < nbr + nbc - 1; ++t) {
        int a_lo = max(0, t - (nbc - 1)), a_hi = min(t, nbr - 1);
        {
            int a_first, a_last;
            fb_live_rows(J, t, r, a_first, a_last);
            a_lo = max(a_lo, a_first); a_hi = min(a_hi, a_last);
        }
        for (int a = a_lo + (int)blockIdx.x; a <= a_hi; a += G) {
            const int i0 = a * FB_T, j0 = (t - a) * FB_T, dbase = i0 + j0;
            if (!fb_block_live(J, i0, j0, r)) continue;
#ifdef PG_FB_STATS
            const unsigned long long st0 = __builtin_readcyclecounter();
#endif
            __syncthreads();                                                       // (the block before is done with the staging arrays)
            // halo: the rows above (with the corner) and the columns to the left; the columns' records; the diagonals' intervals
            for (int k = r; k < FB_H * (FB_T + FB_H); k += 64) {
                const int u = k / (FB_T + FB_H), c_ = k % (FB_T + FB_H);
                const long long at = cell_at(J, i0 - FB_H + u, j0 - FB_H + c_);
                for (int q = 0; q < 3; ++q) M.ha[u][c_][q] = rd(J.F, at, q);
            }
            for (int k = r; k < FB_T * FB_H; k += 64) {
                const int row_ = k / FB_H, u = k % FB_H;
                const long long at = cell_at(J, i0 + row_, j0 - FB_H + u);
                for (int q = 0; q < 3; ++q) M.hb[row_][u][q] = rd(J.F, at, q);
            }
            {
                const int jc = j0 + r;
                const bool cv = jc < J.Ly && jc > 0;
                const int o0 = cv ? J.offR[jc] : 0, o1 = cv ? J.offR[jc + 1] : 0;
                M.c_off[r] = o0; M.c_n[r] = o1 - o0; M.c_st[r] = jc < J.Ly ? J.stR[jc] : 0;
                M.c_e0[r] = o1 > o0 ? J.srcR[o0] : 0; M.c_lw0[r] = o1 > o0 ? J.lwR[o0] : 0.0f;
                M.c_e1[r] = o1 > o0 + 1 ? J.srcR[o0 + 1] : 0; M.c_lw1[r] = o1 > o0 + 1 ? J.lwR[o0 + 1] : 0.0f;
            }
            fb_stage_diagonals(J, M, dbase, r);
            for (int k = r; k < FB_RING * FB_T; k += 64) { double *c = &M.ring[0][0][0] + 3 * k; c[0] = NI; c[1] = NI; c[2] = NI; }
            const int i = i0 + r;
            const bool rv = i < J.Lx && i > 0;
            const int l0 = rv ? J.offL[i] : 0, nl = rv ? J.offL[i + 1] - l0 : 0;
            const int p0 = nl > 0 ? J.srcL[l0] : 0, p1 = nl > 1 ? J.srcL[l0 + 1] : 0, stl = i < J.Lx ? J.stL[i] : 0;
            const double lwl0 = nl > 0 ? (double)J.lwL[l0] : 0.0, lwl1 = nl > 1 ? (double)J.lwL[l0 + 1] : 0.0;
            __syncthreads();
#ifdef PG_FB_STATS
            const unsigned long long st1 = __builtin_readcyclecounter();
#endif
            // The cells a SIMPLE cell reads -- one bwd edge per site, from the previous site -- stay in registers, as in the
            // Viterbi kernels: P this lane's cell of the step before (i, j-1), A the wave shift of P (i-1, j), C the A of the
            // step before (i-1, j-1); a lane starts from the column left of the block, lane 0 takes row i0-1 from the halo.
            const bool row_simple = nl == 1 && p0 == i - 1;
            double Px = M.hb[r][FB_H - 1][0], Py = M.hb[r][FB_H - 1][1], Pm = M.hb[r][FB_H - 1][2];
            double Ax = M.ha[FB_H - 1][FB_H - 1][0], Ay = M.ha[FB_H - 1][FB_H - 1][1], Am = M.ha[FB_H - 1][FB_H - 1][2];   // lane 0: the corner
            double Cx = NI, Cy = NI, Cm = NI;
            for (int s = 0; s <= 2 * FB_T - 2; ++s) {
                const int d = dbase + s;
                if (d >= J.nd) break;
                const int j = d - i, jj = s - r;
                const int mn = M.dmin[s], mx = M.dmax[s];
                const bool active = jj >= 0 && jj < FB_T && i < J.Lx && j < J.Ly && i >= mn && i <= mx;
                double fx = NI, fy = NI, fm = NI;
                {   // this step's register operands
                    const int tc = (s < FB_T ? s : FB_T - 1) + FB_H;               // halo column of (i0-1, j0+s)
                    const double t0 = M.ha[FB_H - 1][tc][0], t1 = M.ha[FB_H - 1][tc][1], t2 = M.ha[FB_H - 1][tc][2];
                    Cx = Ax; Cy = Ay; Cm = Am;
                    Ax = fb_shr1(Px, t0); Ay = fb_shr1(Py, t1); Am = fb_shr1(Pm, t2);
                }
                const bool col_simple = active && M.c_n[jj] == 1 && M.c_e0[jj] == j - 1;
                if (__builtin_amdgcn_ballot_w64(active && !(row_simple && col_simple)) == 0) {
                    if (active) {
                        const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, stl, M.c_st[jj], J.S);
                        const double w = lwl0 + (double)M.c_lw0[jj];
                        const double mm = J.l_ng + J.l_ng + sc + w, xm = J.l_ng + sc + w;
                        fx = lse3(Ax + J.l_ext, Ay + J.l_open, Am + J.l_ng + J.l_open);                        // VA:2153, 2184, 2215
                        fy = lse3(Py + J.l_ext, Px + J.l_open, Pm + J.l_ng + J.l_open);
                        fm = lse3(Cm + mm, Cx + xm, Cy + xm);                                                  // VA:2051, 2080, 2108
                        const fb_gd o = (fb_gd)(unsigned long long)J.F + 3 * (M.doff[s] + (i - mn));       // (a global store: a flat one counts against lgkmcnt too)
                        o[0] = fx; o[1] = fy; o[2] = fm;
                    }
                } else
                if (active) {
                    auto fetch = [&](int p, int q, double &x, double &y, double &m) {
                        x = NI; y = NI; m = NI;
                        if (p < 0 || q < 0) return;
                        const double *c = nullptr;
                        if (p >= i0 && q >= j0) {
                            if (d - (p + q) < FB_RING) c = M.ring[(p + q) % FB_RING][p - i0];
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a cell of this block that left the ring: the wave's own store
                        } else if (p < i0 && p >= i0 - FB_H && q >= j0 - FB_H) c = M.ha[p - (i0 - FB_H)][q - (j0 - FB_H)];
                        else if (p >= i0 && q < j0 && q >= j0 - FB_H) c = M.hb[p - i0][q - (j0 - FB_H)];
                        if (c) { x = c[0]; y = c[1]; m = c[2]; return; }
                        const long long at = cell_at(J, p, q);
                        x = rd(J.F, at, 0); y = rd(J.F, at, 1); m = rd(J.F, at, 2);
                    };
                    if (i == 0 && j == 0) {
                        fm = 0.0;                                                  // fwd_score = 1, VA:730
                    } else {
                        double x, y, m;
                        const int r0 = M.c_off[jj], nr = M.c_n[jj], q0 = M.c_e0[jj], q1 = M.c_e1[jj];
                        const double lwr0 = (double)M.c_lw0[jj], lwr1 = (double)M.c_lw1[jj];
                        auto srcl = [&](int k) { return k == 0 ? p0 : (k == 1 ? p1 : J.srcL[l0 + k]); };
                        auto srcr = [&](int k) { return k == 0 ? q0 : (k == 1 ? q1 : J.srcR[r0 + k]); };
                        for (int k = 0; k < nl; ++k) {
                            fetch(srcl(k), j, x, y, m);
                            fx = lse(fx, lse3(x + J.l_ext, y + J.l_open, m + J.l_ng + J.l_open));              // VA:2153, 2184, 2215
                        }
                        for (int k = 0; k < nr; ++k) {
                            fetch(i, srcr(k), x, y, m);
                            fy = lse(fy, lse3(y + J.l_ext, x + J.l_open, m + J.l_ng + J.l_open));
                        }
                        if (nl > 0 && nr > 0) {
                            const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, stl, M.c_st[jj], J.S);
                            const double mm = J.l_ng + J.l_ng + sc, xm = J.l_ng + sc;                         // VA:1383-1391
                            for (int k1 = 0; k1 < nl; ++k1)
                                for (int k2 = 0; k2 < nr; ++k2) {
                                    fetch(srcl(k1), srcr(k2), x, y, m);
                                    const double w = (k1 == 0 ? lwl0 : (k1 == 1 ? lwl1 : (double)J.lwL[l0 + k1])) +
                                                     (k2 == 0 ? lwr0 : (k2 == 1 ? lwr1 : (double)J.lwR[r0 + k2]));
                                    fm = lse(fm, lse3(m + mm + w, x + xm + w, y + xm + w));                   // VA:2051, 2080, 2108
                                }
                        }
                    }
                    const fb_gd o = (fb_gd)(unsigned long long)J.F + 3 * (M.doff[s] + (i - mn));       // (a global store: a flat one counts against lgkmcnt too)
                    o[0] = fx; o[1] = fy; o[2] = fm;
                }
                double *c = M.ring[d % FB_RING][r];
                c[0] = fx; c[1] = fy; c[2] = fm;
                if (active) { Px = fx; Py = fy; Pm = fm; } else if (jj >= 0) { Px = NI; Py = NI; Pm = NI; }
                fb_lds_fence();
            }
#ifdef PG_FB_STATS
            if (r == 0) {     // sync[16..]: blocks, prologue cycles, step cycles (diagnostic build)
                const unsigned long long st2 = __builtin_readcyclecounter();
                atomicAdd((unsigned long long *)(sync + 16), 1ull); atomicAdd((unsigned long long *)(sync + 18), st1 - st0);
                atomicAdd((unsigned long long *)(sync + 20), st2 - st1);
            }
#endif
        }
#ifdef PG_FB_STATS
        const unsigned long long sb0 = __builtin_readcyclecounter();
#endif
        if (!fb_barrier(sync, G * (t + 1))) return;
#ifdef PG_FB_STATS
        if (r == 0) atomicAdd((unsigned long long *)(sync + 22), __builtin_readcyclecounter() - sb0);
#endif
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // end corner, VA:1440-1552
        double acc = NI;
        const int l0 = J.offL[J.Lx], l1 = J.offL[J.Lx + 1], r0 = J.offR[J.Ly], r1 = J.offR[J.Ly + 1];
        auto mt = [&](int k1, int k2) { return rd(J.F, cell_at(J, J.srcL[k1], J.srcR[k2]), 2) + J.l_ng + (double)J.lwL[k1] + (double)J.lwR[k2]; };
        auto xc = [&](int k1) { return rd(J.F, cell_at(J, J.srcL[k1], J.Ly - 1), 0); };
        auto yc = [&](int k2) { return rd(J.F, cell_at(J, J.Lx - 1, J.srcR[k2]), 1); };
        if (l1 > l0 && r1 > r0) {
            acc = lse(acc, mt(l0, r0)); acc = lse(acc, xc(l0)); acc = lse(acc, yc(r0));
            for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(l0, k2)); acc = lse(acc, yc(k2)); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) {
                acc = lse(acc, mt(k1, r0)); acc = lse(acc, xc(k1));
                for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(k1, k2)); acc = lse(acc, yc(k2)); }
            }
        }
        J.totals[0] = acc;
    }
}

template <bool ALL_LDS>
__global__ __launch_bounds__(64) void pg_fb_backward_tiled(const PgFbJob *jobs) {
    __shared__ FbSmem M;
    const PgFbJob J = jobs[blockIdx.y];
    const int G = J.groups_b, r = (int)threadIdx.x;
    if ((int)blockIdx.x >= G) return;
    int *sync = J.sync + 8;
    const int nbr = (J.Lx + FB_T - 1) / FB_T, nbc = (J.Ly + FB_T - 1) / FB_T;
    const double NI = ninf();
    const bool tab_lds = J.S * J.S <= 256;
    if (tab_lds) for (int k = r; k < J.S * J.S; k += 64) M.ltab[k] = J.ltab[k];
    // (no fill of B: every cell of the band is written by the sweep before anything reads it; what
    //  initialise_array_corner_bwd assigns -- a handful of cells at the end corner, VA:740-854 -- is laid over the -inf a
    //  cell starts from, in the blocks that hold such a cell)
    int round = 0;
    for (int t = nbr + nbc - 2; t >= 0; --t) {
        int a_lo = max(0, t - (nbc - 1)), a_hi = min(t, nbr - 1);
        {
            int a_first, a_last;
            fb_live_rows(J, t, r, a_first, a_last);
            a_lo = max(a_lo, a_first); a_hi = min(a_hi, a_last);
        }
        for (int a = a_lo + (int)blockIdx.x; a <= a_hi; a += G) {
            const int i0 = a * FB_T, j0 = (t - a) * FB_T, dbase = i0 + j0;
            if (!fb_block_live(J, i0, j0, r)) continue;
            __syncthreads();
            // halo: the rows below (with the corner) and the columns to the right; the columns' fwd records (one more column than
            // the block has: a match moves to column j + 1); the diagonals' intervals
            for (int k = r; k < FB_H * (FB_T + FB_H); k += 64) {
                const int u = k / (FB_T + FB_H), c_ = k % (FB_T + FB_H);
                const long long at = cell_at(J, i0 + FB_T + u, j0 + c_);
                for (int q = 0; q < 3; ++q) M.ha[u][c_][q] = rd(J.B, at, q);
            }
            for (int k = r; k < FB_T * FB_H; k += 64) {
                const int row_ = k / FB_H, u = k % FB_H;
                const long long at = cell_at(J, i0 + row_, j0 + FB_T + u);
                for (int q = 0; q < 3; ++q) M.hb[row_][u][q] = rd(J.B, at, q);
            }
            for (int k = r; k <= FB_T; k += 64) {
                const int jc = j0 + k;
                const bool cv = jc < J.Ly;
                const int o0 = cv ? J.foffR[jc] : 0, o1 = cv ? J.foffR[jc + 1] : 0;
                M.c_off[k] = o0; M.c_n[k] = o1 - o0; M.c_st[k] = cv ? J.stR[jc] : 0;
                M.c_e0[k] = o1 > o0 ? J.fdstR[o0] : 0; M.c_lw0[k] = o1 > o0 ? J.flwR[o0] : 0.0f;
                M.c_e1[k] = o1 > o0 + 1 ? J.fdstR[o0 + 1] : 0; M.c_lw1[k] = o1 > o0 + 1 ? J.flwR[o0 + 1] : 0.0f;
            }
            fb_stage_diagonals(J, M, dbase, r);
            for (int k = r; k < FB_RING * FB_T; k += 64) { double *c = &M.ring[0][0][0] + 3 * k; c[0] = NI; c[1] = NI; c[2] = NI; }
            const int i = i0 + r;
            // does an assignment of initialise_array_corner_bwd fall into this block?  (lane r looks at the block's diagonals r, r + 64)
            bool init_here = false;
            for (int u = 0; u < 2 && J.n_init > 0; ++u) {
                const int sd = r + 64 * u, d_ = dbase + sd;
                if (sd > 2 * FB_T - 2 || d_ >= J.nd) continue;
                const int mn_ = J.imin[d_], lo = max(max(mn_, i0), d_ - (j0 + FB_T - 1)), hi = min(min(J.imax[d_], i0 + FB_T - 1), d_ - j0);
                if (lo > hi) continue;
                const long long first = 3 * (J.doff[d_] + (lo - mn_)), last = 3 * (J.doff[d_] + (hi - mn_)) + 2;
                for (int k = 0; k < J.n_init; ++k) init_here = init_here || (J.init_at[k] >= first && J.init_at[k] <= last);
            }
            const bool blk_init = __builtin_amdgcn_ballot_w64(init_here) != 0;
            const bool rv = i < J.Lx;
            const int l0 = rv ? J.foffL[i] : 0, nl = rv ? J.foffL[i + 1] - l0 : 0;
            const int t0 = nl > 0 ? J.fdstL[l0] : 0, t1 = nl > 1 ? J.fdstL[l0 + 1] : 0;
            const int st_t0 = nl > 0 && t0 < J.Lx ? J.stL[t0] : 0, st_t1 = nl > 1 && t1 < J.Lx ? J.stL[t1] : 0;
            const double lwl0 = nl > 0 ? (double)J.flwL[l0] : 0.0, lwl1 = nl > 1 ? (double)J.flwL[l0 + 1] : 0.0;
            __syncthreads();
            // registers, as in the forward sweep but mirrored: P this lane's cell of the step before (i, j+1), A the wave shift
            // that brings lane r+1's P (i+1, j), C the A of the step before (i+1, j+1); a lane starts from the column right of
            // the block, lane 63 takes row i0+64 from the halo
            const bool row_simple = nl == 1 && t0 == i + 1 && t0 < J.Lx;
            double Px = M.hb[r][0][0], Py = M.hb[r][0][1], Pm = M.hb[r][0][2];
            double Ax = M.ha[0][FB_T][0], Ay = M.ha[0][FB_T][1], Am = M.ha[0][FB_T][2];      // lane 63: the corner (i0+64, j0+64)
            double Cx = NI, Cy = NI, Cm = NI;
            for (int s = 2 * FB_T - 2; s >= 0; --s) {
                const int d = dbase + s;
                {   // (every step, also the ones past the matrix's last diagonal: the lanes' columns move with s)
                    const int bc = s - (FB_T - 1);                                   // lane 63's column in the block
                    const int bcc = bc < 0 ? 0 : bc;
                    const double t0_ = M.ha[0][bcc][0], t1_ = M.ha[0][bcc][1], t2_ = M.ha[0][bcc][2];
                    Cx = Ax; Cy = Ay; Cm = Am;
                    (void)Cx; (void)Cy;                                              // (a match moves through M only: VA:2269-2271)
                    Ax = fb_shl1(Px, t0_); Ay = fb_shl1(Py, t1_); Am = fb_shl1(Pm, t2_);
                }
                if (d >= J.nd) continue;
                const int j = d - i, jj = s - r;
                const int mn = M.dmin[s], mx = M.dmax[s];
                const bool active = jj >= 0 && jj < FB_T && i < J.Lx && j < J.Ly && i >= mn && i <= mx;
                double bx = NI, by = NI, bm = NI;
                const bool col_simple = active && M.c_n[jj] == 1 && M.c_e0[jj] == j + 1 && j + 1 < J.Ly;
                if (!blk_init && __builtin_amdgcn_ballot_w64(active && !(row_simple && col_simple)) == 0) {
                    if (active) {
                        const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, st_t0, M.c_st[jj + 1], J.S);
                        const double thru = Cm + sc + lwl0 + (double)M.c_lw0[jj];             // VA:2269-2271
                        bx = lse3(Ax + J.l_ext, Py + J.l_open, thru + J.l_ng);               // VA:2281-2303
                        by = lse3(Ax + J.l_open, Py + J.l_ext, thru + J.l_ng);
                        bm = lse3(Ax + J.l_ng + J.l_open, Py + J.l_ng + J.l_open, thru + J.l_ng + J.l_ng);
                        const fb_gd o = (fb_gd)(unsigned long long)J.B + 3 * (M.doff[s] + (i - mn));
                        o[0] = bx; o[1] = by; o[2] = bm;
                    }
                } else
                if (active) {
                    // state q of cell (t_, u) >= (i, j)
                    auto fetch = [&](int t_, int u, int q) -> double {
                        if (t_ >= J.Lx || u >= J.Ly) return NI;
                        if (t_ < i0 + FB_T && u < j0 + FB_T) {
                            if ((t_ + u) - d < FB_RING) return M.ring[(t_ + u) % FB_RING][t_ - i0][q];
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // a cell of this block that left the ring: the wave's own store
                        } else if (t_ >= i0 + FB_T && t_ < i0 + FB_T + FB_H && u < j0 + FB_T + FB_H) return M.ha[t_ - (i0 + FB_T)][u - j0][q];
                        else if (t_ < i0 + FB_T && u >= j0 + FB_T && u < j0 + FB_T + FB_H) return M.hb[t_ - i0][u - (j0 + FB_T)][q];
                        return rd(J.B, cell_at(J, t_, u), q);
                    };
                    const fb_gd o = (fb_gd)(unsigned long long)J.B + 3 * (M.doff[s] + (i - mn));
                    if (blk_init) {
                        const long long at3 = 3 * (M.doff[s] + (i - mn));
                        for (int k = 0; k < J.n_init; ++k) {
                            const long long w = J.init_at[k] - at3;
                            if (w == 0) bx = J.init_val[k]; else if (w == 1) by = J.init_val[k]; else if (w == 2) bm = J.init_val[k];
                        }
                    }
                    const int r0 = M.c_off[jj], nr = M.c_n[jj], u0 = M.c_e0[jj], u1 = M.c_e1[jj];
                    const double lwr0 = (double)M.c_lw0[jj], lwr1 = (double)M.c_lw1[jj];
                    auto dstl = [&](int k) { return k == 0 ? t0 : (k == 1 ? t1 : J.fdstL[l0 + k]); };
                    auto dstr = [&](int k) { return k == 0 ? u0 : (k == 1 ? u1 : J.fdstR[r0 + k]); };
                    for (int k = 0; k < nl; ++k) {                                 // iterate_fwd_edges_for_gap, left site
                        const int t_ = dstl(k);
                        if (t_ >= J.Lx) continue;                                  // VA:1580
                        const double nx = fetch(t_, j, 0);
                        bx = lse(bx, nx + J.l_ext); by = lse(by, nx + J.l_open); bm = lse(bm, nx + J.l_ng + J.l_open);   // VA:2281-2303
                    }
                    for (int k = 0; k < nr; ++k) {
                        const int u = dstr(k);
                        if (u >= J.Ly) continue;
                        const double ny = fetch(i, u, 1);
                        by = lse(by, ny + J.l_ext); bx = lse(bx, ny + J.l_open); bm = lse(bm, ny + J.l_ng + J.l_open);
                    }
                    for (int k1 = 0; k1 < nl; ++k1)                                // iterate_fwd_edges_for_match
                        for (int k2 = 0; k2 < nr; ++k2) {
                            const int t_ = dstl(k1), u = dstr(k2);
                            if (t_ >= J.Lx || u >= J.Ly) continue;
                            const int sl = k1 == 0 ? st_t0 : (k1 == 1 ? st_t1 : J.stL[t_]);
                            const int sr = (u >= j0 && u <= j0 + FB_T) ? M.c_st[u - j0] : J.stR[u];
                            const double sc = fb_score<ALL_LDS>(tab_lds, M.ltab, J.ltab, sl, sr, J.S);
                            const double thru = fetch(t_, u, 2) + sc + (k1 == 0 ? lwl0 : (k1 == 1 ? lwl1 : (double)J.flwL[l0 + k1])) +
                                                (k2 == 0 ? lwr0 : (k2 == 1 ? lwr1 : (double)J.flwR[r0 + k2]));   // VA:2269-2271
                            bx = lse(bx, thru + J.l_ng); by = lse(by, thru + J.l_ng); bm = lse(bm, thru + J.l_ng + J.l_ng);
                        }
                    o[0] = bx; o[1] = by; o[2] = bm;
                }
                double *c = M.ring[d % FB_RING][r];
                c[0] = bx; c[1] = by; c[2] = bm;
                if (active) { Px = bx; Py = by; Pm = bm; } else if (jj < FB_T) { Px = NI; Py = NI; Pm = NI; }
                fb_lds_fence();
            }
        }
        ++round;
        if (!fb_barrier(sync, G * round)) return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) J.totals[1] = rd(J.B, cell_at(J, 0, 0), 2);
}

struct FwdLists { std::vector<int> off, dst, slot; std::vector<float> lw; };

// A site's fwd list = its outgoing edges in creation order (Edge::index); the log weight is the bwd CSR's.
FwdLists forward_lists(const pagan_graph *g) {
    struct E { int src, dst, eid; float lw; };
    std::vector<E> es;
    es.reserve(g->bwd_off[g->n_sites]);
    for (int s = 0; s < g->n_sites; ++s)
        for (int k = g->bwd_off[s]; k < g->bwd_off[s + 1]; ++k) es.push_back({g->bwd_src[k], s, g->bwd_eid[k], g->bwd_logw[k]});
    std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.src != b.src ? a.src < b.src : a.eid < b.eid; });
    FwdLists f;
    f.off.assign(g->n_sites + 1, 0);
    for (const E &e : es) f.off[e.src + 1]++;
    for (int s = 0; s < g->n_sites; ++s) f.off[s + 1] += f.off[s];
    for (const E &e : es) { f.dst.push_back(e.dst); f.lw.push_back(e.lw); }
    return f;
}

} // namespace

int pagan_internal_replay(const pagan_graph *L, const pagan_graph *R, int64_t cells, const int *endcell, double endscore,
                          const int *trace, pagan_result *out);     // dp_abi.hip

// Arenas of finished handles are kept for the next one on the same device (a pair's two matrices are hundreds of MB:
// allocating and freeing them -- a device-wide synchronisation -- per pair was half of a pair's wall-clock); at most 32 (a batch of a tree level's pairs returns that many at once; eight until round 5)
// idle ones per process, pagan_fb_release_cache frees them.
namespace {
struct FbArenaPool {
    std::mutex m;
    struct Slab { int device; char *p; size_t cap; };
    std::vector<Slab> idle;
    char *take(int device, size_t need, size_t *cap) {
        {
            std::lock_guard<std::mutex> g(m);
            int best = -1;
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].device == device && idle[k].cap >= need && (best < 0 || idle[k].cap < idle[best].cap)) best = (int)k;
            if (best >= 0 && idle[best].cap <= 2 * need + (64u << 20)) { Slab s_ = idle[best]; idle.erase(idle.begin() + best); *cap = s_.cap; return s_.p; }
        }
        char *p = nullptr;
        *cap = need + need / 16;
        if (hipMalloc((void **)&p, *cap) != hipSuccess) {
            (void)hipGetLastError();
            clear();                                           // (idle arenas may be what is in the way)
            if (hipMalloc((void **)&p, *cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        return p;
    }
    // idle arenas are bounded by count AND by bytes (a 10 k x 10 k pair's two matrices are 4.8 GB: eight of those idle would
    // sit on 38 GB that the Viterbi batches of the same process may need); the calling thread stays on its device
    static constexpr size_t kMaxIdleBytes = (size_t)8 << 30;
    void drop_front() {
        int cur = -1;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(idle.front().device); (void)hipFree(idle.front().p); idle.erase(idle.begin());
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    void give(int device, char *p, size_t cap) {
        std::lock_guard<std::mutex> g(m);
        idle.push_back({device, p, cap});
        size_t total = 0;
        for (auto &s_ : idle) total += s_.cap;
        while (idle.size() > 32 || (total > kMaxIdleBytes && idle.size() > 1)) { total -= idle.front().cap; drop_front(); }
    }
    void clear() {
        std::lock_guard<std::mutex> g(m);
        while (!idle.empty()) drop_front();
    }
};
FbArenaPool fb_arena_pool;
} // namespace
extern "C" void pagan_fb_internal_release_cache() { fb_arena_pool.clear(); }

struct pagan_fb {
    int device = 0;
    int Lx = 0, Ly = 0, S = 0;
    const pagan_graph *L = nullptr, *R = nullptr;       // borrowed: must outlive the handle for sample_path
    std::vector<float> score;                           // copy of the probability table
    float gap_open = 0, gap_ext = 0, non_gap = 0;
    DiagIndex dx;
    RowBand rb;
    char *arena = nullptr;
    size_t arena_cap = 0;
    double *dF = nullptr, *dB = nullptr;
    double totals[2] = {0, 0};
    float kernel_ms[2] = {0, 0};                        // pg_fb_forward, pg_fb_backward (HIP events)
    int groups = 1;                                     // workgroups a diagonal's cells were spread over
    bool ring = false;                                  // the LDS-ring sweeps ran
    std::vector<double> hF;                             // downloaded lazily
    long long at(int i, int j) const {
        if (i < 0 || j < 0 || i >= Lx || j >= Ly) return -1;
        const int d = i + j;
        return (i >= dx.imin[d] && i <= dx.imax[d]) ? dx.doff[d] + (i - dx.imin[d]) : -1;
    }
};

extern "C" {

} // extern "C"

namespace {
// One pair up to the upload of its inputs: validation, band index, lists, arena, the job record with its workgroup count
// (`groups_cap`: what the caller's launch leaves this pair) -- what pagan_fb_run and pagan_fb_run_batch share.
struct FbStaged {
    pagan_fb *fb = nullptr;
    const PgFbJob *d_job = nullptr;          // the pair's job record on the device
    PgFbJob job;                             // ... and what it holds
    char *d_tot = nullptr, *d_sync = nullptr;
    int groups = 1, groups_b = 1, block = 64;
    bool ring = false;                       // the sweeps are pg_fb_forward_ring / pg_fb_backward_ring (block = their B)
    double t_host[4] = {0, 0, 0, 0};         // band index + lists, arena, staging, upload (seconds)
    size_t arena_bytes = 0;
};
static int fb_stage(const pagan_graph *left, const pagan_graph *right, const pagan_model_prob *model, const pagan_band *band,
                    const pagan_opts *opts, int groups_cap, int groups_cap_b, FbStaged *st) {
    if (!left || !right || !model || !st || !model->score || model->n_states < 1) return PAGAN_E_ARG;
    int rc = check_graph(left);
    if (rc == PAGAN_OK) rc = check_graph(right);
    if (rc != PAGAN_OK) return rc;
    const int Lx = left->n_sites - 1, Ly = right->n_sites - 1, S = model->n_states;
    auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double th0 = now_();
    for (int s = 1; s < Lx; ++s) if (left->state[s] < 0 || left->state[s] >= S) return PAGAN_E_MODEL;
    for (int s = 1; s < Ly; ++s) if (right->state[s] < 0 || right->state[s] >= S) return PAGAN_E_MODEL;
    if (!(model->gap_open > 0) || !(model->gap_ext > 0) || !(model->non_gap > 0)) return PAGAN_E_MODEL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAGAN_E_NODEVICE;
    pagan_fb *fb = new pagan_fb();
    std::unique_ptr<pagan_fb, void (*)(pagan_fb *)> guard(fb, [](pagan_fb *p) { pagan_fb_destroy(p); });
    if (opts && opts->device >= 0) { FB_TRY(hipSetDevice(opts->device)); fb->device = opts->device; }
    else FB_TRY(hipGetDevice(&fb->device));
    fb->Lx = Lx; fb->Ly = Ly; fb->S = S; fb->L = left; fb->R = right;
    fb->score.assign(model->score, model->score + (size_t)S * S);
    fb->gap_open = model->gap_open; fb->gap_ext = model->gap_ext; fb->non_gap = model->non_gap;
    rc = fb->rb.build(Lx, Ly, band);
    if (rc != PAGAN_OK) return rc;
    fb->dx.build(Lx, Ly, fb->rb);
    const long long cells = fb->dx.cells;
    const int nd = Lx + Ly - 1;
    const FwdLists fl = forward_lists(left), fr = forward_lists(right);
    std::vector<double> ltab((size_t)S * S);
    for (size_t k = 0; k < ltab.size(); ++k) ltab[k] = std::log((double)model->score[k]);
    // initialise_array_corner_bwd (VA:740-854): assignments onto the cells the end corner reads
    std::vector<long long> init_at;
    std::vector<double> init_val;
    const double l_ng = std::log((double)model->non_gap);
    int init_dmin = nd;
    auto put_init = [&](int i, int j, int s, double v) {
        const long long a = fb->at(i, j);
        if (a < 0) return;
        init_dmin = std::min(init_dmin, i + j);
        for (size_t k = 0; k < init_at.size(); ++k) if (init_at[k] == 3 * a + s) { init_val[k] = v; return; }   // later assignment wins
        init_at.push_back(3 * a + s); init_val.push_back(v);
    };
    put_init(Lx - 1, Ly - 1, PAGAN_M_MAT, l_ng);
    {
        const int l0 = left->bwd_off[Lx], l1 = left->bwd_off[Lx + 1], r0 = right->bwd_off[Ly], r1 = right->bwd_off[Ly + 1];
        if (l1 > l0 && r1 > r0)
            for (int k1 = l0; k1 < l1; ++k1)
                for (int k2 = r0; k2 < r1; ++k2)
                    put_init(left->bwd_src[k1], right->bwd_src[k2], PAGAN_M_MAT, l_ng + (double)left->bwd_logw[k1] + (double)right->bwd_logw[k2]);
        for (int k1 = l0; k1 < l1; ++k1) put_init(left->bwd_src[k1], Ly - 1, PAGAN_X_MAT, 0.0);
        for (int k2 = r0; k2 < r1; ++k2) put_init(Lx - 1, right->bwd_src[k2], PAGAN_Y_MAT, 0.0);
    }
    // one arena: inputs, then F and B
    size_t cur = 0;
    auto take = [&](size_t bytes) { const size_t o = cur; cur = (cur + bytes + 255) / 256 * 256; return o; };
    const int nbL = left->bwd_off[left->n_sites], nbR = right->bwd_off[right->n_sites];
    const size_t o_job = take(sizeof(PgFbJob));
    const size_t o_stL = take(4 * (size_t)left->n_sites), o_offL = take(4 * ((size_t)left->n_sites + 1)), o_srcL = take(4 * (size_t)nbL), o_lwL = take(4 * (size_t)nbL);
    const size_t o_stR = take(4 * (size_t)right->n_sites), o_offR = take(4 * ((size_t)right->n_sites + 1)), o_srcR = take(4 * (size_t)nbR), o_lwR = take(4 * (size_t)nbR);
    const size_t o_foL = take(4 * fl.off.size()), o_fdL = take(4 * fl.dst.size()), o_fwL = take(4 * fl.lw.size());
    const size_t o_foR = take(4 * fr.off.size()), o_fdR = take(4 * fr.dst.size()), o_fwR = take(4 * fr.lw.size());
    const size_t o_tab = take(8 * ltab.size());
    const size_t o_imin = take(4 * (size_t)nd), o_imax = take(4 * (size_t)nd), o_doff = take(8 * (size_t)nd);
    const size_t o_iat = take(8 * init_at.size()), o_ival = take(8 * init_val.size());
    const size_t o_tot = take(16);
    const size_t o_sync = take(256);                      // two barrier counters + give-up words of the wide sweeps (zero)
    const size_t in_bytes = cur;
    const size_t o_F = take(24 * (size_t)cells), o_B = take(24 * (size_t)cells);
    const double th1 = now_();
    fb->arena = fb_arena_pool.take(fb->device, cur, &fb->arena_cap);
    if (!fb->arena) return PAGAN_E_NOMEM;
    std::vector<char> stage(in_bytes, 0);
    auto put = [&](size_t off, const void *src, size_t bytes) { if (bytes) std::memcpy(stage.data() + off, src, bytes); };
    put(o_stL, left->state, 4 * (size_t)left->n_sites); put(o_offL, left->bwd_off, 4 * ((size_t)left->n_sites + 1));
    put(o_srcL, left->bwd_src, 4 * (size_t)nbL); put(o_lwL, left->bwd_logw, 4 * (size_t)nbL);
    put(o_stR, right->state, 4 * (size_t)right->n_sites); put(o_offR, right->bwd_off, 4 * ((size_t)right->n_sites + 1));
    put(o_srcR, right->bwd_src, 4 * (size_t)nbR); put(o_lwR, right->bwd_logw, 4 * (size_t)nbR);
    put(o_foL, fl.off.data(), 4 * fl.off.size()); put(o_fdL, fl.dst.data(), 4 * fl.dst.size()); put(o_fwL, fl.lw.data(), 4 * fl.lw.size());
    put(o_foR, fr.off.data(), 4 * fr.off.size()); put(o_fdR, fr.dst.data(), 4 * fr.dst.size()); put(o_fwR, fr.lw.data(), 4 * fr.lw.size());
    put(o_tab, ltab.data(), 8 * ltab.size());
    put(o_imin, fb->dx.imin.data(), 4 * (size_t)nd); put(o_imax, fb->dx.imax.data(), 4 * (size_t)nd); put(o_doff, fb->dx.doff.data(), 8 * (size_t)nd);
    put(o_iat, init_at.data(), 8 * init_at.size()); put(o_ival, init_val.data(), 8 * init_val.size());
    char *b = fb->arena;
    // wide diagonals: 64 x 64 blocks over as many workgroups as a block anti-diagonal has blocks; PAGAN_FB_GROUPS=1 keeps the
    // one-workgroup sweeps
    const int mw = fb->dx.max_width;
    int groups = mw > 256 ? std::min({FB_MAX_GROUPS, (Lx + FB_T - 1) / FB_T, (Ly + FB_T - 1) / FB_T}) : 1;
    // A long tunnel (round 5; the leaf pairs of 32 x 100 kb: 2e5 diagonals of ~25 cells) on the block schedule as well: its one
    // workgroup paid a __syncthreads() and a round trip to L2 per cell diagonal (2.7 us: 0.55 s a sweep); as blocks it is the few
    // blocks the band has on a block anti-diagonal (fb_live_rows), a wave each, operands in registers / LDS, one counter barrier per
    // 64 diagonals.  Workgroups: the blocks a diagonal of `mw` cells can lie in while it moves through 127 diagonals.
    // PAGAN_FB_BAND_MIN_ND: the shortest pair (in cell diagonals) that takes this path (tests: 0; "off": none).
    // ... and a long tunnel between plain sequences (every site one edge, from the site before it: leaves) on the LDS-ring sweeps
    // (pg_fb_forward_ring): PAGAN_FB_RING=0 switches them off, PAGAN_FB_RING_MIN_ND is their shortest pair (default 256 diagonals)
    bool ring = false;
    {
        int min_nd = 256;            // (a step of the ring sweeps is 1 us against the one-workgroup kernels' 2.7: worth it from a few hundred diagonals on)
        if (const char *e = std::getenv("PAGAN_FB_RING_MIN_ND")) min_nd = std::atoi(e);
        const char *re = std::getenv("PAGAN_FB_RING");
        auto plain = [](const pagan_graph *g) {
            for (int sidx = 1; sidx < g->n_sites; ++sidx)
                if (g->bwd_off[sidx + 1] - g->bwd_off[sidx] != 1 || g->bwd_src[g->bwd_off[sidx]] != sidx - 1) return false;
            return g->bwd_off[1] == 0;
        };
        bool gapless = true;                                          // (no cell diagonal without a cell)
        for (int d = 0; d < nd && gapless; ++d) gapless = fb->dx.imax[d] >= fb->dx.imin[d];
        ring = !(re && std::strcmp(re, "0") == 0) && nd >= min_nd && mw <= FB_RG_MAXB && Lx >= 2 && Ly >= 2 && (int)init_at.size() <= FB_RG_INIT && gapless && plain(left) && plain(right);
        if (ring && !std::getenv("PAGAN_FB_GROUPS")) groups = 1;
    }
    {
        int min_nd = 4096;
        if (const char *e = std::getenv("PAGAN_FB_BAND_MIN_ND")) min_nd = std::strcmp(e, "off") == 0 ? 0x7fffffff : std::atoi(e);
        if (ring) {}
        else if (groups == 1 && nd >= min_nd && Lx >= 2 && Ly >= 2)
            groups = std::max(2, std::min({FB_MAX_GROUPS, (mw + FB_T - 1) / FB_T + 2, (Lx + FB_T - 1) / FB_T, (Ly + FB_T - 1) / FB_T}));
        // (and a pair that is "wide" by a box of its tunnel -- one of the 16 leaf pairs has a diagonal of 295 cells -- does not need
        //  the 64 workgroups of a full matrix at every barrier: a diagonal of mw cells lies in mw / 64 + 2 blocks at most)
        else if (groups > 1) groups = std::max(2, std::min(groups, (mw + FB_T - 1) / FB_T + 2));
    }
    if (const char *e = std::getenv("PAGAN_FB_GROUPS")) { groups = std::max(1, std::min(FB_MAX_GROUPS, std::atoi(e))); ring = false; }
    const int groups_b = groups > 1 ? std::max(2, std::min(groups, groups_cap_b)) : 1;
    if (groups > 1) groups = std::max(2, std::min(groups, groups_cap));
    fb->groups = groups;
    PgFbJob J;
    J.Lx = Lx; J.Ly = Ly; J.nd = nd; J.S = S;
    J.l_ext = std::log((double)model->gap_ext); J.l_open = std::log((double)model->gap_open); J.l_ng = l_ng;
    J.stL = (const int *)(b + o_stL); J.offL = (const int *)(b + o_offL); J.srcL = (const int *)(b + o_srcL); J.lwL = (const float *)(b + o_lwL);
    J.stR = (const int *)(b + o_stR); J.offR = (const int *)(b + o_offR); J.srcR = (const int *)(b + o_srcR); J.lwR = (const float *)(b + o_lwR);
    J.foffL = (const int *)(b + o_foL); J.fdstL = (const int *)(b + o_fdL); J.flwL = (const float *)(b + o_fwL);
    J.foffR = (const int *)(b + o_foR); J.fdstR = (const int *)(b + o_fdR); J.flwR = (const float *)(b + o_fwR);
    J.ltab = (const double *)(b + o_tab);
    J.imin = (const int *)(b + o_imin); J.imax = (const int *)(b + o_imax); J.doff = (const long long *)(b + o_doff);
    J.cells = cells;
    J.F = (double *)(b + o_F); J.B = (double *)(b + o_B);
    J.n_init = (int)init_at.size(); J.init_at = (const long long *)(b + o_iat); J.init_val = (const double *)(b + o_ival);
    J.totals = (double *)(b + o_tot);
    J.groups = groups; J.groups_b = groups_b; J.sync = (int *)(b + o_sync);
    J.init_dmin = init_dmin;
    std::memcpy(stage.data() + o_job, &J, sizeof(J));
    fb->dF = J.F; fb->dB = J.B;
    const double th2 = now_();
    FB_TRY(hipMemcpy(fb->arena, stage.data(), in_bytes, hipMemcpyHostToDevice));
    const double th3 = now_();
    // a thread per cell of the widest diagonal, up to the 1024 of a workgroup (a cell is ~20 exp / log1p calls: a thread with
    // eight cells of a 2,000-cell diagonal was the whole sweep's pace)
    st->block = mw >= 768 ? 1024 : mw >= 384 ? 512 : mw >= 192 ? 256 : (mw >= 96 ? 128 : 64);
    st->ring = ring && groups == 1;
    if (st->ring) st->block = mw > 512 ? 1024 : (mw > 256 ? 512 : (mw > 128 ? 256 : (mw > 64 ? 128 : 64)));     // a thread per row of the widest diagonal
    fb->ring = st->ring;
    st->groups = groups; st->groups_b = groups_b;
    st->d_job = (const PgFbJob *)(b + o_job); st->d_tot = b + o_tot; st->d_sync = b + o_sync; st->job = J;
    st->t_host[0] = th1 - th0; st->t_host[1] = th2 - th1; st->t_host[2] = 0.0; st->t_host[3] = th3 - th2;
    st->arena_bytes = cur;
    st->fb = guard.release();
    return PAGAN_OK;
}

// ... and behind the kernels: the totals, the barriers' give-up words
static int fb_finish(FbStaged *st, float fwd_ms, float bwd_ms) {
    pagan_fb *fb = st->fb;
    fb->kernel_ms[0] = fwd_ms; fb->kernel_ms[1] = bwd_ms;
    FB_TRY(hipMemcpy(fb->totals, st->d_tot, 16, hipMemcpyDeviceToHost));
    int sy[64];
    FB_TRY(hipMemcpy(sy, st->d_sync, sizeof(sy), hipMemcpyDeviceToHost));
    if (sy[1] != 0 || sy[9] != 0) return PAGAN_E_INTERNAL;          // a barrier of a wide sweep ran into its limit
#ifdef PG_FB_STATS
    {
        const unsigned long long *q = (const unsigned long long *)(sy + 16);
        std::fprintf(stderr, "pagan_fb: forward: %llu blocks, prologue %.0f cycles per block, steps %.0f per block, barrier %.0f per workgroup and block diagonal (%d groups)\n",
                     q[0], q[0] ? (double)q[1] / q[0] : 0.0, q[0] ? (double)q[2] / q[0] : 0.0,
                     st->groups ? (double)q[3] / st->groups / ((fb->Lx + 63) / 64 + (fb->Ly + 63) / 64 - 1) : 0.0, st->groups);
    }
#endif
    return PAGAN_OK;
}

// A sweep whose workgroups meet at a counter barrier needs ALL of them on the chip at once; a caller may have any number
// of alignments in flight from as many threads, and workgroups of one sweep holding compute units while they wait for
// siblings that other waiting sweeps keep out would never end (the barrier's spin limit would turn that into an error,
// seconds later).  So the tiled sweeps of a device share a budget of workgroup slots well inside what the device holds
// at this kernel's LDS size (3 per compute unit); a launch that does not fit waits here, on the host.
struct FbSlots {
    std::mutex m; std::condition_variable cv; int used = 0;
    void take(int n, int cap) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return used == 0 || used + n <= cap; }); used += n; }
    void give(int n) { { std::lock_guard<std::mutex> l(m); used -= n; } cv.notify_all(); }
};
static FbSlots fb_slots_of[64];
static int fb_slot_cap(int device) {
    int n_cu = 0, per_cu_x16 = 46;                              // 2.9 workgroups per compute unit of the 3 that fit (PAGAN_FB_SLOTS_X16: A/B; 2.5 until round 5)
    if (const char *e = std::getenv("PAGAN_FB_SLOTS_X16")) per_cu_x16 = std::max(8, std::min(48, std::atoi(e)));
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) return per_cu_x16 * n_cu / 16;
    return 512;
}
// threads a row of a ring sweep's workgroup of B rows: three up to B = 256, two with 512 (PAGAN_FB_RING_SPLIT=0: one), see FB_RING_THREADS
static int fb_ring_split(int B) {
    const char *e = std::getenv("PAGAN_FB_RING_SPLIT");
    if (e && std::strcmp(e, "0") == 0) return 1;
    return B <= 256 ? 3 : (B <= 512 ? 2 : 1);          // (1,024 threads is what a workgroup holds)
}
template <bool FWD, bool ALL_LDS>
static void fb_launch_ring_(int nsplit, unsigned grid, int B, hipStream_t st, const PgFbJob *jobs) {
    const dim3 g(grid), t((unsigned)(nsplit * B));
    if (FWD) {
        if (B > 512) hipLaunchKernelGGL((pg_fb_forward_ring<ALL_LDS, 1, 1024>), g, t, 0, st, jobs);
        else if (nsplit == 3) hipLaunchKernelGGL((pg_fb_forward_ring<ALL_LDS, 3, 512>), g, t, 0, st, jobs);
        else if (nsplit == 2) hipLaunchKernelGGL((pg_fb_forward_ring<ALL_LDS, 2, 512>), g, t, 0, st, jobs);
        else hipLaunchKernelGGL((pg_fb_forward_ring<ALL_LDS, 1, 512>), g, t, 0, st, jobs);
    } else {
        if (B > 512) hipLaunchKernelGGL((pg_fb_backward_ring<ALL_LDS, 1, 1024>), g, t, 0, st, jobs);
        else if (nsplit == 3) hipLaunchKernelGGL((pg_fb_backward_ring<ALL_LDS, 3, 512>), g, t, 0, st, jobs);
        else if (nsplit == 2) hipLaunchKernelGGL((pg_fb_backward_ring<ALL_LDS, 2, 512>), g, t, 0, st, jobs);
        else hipLaunchKernelGGL((pg_fb_backward_ring<ALL_LDS, 1, 512>), g, t, 0, st, jobs);
    }
}
// one launch of ring sweeps: `grid` pairs of B rows each, forward or backward, score tables all in LDS or not
static void fb_launch_ring(bool fwd, bool all_lds, unsigned grid, int B, hipStream_t st, const PgFbJob *jobs) {
    const int ns = fb_ring_split(B);
    if (fwd) { if (all_lds) fb_launch_ring_<true, true>(ns, grid, B, st, jobs); else fb_launch_ring_<true, false>(ns, grid, B, st, jobs); }
    else { if (all_lds) fb_launch_ring_<false, true>(ns, grid, B, st, jobs); else fb_launch_ring_<false, false>(ns, grid, B, st, jobs); }
}
struct FbSlotLease { FbSlots *s; int n; ~FbSlotLease() { if (n > 0) s->give(n); } };

} // namespace

extern "C" {

int pagan_fb_run(const pagan_graph *left, const pagan_graph *right, const pagan_model_prob *model, const pagan_band *band,
                 const pagan_opts *opts, pagan_fb **out) {
    if (!out) return PAGAN_E_ARG;
    auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    FbStaged st;
    int rc = fb_stage(left, right, model, band, opts, FB_MAX_GROUPS, FB_MAX_GROUPS, &st);
    if (rc != PAGAN_OK) return rc;
    pagan_fb *fb = st.fb;
    std::unique_ptr<pagan_fb, void (*)(pagan_fb *)> guard(fb, [](pagan_fb *p) { pagan_fb_destroy(p); });
    const double th3 = now_();
    const int groups = st.groups, block = st.block;
    const bool all_lds = st.job.S * st.job.S <= 256;         // (the kernels without the score table's load from memory)
    // the two sweeps are independent of each other: side by side on two streams
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    FB_TRY(hipStreamCreate(&s1)); FB_TRY(hipStreamCreate(&s2));
    FB_TRY(hipEventCreate(&e0)); FB_TRY(hipEventCreate(&e1)); FB_TRY(hipEventCreate(&e2)); FB_TRY(hipEventCreate(&e3));
    FB_TRY(hipEventRecord(e0, s1));
    FbSlots &slots = fb_slots_of[fb->device & 63];
    FbSlotLease lease{&slots, groups > 1 ? 2 * groups : 0};
    if (groups > 1) slots.take(2 * groups, fb_slot_cap(fb->device));
    if (groups > 1) { if (all_lds) hipLaunchKernelGGL((pg_fb_forward_tiled<true>), dim3(groups, 1), dim3(64), 0, s1, st.d_job); else hipLaunchKernelGGL((pg_fb_forward_tiled<false>), dim3(groups, 1), dim3(64), 0, s1, st.d_job); }
    else if (st.ring) fb_launch_ring(true, all_lds, 1, block, s1, st.d_job);
    else hipLaunchKernelGGL(pg_fb_forward, dim3(1), dim3(block), 0, s1, st.d_job);
    FB_TRY(hipEventRecord(e1, s1));
    FB_TRY(hipEventRecord(e2, s2));
    if (groups > 1) { if (all_lds) hipLaunchKernelGGL((pg_fb_backward_tiled<true>), dim3(groups, 1), dim3(64), 0, s2, st.d_job); else hipLaunchKernelGGL((pg_fb_backward_tiled<false>), dim3(groups, 1), dim3(64), 0, s2, st.d_job); }
    else if (st.ring) fb_launch_ring(false, all_lds, 1, block, s2, st.d_job);
    else hipLaunchKernelGGL(pg_fb_backward, dim3(1), dim3(block), 0, s2, st.d_job);
    FB_TRY(hipEventRecord(e3, s2));
    FB_TRY(hipGetLastError());
    const double th4 = now_();
    FB_TRY(hipStreamSynchronize(s1)); FB_TRY(hipStreamSynchronize(s2));
    const double th5 = now_();
    float fwd_ms = 0, bwd_ms = 0;
    (void)hipEventElapsedTime(&fwd_ms, e0, e1);
    (void)hipEventElapsedTime(&bwd_ms, e2, e3);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
    rc = fb_finish(&st, fwd_ms, bwd_ms);
    if (rc != PAGAN_OK) return rc;
    if (std::getenv("PAGAN_DP_VERBOSE"))
        std::fprintf(stderr, "pagan_fb: host: band index + lists %.1f ms, arena (%.0f MB) %.1f ms, upload %.1f ms, streams + launch (+ slot wait) %.1f ms, kernels %.1f ms, rest %.1f ms\n",
                     1e3 * st.t_host[0], st.arena_bytes / 1048576.0, 1e3 * st.t_host[1], 1e3 * st.t_host[3], 1e3 * (th4 - th3), 1e3 * (th5 - th4), 1e3 * (now_() - th5));
    *out = guard.release();
    return PAGAN_OK;
}

// Several pairs at once (round 5): the sweeps of ALL the pairs' wide matrices in ONE launch per direction, grid = (most workgroups
// of any pair, pairs) -- concurrency is no longer what the HIP runtime's hardware queues allow (eight kernels at a time: four
// pairs), but what the chip holds.  Every tiled pair gets an equal share of the device's workgroup slots (its sweeps may then
// take a block anti-diagonal in two or three turns); pairs whose diagonals are narrow keep their one-workgroup kernels, a
// stream each.  out[k] as from pagan_fb_run; on an error nothing is handed back.
int pagan_fb_run_batch(int32_t n, const pagan_graph *const *left, const pagan_graph *const *right, const pagan_model_prob *const *model,
                       const pagan_band *const *band, const pagan_opts *opts, pagan_fb **out) {
    if (n < 0 || (n > 0 && (!left || !right || !model || !out))) return PAGAN_E_ARG;
    for (int k = 0; k < n; ++k) out[k] = nullptr;
    if (n == 0) return PAGAN_OK;
    int device = 0;
    if (opts && opts->device >= 0) device = opts->device; else if (hipGetDevice(&device) != hipSuccess) return PAGAN_E_NODEVICE;
    const int cap = fb_slot_cap(device);
    std::vector<FbStaged> st(n);
    std::vector<int> rcs(n, PAGAN_OK);
    int done = 0, rc = PAGAN_OK;
    auto cleanup = [&]() { for (int k = 0; k < n; ++k) if (st[k].fb) { pagan_fb_destroy(st[k].fb); st[k].fb = nullptr; } };
    while (done < n && rc == PAGAN_OK) {
        // a chunk of pairs whose tiled sweeps fit the slot budget with at least eight workgroups a sweep
        const int chunk = std::min(n - done, std::max(1, cap / 16));
        // the backward sweep is the slower of the two when a block anti-diagonal takes several turns: it gets the larger share
        // of the slots (PAGAN_FB_SPLIT: the forward sweeps' per cent, A/B)
        int split = 40;
        if (const char *e = std::getenv("PAGAN_FB_SPLIT")) split = std::max(20, std::min(80, std::atoi(e)));
        const int per_pair = std::max(4, cap / chunk);            // slots of one pair, both sweeps
        const int groups_cap = std::max(2, (per_pair * split + 50) / 100), groups_cap_b = std::max(2, per_pair - groups_cap);
        {
            std::vector<std::thread> pool;
            const int nt = std::min(chunk, 16);
            std::atomic<int> next{0};
            for (int t = 0; t < nt; ++t) pool.emplace_back([&] {
                if (opts && opts->device >= 0) (void)hipSetDevice(opts->device);
                for (int q; (q = next.fetch_add(1)) < chunk;) {
                    const int k = done + q;
                    rcs[k] = fb_stage(left[k], right[k], model[k], band ? band[k] : nullptr, opts, groups_cap, groups_cap_b, &st[k]);
                }
            });
            for (auto &th : pool) th.join();
        }
        for (int q = 0; q < chunk; ++q) if (rcs[done + q] != PAGAN_OK) rc = rcs[done + q];
        if (rc != PAGAN_OK) break;
        std::vector<PgFbJob> tiled;
        std::vector<int> tiled_k, small_k, ring_k;
        int gmax = 0, gmax_b = 0, slots_needed = 0;
        bool tiled_all_lds = true;                                // every tiled pair's score table fits LDS
        for (int q = 0; q < chunk; ++q) {
            const int k = done + q;
            if (st[k].groups > 1) {
                tiled.push_back(st[k].job); tiled_k.push_back(k);
                tiled_all_lds = tiled_all_lds && st[k].job.S * st[k].job.S <= 256;
                gmax = std::max(gmax, st[k].groups); gmax_b = std::max(gmax_b, st[k].groups_b); slots_needed += st[k].groups + st[k].groups_b;
            } else if (st[k].ring) ring_k.push_back(k);
            else small_k.push_back(k);
        }
        // the LDS-ring sweeps of the chunk: one launch per direction and workgroup size (a workgroup a pair)
        auto ring_key = [&](int k) { return 2 * st[k].block + (st[k].job.S * st[k].job.S <= 256 ? 1 : 0); };     // workgroup size, table in LDS
        std::stable_sort(ring_k.begin(), ring_k.end(), [&](int a, int b_) { return ring_key(a) < ring_key(b_); });
        std::vector<PgFbJob> ring_jobs;
        for (int k : ring_k) ring_jobs.push_back(st[k].job);
        if (rc != PAGAN_OK) break;
        PgFbJob *d_jobs = nullptr, *d_ring = nullptr;
        hipStream_t s1 = nullptr, s2 = nullptr, r1 = nullptr, r2 = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr, g0 = nullptr, g1 = nullptr, g2 = nullptr, g3 = nullptr;
        std::vector<hipStream_t> small_s(2 * small_k.size(), nullptr), ring_s;
        std::vector<hipEvent_t> ring_e;
        auto release = [&]() {
            if (d_jobs) (void)hipFree(d_jobs);
            if (d_ring) (void)hipFree(d_ring);
            for (hipStream_t x : {s1, s2, r1, r2}) if (x) (void)hipStreamDestroy(x);
            for (hipEvent_t e : {e0, e1, e2, e3, g0, g1, g2, g3}) if (e) (void)hipEventDestroy(e);
            for (hipStream_t x : small_s) if (x) (void)hipStreamDestroy(x);
            for (hipStream_t x : ring_s) if (x) (void)hipStreamDestroy(x);
            for (hipEvent_t e : ring_e) if (e) (void)hipEventDestroy(e);
        };
        auto body = [&]() -> int {
            FB_TRY(hipStreamCreate(&s1)); FB_TRY(hipStreamCreate(&s2));
            FB_TRY(hipEventCreate(&e0)); FB_TRY(hipEventCreate(&e1)); FB_TRY(hipEventCreate(&e2)); FB_TRY(hipEventCreate(&e3));
            FbSlots &slots = fb_slots_of[device & 63];
            FbSlotLease lease{&slots, slots_needed};
            if (slots_needed > 0) slots.take(slots_needed, cap);
            if (!tiled.empty()) {
                FB_TRY(hipMalloc((void **)&d_jobs, tiled.size() * sizeof(PgFbJob)));
                FB_TRY(hipMemcpy(d_jobs, tiled.data(), tiled.size() * sizeof(PgFbJob), hipMemcpyHostToDevice));
                const bool bwd_first = std::getenv("PAGAN_FB_BWD_FIRST") != nullptr;      // (A/B: which sweep's workgroups are dealt out first)
                if (bwd_first) {
                    FB_TRY(hipEventRecord(e2, s2));
                    { if (tiled_all_lds) hipLaunchKernelGGL((pg_fb_backward_tiled<true>), dim3(gmax_b, (unsigned)tiled.size()), dim3(64), 0, s2, (const PgFbJob *)d_jobs); else hipLaunchKernelGGL((pg_fb_backward_tiled<false>), dim3(gmax_b, (unsigned)tiled.size()), dim3(64), 0, s2, (const PgFbJob *)d_jobs); }
                    FB_TRY(hipEventRecord(e3, s2));
                }
                FB_TRY(hipEventRecord(e0, s1));
                { if (tiled_all_lds) hipLaunchKernelGGL((pg_fb_forward_tiled<true>), dim3(gmax, (unsigned)tiled.size()), dim3(64), 0, s1, (const PgFbJob *)d_jobs); else hipLaunchKernelGGL((pg_fb_forward_tiled<false>), dim3(gmax, (unsigned)tiled.size()), dim3(64), 0, s1, (const PgFbJob *)d_jobs); }
                FB_TRY(hipEventRecord(e1, s1));
                if (!bwd_first) {
                    FB_TRY(hipEventRecord(e2, s2));
                    { if (tiled_all_lds) hipLaunchKernelGGL((pg_fb_backward_tiled<true>), dim3(gmax_b, (unsigned)tiled.size()), dim3(64), 0, s2, (const PgFbJob *)d_jobs); else hipLaunchKernelGGL((pg_fb_backward_tiled<false>), dim3(gmax_b, (unsigned)tiled.size()), dim3(64), 0, s2, (const PgFbJob *)d_jobs); }
                    FB_TRY(hipEventRecord(e3, s2));
                }
            }
            if (!ring_jobs.empty()) {
                FB_TRY(hipStreamCreate(&r1)); FB_TRY(hipStreamCreate(&r2));
                FB_TRY(hipEventCreate(&g0)); FB_TRY(hipEventCreate(&g1)); FB_TRY(hipEventCreate(&g2)); FB_TRY(hipEventCreate(&g3));
                FB_TRY(hipMalloc((void **)&d_ring, ring_jobs.size() * sizeof(PgFbJob)));
                FB_TRY(hipMemcpy(d_ring, ring_jobs.data(), ring_jobs.size() * sizeof(PgFbJob), hipMemcpyHostToDevice));
                FB_TRY(hipEventRecord(g0, r1)); FB_TRY(hipEventRecord(g2, r2));
                // (the workgroup sizes side by side, not one after the other: the first group (size, table in LDS or not) on r1 / r2, the others -- a handful at most --
                //  on streams of their own that r1 / r2 wait for)
                for (size_t a = 0; a < ring_k.size();) {
                    size_t z = a;
                    while (z < ring_k.size() && ring_key(ring_k[z]) == ring_key(ring_k[a])) ++z;
                    const int blk = st[ring_k[a]].block;
                    const bool lds = (ring_key(ring_k[a]) & 1) != 0;
                    hipStream_t f = r1, bk = r2;
                    if (a > 0) {
                        hipStream_t x1 = nullptr, x2 = nullptr;
                        FB_TRY(hipStreamCreate(&x1)); ring_s.push_back(x1);
                        FB_TRY(hipStreamCreate(&x2)); ring_s.push_back(x2);
                        f = x1; bk = x2;
                    }
                    fb_launch_ring(true, lds, (unsigned)(z - a), blk, f, (const PgFbJob *)(d_ring + a));
                    fb_launch_ring(false, lds, (unsigned)(z - a), blk, bk, (const PgFbJob *)(d_ring + a));
                    if (a > 0) {
                        hipEvent_t ev1 = nullptr, ev2 = nullptr;
                        FB_TRY(hipEventCreateWithFlags(&ev1, hipEventDisableTiming)); ring_e.push_back(ev1);
                        FB_TRY(hipEventCreateWithFlags(&ev2, hipEventDisableTiming)); ring_e.push_back(ev2);
                        FB_TRY(hipEventRecord(ev1, f)); FB_TRY(hipStreamWaitEvent(r1, ev1, 0));
                        FB_TRY(hipEventRecord(ev2, bk)); FB_TRY(hipStreamWaitEvent(r2, ev2, 0));
                    }
                    a = z;
                }
                FB_TRY(hipEventRecord(g1, r1)); FB_TRY(hipEventRecord(g3, r2));
            }
            for (size_t q = 0; q < small_k.size(); ++q) {
                const FbStaged &x = st[small_k[q]];
                FB_TRY(hipStreamCreate(&small_s[2 * q])); FB_TRY(hipStreamCreate(&small_s[2 * q + 1]));
                hipLaunchKernelGGL(pg_fb_forward, dim3(1), dim3(x.block), 0, small_s[2 * q], x.d_job);
                hipLaunchKernelGGL(pg_fb_backward, dim3(1), dim3(x.block), 0, small_s[2 * q + 1], x.d_job);
            }
            FB_TRY(hipGetLastError());
            FB_TRY(hipStreamSynchronize(s1)); FB_TRY(hipStreamSynchronize(s2));
            if (r1) { FB_TRY(hipStreamSynchronize(r1)); FB_TRY(hipStreamSynchronize(r2)); }
            for (hipStream_t x : small_s) FB_TRY(hipStreamSynchronize(x));
            float fwd_ms = 0, bwd_ms = 0, rf_ms = 0, rb_ms = 0;
            if (!tiled.empty()) { (void)hipEventElapsedTime(&fwd_ms, e0, e1); (void)hipEventElapsedTime(&bwd_ms, e2, e3); }
            if (!ring_jobs.empty()) { (void)hipEventElapsedTime(&rf_ms, g0, g1); (void)hipEventElapsedTime(&rb_ms, g2, g3); }
            for (size_t q = 0; q < ring_k.size(); ++q) { const int r_ = fb_finish(&st[ring_k[q]], q == 0 ? rf_ms : 0.0f, q == 0 ? rb_ms : 0.0f); if (r_ != PAGAN_OK) return r_; }
            // (the launch's kernel times go to its first pair; the others report 0: a sum over the batch is the launch's)
            for (size_t q = 0; q < tiled_k.size(); ++q) { const int r_ = fb_finish(&st[tiled_k[q]], q == 0 ? fwd_ms : 0.0f, q == 0 ? bwd_ms : 0.0f); if (r_ != PAGAN_OK) return r_; }
            for (int k : small_k) { const int r_ = fb_finish(&st[k], 0.0f, 0.0f); if (r_ != PAGAN_OK) return r_; }
            return PAGAN_OK;
        };
        rc = body();
        release();
        done += chunk;
    }
    if (rc != PAGAN_OK) { cleanup(); return rc; }
    for (int k = 0; k < n; ++k) out[k] = st[k].fb;
    return PAGAN_OK;
}

int pagan_fb_kernel_ms(const pagan_fb *fb, double ms[2]) {
    if (!fb || !ms) return PAGAN_E_ARG;
    ms[0] = fb->kernel_ms[0]; ms[1] = fb->kernel_ms[1];
    return PAGAN_OK;
}

int pagan_fb_groups(const pagan_fb *fb) { return fb ? (fb->ring ? 0 : fb->groups) : PAGAN_E_ARG; }

int pagan_fb_totals(const pagan_fb *fb, double *log_fwd, double *log_bwd, int64_t *cells) {
    if (!fb) return PAGAN_E_ARG;
    if (log_fwd) *log_fwd = fb->totals[0];
    if (log_bwd) *log_bwd = fb->totals[1];
    if (cells) *cells = fb->dx.cells;
    return PAGAN_OK;
}

// which: 0 = log forward, 1 = log backward, 2 = posterior (compute_posterior_score, VA:1029-1034).
// dst [Lx][Ly][3] row-major, state order X, Y, M; outside the tunnel -inf (logs) / 0 (posterior).
int pagan_fb_dump(pagan_fb *fb, int32_t which, double *dst) {
    if (!fb || !dst || which < 0 || which > 2) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    const size_t n = 3 * (size_t)fb->dx.cells;
    std::vector<double> a(n), bb;
    FB_TRY(hipMemcpy(a.data(), which == 1 ? fb->dB : fb->dF, 8 * n, hipMemcpyDeviceToHost));
    if (which == 2) { bb.resize(n); FB_TRY(hipMemcpy(bb.data(), fb->dB, 8 * n, hipMemcpyDeviceToHost)); }
    const double outside = which == 2 ? 0.0 : -HUGE_VAL;
    for (int i = 0; i < fb->Lx; ++i)
        for (int j = 0; j < fb->Ly; ++j) {
            const long long at = fb->at(i, j);
            double *o = dst + ((size_t)i * fb->Ly + j) * 3;
            for (int s = 0; s < 3; ++s)
                o[s] = at < 0 ? outside : which == 2 ? std::exp(a[3 * at + s] + bb[3 * at + s] - fb->totals[0]) : a[3 * at + s];
        }
    return PAGAN_OK;
}

// Posterior of n cells given as (state, i, j) triples.
int pagan_fb_posterior_cells(pagan_fb *fb, int32_t n, const int32_t *cells, double *post) {
    if (!fb || n < 0 || (n > 0 && (!cells || !post))) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    for (int k = 0; k < n; ++k) {
        const int s = cells[3 * k];
        const long long at = fb->at(cells[3 * k + 1], cells[3 * k + 2]);
        if (s < 0 || s > 2) return PAGAN_E_ARG;
        if (at < 0) { post[k] = 0.0; continue; }
        double f = 0, b = 0;
        FB_TRY(hipMemcpy(&f, fb->dF + 3 * at + s, 8, hipMemcpyDeviceToHost));
        FB_TRY(hipMemcpy(&b, fb->dB + 3 * at + s, 8, hipMemcpyDeviceToHost));
        post[k] = std::exp(f + b - fb->totals[0]);
    }
    return PAGAN_OK;
}

// sample_new_path (VA:1193-1322): a path drawn from the posterior by walking back from the end corner; at every step
// the predecessors are listed in the forward pass's candidate order with weight fwd(pred) * transition (add_sample_*,
// VA:2309-2446) and the first one whose running sum reaches u * total is taken (VA:1757-1769).  u[k] in [0, 1) stands
// for rand()/(RAND_MAX+1): one number per step, the end corner first (at most Lx + Ly + 1 are used).  The result has
// the shape of a Viterbi result (columns, used edges; score = log of the full probability).  visited (optional,
// 3 * (Lx + Ly) ints): the cells of the path end -> start as (i, j, state).
int pagan_fb_sample_path(pagan_fb *fb, const double *u, int32_t n_u, pagan_result *out, int32_t *visited, int32_t *n_visited) {
    if (!fb || !u || !out) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    if (fb->hF.empty()) {
        fb->hF.resize(3 * (size_t)fb->dx.cells);
        FB_TRY(hipMemcpy(fb->hF.data(), fb->dF, 8 * fb->hF.size(), hipMemcpyDeviceToHost));
    }
    const pagan_graph *L = fb->L, *R = fb->R;
    const int Lx = fb->Lx, Ly = fb->Ly, S = fb->S;
    auto F = [&](int s, int i, int j) { const long long a = fb->at(i, j); return a < 0 ? -HUGE_VAL : fb->hF[3 * a + s]; };
    const double ext = std::log((double)fb->gap_ext), open = std::log((double)fb->gap_open), ng = std::log((double)fb->non_gap);
    struct Cand { double lw; int state, i, j, k1, k2; };
    std::vector<Cand> c;
    auto pick = [&](double uu) -> int {
        double hi = -HUGE_VAL;
        for (const Cand &x : c) hi = std::max(hi, x.lw);
        if (c.empty() || hi == -HUGE_VAL) return -1;
        double tot = 0;
        for (const Cand &x : c) tot += std::exp(x.lw - hi);
        const double rv = tot * uu;
        size_t k = 0;
        double sum = std::exp(c[0].lw - hi);
        while (sum < rv && k + 1 < c.size()) { ++k; sum += std::exp(c[k].lw - hi); }
        return (int)k;
    };
    int used = 0;
    {   // iterate_bwd_edges_for_sampled_end_corner, VA:1904-2025
        const int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1], r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        auto mt = [&](int k1, int k2) { c.push_back({F(2, L->bwd_src[k1], R->bwd_src[k2]) + ng + (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2], 2, L->bwd_src[k1], R->bwd_src[k2], k1 - l0, k2 - r0}); };
        auto xc = [&](int k1) { c.push_back({F(0, L->bwd_src[k1], Ly - 1), 0, L->bwd_src[k1], Ly - 1, k1 - l0, -1}); };
        auto yc = [&](int k2) { c.push_back({F(1, Lx - 1, R->bwd_src[k2]), 1, Lx - 1, R->bwd_src[k2], -1, k2 - r0}); };
        if (l1 > l0 && r1 > r0) {
            mt(l0, r0); xc(l0); yc(r0);
            for (int k2 = r0 + 1; k2 < r1; ++k2) { mt(l0, k2); yc(k2); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) { mt(k1, r0); xc(k1); for (int k2 = r0 + 1; k2 < r1; ++k2) { mt(k1, k2); yc(k2); } }
        }
    }
    if (used >= n_u) return PAGAN_E_ARG;
    int k = pick(u[used++]);
    int endcell[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < 0) {                                     // full probability 0: nothing to sample
        endcell[0] = 1; endcell[4] = endcell[5] = -1;
        return pagan_internal_replay(L, R, fb->dx.cells, endcell, fb->totals[0], nullptr, out);
    }
    int state = c[k].state, i = c[k].i, j = c[k].j;
    endcell[1] = state; endcell[2] = i; endcell[3] = j; endcell[4] = c[k].k1; endcell[5] = c[k].k2;
    std::vector<int> trace;
    trace.reserve(3 * ((size_t)Lx + Ly));
    while (!(i < 1 && j < 1)) {
        c.clear();
        if (state == 2) {
            const double sc = std::log((double)fb->score[L->state[i] + (size_t)R->state[j] * S]);
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; ++k1)
                for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; ++k2) {
                    const int p = L->bwd_src[k1], q = R->bwd_src[k2];
                    const double w = (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2];
                    const int a = k1 - L->bwd_off[i], b = k2 - R->bwd_off[j];
                    c.push_back({F(2, p, q) + ng + ng + sc + w, 2, p, q, a, b});
                    c.push_back({F(0, p, q) + ng + sc + w, 0, p, q, a, b});
                    c.push_back({F(1, p, q) + ng + sc + w, 1, p, q, a, b});
                }
        } else if (state == 0) {
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; ++k1) {
                const int p = L->bwd_src[k1], a = k1 - L->bwd_off[i];
                c.push_back({F(0, p, j) + ext, 0, p, j, a, 0}); c.push_back({F(1, p, j) + open, 1, p, j, a, 0});
                c.push_back({F(2, p, j) + ng + open, 2, p, j, a, 0});
            }
        } else {
            for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; ++k2) {
                const int q = R->bwd_src[k2], b = k2 - R->bwd_off[j];
                c.push_back({F(1, i, q) + ext, 1, i, q, 0, b}); c.push_back({F(0, i, q) + open, 0, i, q, 0, b});
                c.push_back({F(2, i, q) + ng + open, 2, i, q, 0, b});
            }
        }
        if (used >= n_u) return PAGAN_E_ARG;
        k = pick(u[used++]);
        if (k < 0 || (size_t)trace.size() >= 3 * ((size_t)Lx + Ly)) return PAGAN_E_INTERNAL;
        trace.push_back(i); trace.push_back(j);
        trace.push_back((int)((unsigned)state | ((unsigned)c[k].k1 << 4) | ((unsigned)c[k].k2 << 18)));
        state = c[k].state; i = c[k].i; j = c[k].j;
    }
    endcell[6] = (int)(trace.size() / 3);
    if (visited) for (size_t t = 0; t < trace.size() / 3; ++t) { visited[3 * t] = trace[3 * t]; visited[3 * t + 1] = trace[3 * t + 1]; visited[3 * t + 2] = trace[3 * t + 2] & 3; }
    if (n_visited) *n_visited = endcell[6];
    trace.resize(trace.size() + 3, 0);
    return pagan_internal_replay(L, R, fb->dx.cells, endcell, fb->totals[0], trace.data(), out);
}

void pagan_fb_destroy(pagan_fb *fb) {
    if (!fb) return;
    if (fb->arena) fb_arena_pool.give(fb->device, fb->arena, fb->arena_cap);
    delete fb;
}

} // extern "C"
