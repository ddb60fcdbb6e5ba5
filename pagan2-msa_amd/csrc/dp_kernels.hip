// dp_kernels.hip -- gfx950 kernels for the pairwise graph-vs-graph Viterbi DP.
//
// The recurrence (reference: Viterbi_alignment::compute_fwd_scores,
// src/main/viterbi_alignment.cpp:856-971, with iterate_bwd_edges_for_gap VA:1328-1349,
// iterate_bwd_edges_for_match VA:1353-1436 and score_* VA:2029-2219) is a scalar fp64
// max-plus over the incoming graph edges of the two sites -- no MFMA shape in it.  Every
// predecessor of cell (i,j) lies on an earlier anti-diagonal, so the fill sweeps d = i+j as
// a wavefront, one lane per in-band cell of the diagonal.
//
// Bit-exactness rules kept here (they decide the traceback through fp64 ties):
//   - every float parameter is promoted separately and added left to right, exactly as the
//     C++ expressions in the reference evaluate: (s + f1) + f2;
//   - `2*log_non_gap` and `0 + log_non_gap` are FLOAT operations (VA:1364-1367);
//   - a candidate replaces the incumbent only if strictly greater (first_is_bigger,
//     src/main/basic_alignment.h:449-462), candidates in the reference's order.
// Compiled with -ffp-contract=off; there are no multiplies to fuse on the fp64 path anyway.
//
// Code-shape rule of this file: nothing takes the address of a job descriptor or of an LDS
// object and stores it; job fields are read into scalars, LDS is reached through the
// __shared__ symbol.  (A pointer parked in a struct made hipcc fall back to flat_* accesses
// and scratch, and every flat access waits on vmcnt(0), i.e. on all earlier HBM stores.)
#include <hip/hip_runtime.h>
#include "dp_device.h"

#define PG_X 0
#define PG_Y 1
#define PG_M 2

namespace {

__device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }

__device__ __forceinline__ unsigned pack_bp(unsigned from, int k1, int k2) {
    return from | ((unsigned)k1 << 2) | ((unsigned)k2 << 17);
}

// Pointers read out of a descriptor in memory are generic to the compiler (flat_* accesses,
// which count on vmcnt AND lgkmcnt and so serialise against LDS traffic and pending stores);
// they all point into the batch's HBM arena, so say so.
#define PG_GLOBAL __attribute__((address_space(1)))
typedef PG_GLOBAL const int *gint_p;
typedef PG_GLOBAL const float *gfloat_p;
typedef PG_GLOBAL const long long *gll_p;
typedef PG_GLOBAL double *gdouble_w;
typedef PG_GLOBAL unsigned *gu32_w;
typedef PG_GLOBAL int *gint_w;

// Job descriptor flattened to scalars (SGPRs): no arrays, never address-taken.
struct View {
    int Lx, Ly, nd, S;
    float go, ge, gE, ng;
    gint_p stL, offL, srcL; gfloat_p lwL;
    gint_p stR, offR, srcR; gfloat_p lwR;
    gfloat_p table;
    gint_p imin, imax; gll_p doff;
    gdouble_w sc;            // [cells][3]  X, Y, M
    gu32_w bp;               // [cells][3]
    gint_w trace, endcell; gdouble_w endscore;
};

__device__ __forceinline__ View load_view(const PgDevJob *__restrict__ j) {
    View v;
    v.Lx = j->Lx; v.Ly = j->Ly; v.nd = j->nd; v.S = j->S;
    v.go = j->go; v.ge = j->ge; v.gE = j->gE; v.ng = j->ng;
    v.stL = (gint_p)j->stL; v.offL = (gint_p)j->offL; v.srcL = (gint_p)j->srcL; v.lwL = (gfloat_p)j->lwL;
    v.stR = (gint_p)j->stR; v.offR = (gint_p)j->offR; v.srcR = (gint_p)j->srcR; v.lwR = (gfloat_p)j->lwR;
    v.table = (gfloat_p)j->table; v.imin = (gint_p)j->imin; v.imax = (gint_p)j->imax; v.doff = (gll_p)j->doff;
    v.sc = (gdouble_w)j->sc; v.bp = (gu32_w)j->bp;
    v.trace = (gint_w)j->trace; v.endcell = (gint_w)j->endcell; v.endscore = (gdouble_w)j->endscore;
    return v;
}

// One cell's results: 24 B of scores + 12 B of back-pointers, contiguous per cell.
__device__ __forceinline__ void store_cell(gdouble_w sc, gu32_w bp, long long at, double bx, double by, double bm,
                                           unsigned px, unsigned py, unsigned pm) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    gdouble_w s = sc + 3 * at;
    d2 xy; xy.x = bx; xy.y = by;
    *(PG_GLOBAL d2 *)s = xy;           // 16 B, 8-byte aligned (global stores need dword alignment only)
    s[2] = bm;
    u3 b; b.x = px; b.y = py; b.z = pm;
    *(PG_GLOBAL u3 *)(bp + 3 * at) = b;
}

// Band interval + storage offset of one anti-diagonal.
struct Diag { int mn, mx; long long off; };

// Linear HBM index of cell (p,q) or -1 outside the tunnel (Tunnel_slice::at returns the shared
// -inf entry there, src/utils/tunnel_matrix.h:85-98).  d1/d2 = descriptors of diagonals d-1, d-2.
__device__ __forceinline__ long long hbm_index(const View &J, int d, const Diag &d1, const Diag &d2, int p, int q) {
    const int dd = p + q;
    int mn, mx; long long off;
    if (dd == d - 1) { mn = d1.mn; mx = d1.mx; off = d1.off; }
    else if (dd == d - 2) { mn = d2.mn; mx = d2.mx; off = d2.off; }
    else { mn = J.imin[dd]; mx = J.imax[dd]; off = J.doff[dd]; }
    return (p >= mn && p <= mx) ? off + (p - mn) : -1;
}

// One DP cell with every operand in HBM/L2: the three states of (i,j), written at `at`.
__device__ __forceinline__ void fill_cell_hbm(const View &J, int d, const Diag &d1, const Diag &d2, int i, int j,
                                              long long at, bool no_terminal_edges, bool reduced_terminal) {
    const double NI = neg_inf();
    double bx = NI, by = NI, bm = NI;
    unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
    if (i == 0 && j == 0) {
        bm = 0.0;                                   // initialise_array_corner, VA:725-736
    } else {
        const double go = (double)J.go, ng = (double)J.ng;
        int l0 = 0, l1 = 0, r0 = 0, r1 = 0;
        if (i > 0) { l0 = J.offL[i]; l1 = J.offL[i + 1]; }
        if (j > 0) { r0 = J.offR[j]; r1 = J.offR[j + 1]; }
        // ---- X: gap in the right sequence, consumes left site i (VA:898-915) ----
        if (i > 0) {
            const bool end_gap = (j == 0 || j == J.Ly - 1) && !no_terminal_edges;   // VA:864-868
            const double ext = (double)(end_gap ? J.gE : J.ge);
            for (int e = l0; e < l1; ++e) {
                const int p = J.srcL[e];
                const long long ix = hbm_index(J, d, d1, d2, p, j);
                double xs = NI, ys = NI, ms = NI;
                if (ix >= 0) { xs = J.sc[3 * ix + PG_X]; ys = J.sc[3 * ix + PG_Y]; ms = J.sc[3 * ix + PG_M]; }
                const double open = (reduced_terminal && p == 0) ? 0.0 : go;        // BA.h:490-513
                double c = xs + ext;                                                 // score_gap_ext
                if (c > bx) { bx = c; px = pack_bp(PG_X, e - l0, 0); }
                c = (ys + 0.0) + go;                                                 // score_gap_double
                if (c > bx) { bx = c; px = pack_bp(PG_Y, e - l0, 0); }
                c = (ms + ng) + open;                                                // score_gap_open
                if (c > bx) { bx = c; px = pack_bp(PG_M, e - l0, 0); }
            }
        }
        // ---- Y: gap in the left sequence, consumes right site j (VA:927-944) ----
        if (j > 0) {
            const bool end_gap = (i == 0 || i == J.Lx - 1) && !no_terminal_edges;   // VA:875-879
            const double ext = (double)(end_gap ? J.gE : J.ge);
            for (int e = r0; e < r1; ++e) {
                const int q = J.srcR[e];
                const long long ix = hbm_index(J, d, d1, d2, i, q);
                double xs = NI, ys = NI, ms = NI;
                if (ix >= 0) { xs = J.sc[3 * ix + PG_X]; ys = J.sc[3 * ix + PG_Y]; ms = J.sc[3 * ix + PG_M]; }
                const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                double c = ys + ext;
                if (c > by) { by = c; py = pack_bp(PG_Y, 0, e - r0); }
                c = (xs + 0.0) + go;
                if (c > by) { by = c; py = pack_bp(PG_X, 0, e - r0); }
                c = (ms + ng) + open;
                if (c > by) { by = c; py = pack_bp(PG_M, 0, e - r0); }
            }
        }
        // ---- M: both sites consumed (VA:956-963, 1353-1436) ----
        if (i > 0 && j > 0 && l1 > l0 && r1 > r0) {
            const float sm = J.table[J.stL[i] + J.stR[j] * J.S];                    // VA:1363
            const double tM = (double)(2 * J.ng) + (double)sm;                      // VA:1364
            const double tX = (double)(0.0f + J.ng) + (double)sm;                   // VA:1366-1367
            for (int e1 = l0; e1 < l1; ++e1) {
                const int p = J.srcL[e1];
                const double lw = (double)J.lwL[e1];
                for (int e2 = r0; e2 < r1; ++e2) {
                    const int q = J.srcR[e2];
                    const double rw = (double)J.lwR[e2];
                    const long long ix = hbm_index(J, d, d1, d2, p, q);
                    double xs = NI, ys = NI, ms = NI;
                    if (ix >= 0) { xs = J.sc[3 * ix + PG_X]; ys = J.sc[3 * ix + PG_Y]; ms = J.sc[3 * ix + PG_M]; }
                    double c = ((ms + tM) + lw) + rw;                                // score_m_match
                    if (c > bm) { bm = c; pm = pack_bp(PG_M, e1 - l0, e2 - r0); }
                    c = ((xs + tX) + lw) + rw;                                       // score_x_match
                    if (c > bm) { bm = c; pm = pack_bp(PG_X, e1 - l0, e2 - r0); }
                    c = ((ys + tX) + lw) + rw;                                       // score_y_match
                    if (c > bm) { bm = c; pm = pack_bp(PG_Y, e1 - l0, e2 - r0); }
                }
            }
        }
    }
    store_cell(J.sc, J.bp, at, bx, by, bm, px, py, pm);
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Wide wavefront: scores exchanged through HBM/L2.  One workgroup per alignment, each diagonal
// one block-stride pass followed by a workgroup barrier whose release/acquire makes the stores
// visible to the next diagonal's loads.  Used for full (--no-anchors) matrices, whose
// diagonals are hundreds to thousands of cells wide.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pg_fill_wavefront(const PgDevJob *__restrict__ jobs,
                                                           const int *__restrict__ which, unsigned flags) {
    const View J = load_view(jobs + which[blockIdx.x]);
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    Diag d1 = {0, -1, 0}, d2 = {0, -1, 0};
    for (int d = 0; d < J.nd; ++d) {
        const int lo = J.imin[d], hi = J.imax[d];
        const long long base = J.doff[d];
        for (int i = lo + (int)threadIdx.x; i <= hi; i += BLOCK)
            fill_cell_hbm(J, d, d1, d2, i, d - i, base + (i - lo), no_terminal_edges, reduced_terminal);
        d2 = d1;
        d1 = {lo, hi, base};
        __syncthreads();
    }
}

template __global__ void pg_fill_wavefront<64>(const PgDevJob *, const int *, unsigned);
template __global__ void pg_fill_wavefront<256>(const PgDevJob *, const int *, unsigned);
template __global__ void pg_fill_wavefront<1024>(const PgDevJob *, const int *, unsigned);

// ---------------------------------------------------------------------------------------------
// Banded wavefront with the active band staged in LDS ("ring" kernel).
//
// One workgroup of two wave64 per alignment: a COMPUTE wave and a LOADER wave.
//
// Compute wave: lane l owns the rows i with i % 64 == l that are inside the band on the
// current anti-diagonal -- one row on most diagonals (anchors-offset 15 gives a median of 16
// cells per diagonal, a 99th percentile near 100), a few on the wide ones.  Per step:
//   - the scores of the last RK diagonals live in an LDS ring  sc[d % RK][i % WMAX][M,X,Y];
//     every predecessor within RK diagonals is an LDS read guarded by that diagonal's
//     [imin,imax] interval;
//   - site data, bwd edges and the per-diagonal band index are read from LDS rings that the
//     loader wave keeps filled ahead of the band;
//   - scores and back-pointers stream to HBM with coalesced stores nobody waits for.
//   In the steady state this wave issues NO vector-memory load: vmcnt returns in order, so a
//   single load would stall on every store still in flight to HBM (measured: 5.6 us per
//   diagonal with the loads in this wave, see profiles/).
//   - a graph edge reaching >= RK diagonals back reads what this same wave stored to HBM
//     earlier (a wave's vector memory operations are performed in order);
//   - a diagonal wider than WMAX (a box between distant anchors) is computed from HBM
//     operands like the wide kernel does, and marked "not in the ring".
// Loader wave: coalesced loads of the next 64 left sites / right sites (+ their bwd edges) /
// diagonal descriptors into the rings whenever the compute wave's published progress leaves
// room; publishes "loaded up to" counters after its LDS writes have landed.
// The two waves meet once at the start (__syncthreads) and then only through those counters.
#define RK 8
#define WMAX 256
#define RW 512
#define EC 2048
#define DR 256
// bit set in the cached state word of a site whose only bwd edge comes from its predecessor
// site with weight 1 (log-weight +0.0): adding that weight is an exact no-op
#define PG_SIMPLE 0x10000

struct RingSmem {
    double sc[RK][WMAX][3];                 // X, Y, M
    int dmn[RK], dmx[RK], did[RK];          // per slot: band interval and WHICH diagonal it holds (-1: none)
    int stL[RW], ebL[RW], eeL[RW];
    int stR[RW], ebR[RW], eeR[RW];
    int esL[EC]; float ewL[EC];
    int esR[EC]; float ewR[EC];
    int dlo[DR], dhi[DR]; long long dbase[DR];
    float table[256];
    int rows_loaded, cols_loaded, diags_loaded;     // loader -> compute
    int prog_row, prog_col, prog_d;                 // compute -> loader: lowest row / column / diagonal still needed
};

unsigned pg_ring_lds_bytes() { return (unsigned)sizeof(RingSmem); }

extern __shared__ __attribute__((aligned(16))) char pg_ring_lds[];
#define SM (*reinterpret_cast<RingSmem *>(pg_ring_lds))
// Cross-wave counters: relaxed workgroup-scope atomics on the LDS symbol (plain ds_read/ds_write
// that the compiler neither caches nor fences; a `volatile` cast turned into flat accesses).
#define SM_GET(field) __hip_atomic_load(&SM.field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SM_PUT(field, v) __hip_atomic_store(&SM.field, (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// Loads the COMPUTE wave needs only on rare paths (an edge reaching past the ring).  Issued as
// inline asm with their own wait so that the compiler's waitcnt insertion never places a
// vmcnt(0) -- which would also wait for every store in flight -- on the common path.
__device__ __forceinline__ double far_f64(PG_GLOBAL const double *p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int far_i32(PG_GLOBAL const int *p) {
    int v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float far_f32(PG_GLOBAL const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ long long far_i64(PG_GLOBAL const long long *p) {
    long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

namespace {

// Scores of cell (p,q) on an earlier diagonal; -inf outside the tunnel.
__device__ __forceinline__ void ring_load(const View &J, int d, int mn1, int mx1, bool in1, int mn2, int mx2, bool in2,
                                          int p, int q, double &ms, double &xs, double &ys) {
    const double NI = neg_inf();
    const int dd = p + q, age = d - dd;
    ms = NI; xs = NI; ys = NI;
    bool ring;
    int mn, mx;
    if (age == 1) { ring = in1; mn = mn1; mx = mx1; }
    else if (age == 2) { ring = in2; mn = mn2; mx = mx2; }
    else {
        ring = age < RK && SM.did[dd & (RK - 1)] == dd;
        if (ring) { mn = SM.dmn[dd & (RK - 1)]; mx = SM.dmx[dd & (RK - 1)]; }
        else { mn = far_i32(J.imin + dd); mx = far_i32(J.imax + dd); }
    }
    if (p < mn || p > mx) return;
    if (ring) {
        xs = SM.sc[dd & (RK - 1)][p & (WMAX - 1)][PG_X];
        ys = SM.sc[dd & (RK - 1)][p & (WMAX - 1)][PG_Y];
        ms = SM.sc[dd & (RK - 1)][p & (WMAX - 1)][PG_M];
    } else {
        const long long ix = far_i64(J.doff + dd) + (p - mn);
        xs = far_f64(J.sc + 3 * ix + PG_X); ys = far_f64(J.sc + 3 * ix + PG_Y); ms = far_f64(J.sc + 3 * ix + PG_M);
    }
}

// Loader wave: keeps the LDS rings ahead of the compute wave.  Every ring entry below the
// published progress is dead and may be overwritten.
__device__ __forceinline__ void ring_loader(const View &J, int lane) {
    int rows = 0, cols = 0, diags = 0;
    while (rows < J.Lx || cols < J.Ly || diags < J.nd) {
        bool progressed = false;
        const int p_row = SM_GET(prog_row), p_col = SM_GET(prog_col), p_d = SM_GET(prog_d);
        if (diags < J.nd && diags + 64 - p_d <= DR) {
            const int dd = diags + lane;
            if (dd < J.nd) { SM.dlo[dd & (DR - 1)] = J.imin[dd]; SM.dhi[dd & (DR - 1)] = J.imax[dd]; SM.dbase[dd & (DR - 1)] = J.doff[dd]; }
            diags += 64;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) SM_PUT(diags_loaded, diags);
            progressed = true;
        }
        if (rows < J.Lx && rows + 64 - p_row <= RW) {
            const int rend = rows + 64 < J.Lx ? rows + 64 : J.Lx;
            const int e0 = J.offL[rows], e1 = J.offL[rend];
            if (e1 - J.offL[p_row < J.Lx ? p_row : J.Lx] <= EC) {
                const int r = rows + lane;
                if (r < J.Lx) {
                    const int b = J.offL[r], en = J.offL[r + 1];
                    int st = J.stL[r] & 0xffff;
                    if (r > 0 && en - b == 1 && J.srcL[b] == r - 1 && J.lwL[b] == 0.0f) st |= PG_SIMPLE;
                    SM.stL[r & (RW - 1)] = st; SM.ebL[r & (RW - 1)] = b; SM.eeL[r & (RW - 1)] = en;
                }
                for (int e = e0 + lane; e < e1; e += 64) { SM.esL[e & (EC - 1)] = J.srcL[e]; SM.ewL[e & (EC - 1)] = J.lwL[e]; }
                rows += 64;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (lane == 0) SM_PUT(rows_loaded, rows);
                progressed = true;
            }
        }
        if (cols < J.Ly && cols + 64 - p_col <= RW) {
            const int rend = cols + 64 < J.Ly ? cols + 64 : J.Ly;
            const int e0 = J.offR[cols], e1 = J.offR[rend];
            if (e1 - J.offR[p_col < J.Ly ? p_col : J.Ly] <= EC) {
                const int r = cols + lane;
                if (r < J.Ly) {
                    const int b = J.offR[r], en = J.offR[r + 1];
                    int st = J.stR[r] & 0xffff;
                    if (r > 0 && en - b == 1 && J.srcR[b] == r - 1 && J.lwR[b] == 0.0f) st |= PG_SIMPLE;
                    SM.stR[r & (RW - 1)] = st; SM.ebR[r & (RW - 1)] = b; SM.eeR[r & (RW - 1)] = en;
                }
                for (int e = e0 + lane; e < e1; e += 64) { SM.esR[e & (EC - 1)] = J.srcR[e]; SM.ewR[e & (EC - 1)] = J.lwR[e]; }
                cols += 64;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (lane == 0) SM_PUT(cols_loaded, cols);
                progressed = true;
            }
        }
        if (!progressed) __builtin_amdgcn_s_sleep(16);
    }
}

} // namespace

__global__ __launch_bounds__(128) void pg_fill_ring(const PgDevJob *__restrict__ jobs, const int *__restrict__ which,
                                                    unsigned flags) {
    const View J = load_view(jobs + which[blockIdx.x]);
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const int lane = threadIdx.x & 63;
    const bool tab_lds = J.S <= 16;
    if (tab_lds) for (int k = threadIdx.x; k < J.S * J.S; k += 128) SM.table[k] = J.table[k];
    for (int k = threadIdx.x; k < RK; k += 128) { SM.dmn[k] = 0; SM.dmx[k] = -1; SM.did[k] = -1; }
    if (threadIdx.x == 0) {
        SM.rows_loaded = 0; SM.cols_loaded = 0; SM.diags_loaded = 0;
        SM.prog_row = 0; SM.prog_col = 0; SM.prog_d = 0;
    }
    __syncthreads();
    if (threadIdx.x >= 64) { ring_loader(J, lane); return; }

    const double NI = neg_inf();
    int mn1 = 0, mx1 = -1, mn2 = 0, mx2 = -1;
    bool in1 = false, in2 = false;
    Diag g1 = {0, -1, 0}, g2 = {0, -1, 0};
    const double go = (double)J.go, ng = (double)J.ng;
    const double tng2 = (double)(2 * J.ng), tng1 = (double)(0.0f + J.ng);
    int rows_ok = 0, cols_ok = 0, diags_ok = 0;          // cached copies of the loader's counters
    for (int d = 0; d < J.nd; ++d) {
        while (diags_ok <= d) { diags_ok = SM_GET(diags_loaded); if (diags_ok <= d) __builtin_amdgcn_s_sleep(2); }
        asm volatile("" ::: "memory");      // ring reads stay behind the counter they depend on
        const int lo = SM.dlo[d & (DR - 1)], hi = SM.dhi[d & (DR - 1)];
        const long long base = SM.dbase[d & (DR - 1)];
        const bool wide = hi - lo + 1 > WMAX;
        if (lane == 0 && (d & 7) == 0) { SM_PUT(prog_row, lo); SM_PUT(prog_col, d - hi > 0 ? d - hi : 0); SM_PUT(prog_d, d); }
        if (wide) {
            // rare: a box between anchors wider than the ring -- every operand from HBM/L2
            for (int i = lo + lane; i <= hi; i += 64)
                fill_cell_hbm(J, d, g1, g2, i, d - i, base + (i - lo), no_terminal_edges, reduced_terminal);
        } else {
            if (hi >= lo) {
                while (rows_ok <= hi && rows_ok < J.Lx) { rows_ok = SM_GET(rows_loaded); if (rows_ok <= hi && rows_ok < J.Lx) __builtin_amdgcn_s_sleep(2); }
                while (cols_ok <= d - lo && cols_ok < J.Ly) { cols_ok = SM_GET(cols_loaded); if (cols_ok <= d - lo && cols_ok < J.Ly) __builtin_amdgcn_s_sleep(2); }
                asm volatile("" ::: "memory");
            }
            // ---- fast path: every cell of this diagonal is "simple" and both previous diagonals are
            // in the ring.  Straight-line code: three 24-byte LDS reads, nine candidates.
            const int i0 = lo + ((lane - lo) & 63);
            bool simple = true;
            int wi = 0, wj = 0;
            if (i0 <= hi) {
                wi = SM.stL[i0 & (RW - 1)]; wj = SM.stR[(d - i0) & (RW - 1)];
                simple = (wi & wj & PG_SIMPLE) != 0;      // implies i0 >= 1 and j >= 1
            }
            if (hi - lo < 64 && in1 && in2 && __all(simple)) {
                if (i0 <= hi) {
                    const int i = i0, j = d - i0;
                    const int s1 = (d - 1) & (RK - 1), s2 = (d - 2) & (RK - 1);
                    const bool inA = i - 1 >= mn1 && i - 1 <= mx1;       // (i-1, j)   on d-1
                    const bool inB = i >= mn1 && i <= mx1;               // (i, j-1)   on d-1
                    const bool inC = i - 1 >= mn2 && i - 1 <= mx2;       // (i-1, j-1) on d-2
                    const double *A = SM.sc[s1][(i - 1) & (WMAX - 1)];
                    const double *B = SM.sc[s1][i & (WMAX - 1)];
                    const double *Cc = SM.sc[s2][(i - 1) & (WMAX - 1)];
                    // unconditional reads (the slots always exist), selected afterwards
                    const double a0 = A[PG_X], a1 = A[PG_Y], a2 = A[PG_M];
                    const double b0 = B[PG_X], b1 = B[PG_Y], b2 = B[PG_M];
                    const double c0 = Cc[PG_X], c1 = Cc[PG_Y], c2 = Cc[PG_M];
                    const double xA = inA ? a0 : NI, yA = inA ? a1 : NI, mA = inA ? a2 : NI;
                    const double xB = inB ? b0 : NI, yB = inB ? b1 : NI, mB = inB ? b2 : NI;
                    const double xC = inC ? c0 : NI, yC = inC ? c1 : NI, mC = inC ? c2 : NI;
                    const int ti = (wi & 0xffff) + (wj & 0xffff) * J.S;
                    const float smf = tab_lds ? SM.table[ti] : far_f32(J.table + ti);
                    const double extX = (double)((j == J.Ly - 1 && !no_terminal_edges) ? J.gE : J.ge);
                    const double extY = (double)((i == J.Lx - 1 && !no_terminal_edges) ? J.gE : J.ge);
                    const double openX = (reduced_terminal && i == 1) ? 0.0 : go;
                    const double openY = (reduced_terminal && j == 1) ? 0.0 : go;
                    double bx = NI, by = NI, bm = NI, c;
                    unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
                    c = xA + extX;          if (c > bx) { bx = c; px = PG_X; }
                    c = (yA + 0.0) + go;    if (c > bx) { bx = c; px = PG_Y; }
                    c = (mA + ng) + openX;  if (c > bx) { bx = c; px = PG_M; }
                    c = yB + extY;          if (c > by) { by = c; py = PG_Y; }
                    c = (xB + 0.0) + go;    if (c > by) { by = c; py = PG_X; }
                    c = (mB + ng) + openY;  if (c > by) { by = c; py = PG_M; }
                    const double tM = tng2 + (double)smf, tX = tng1 + (double)smf;
                    c = mC + tM;            if (c > bm) { bm = c; pm = PG_M; }     // + 0.0 + 0.0 (log-weights) omitted: exact
                    c = xC + tX;            if (c > bm) { bm = c; pm = PG_X; }
                    c = yC + tX;            if (c > bm) { bm = c; pm = PG_Y; }
                    SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_X] = bx;
                    SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_Y] = by;
                    SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_M] = bm;
                    store_cell(J.sc, J.bp, base + (i - lo), bx, by, bm, px, py, pm);
                }
            } else
            for (int i = lo + ((lane - lo) & 63); i <= hi; i += 64) {
                const int j = d - i;
                double bx = NI, by = NI, bm = NI;
                unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
                if (i == 0 && j == 0) {
                    bm = 0.0;
                } else {
                    int l0 = 0, l1 = 0, r0 = 0, r1 = 0;
                    if (i > 0) { l0 = SM.ebL[i & (RW - 1)]; l1 = SM.eeL[i & (RW - 1)]; }
                    if (j > 0) { r0 = SM.ebR[j & (RW - 1)]; r1 = SM.eeR[j & (RW - 1)]; }
                    if (i > 0) {
                        const bool end_gap = (j == 0 || j == J.Ly - 1) && !no_terminal_edges;
                        const double ext = (double)(end_gap ? J.gE : J.ge);
                        for (int e = l0; e < l1; ++e) {
                            const int p = SM.esL[e & (EC - 1)];
                            double ms, xs, ys;
                            ring_load(J, d, mn1, mx1, in1, mn2, mx2, in2, p, j, ms, xs, ys);
                            const double open = (reduced_terminal && p == 0) ? 0.0 : go;
                            double c = xs + ext;
                            if (c > bx) { bx = c; px = pack_bp(PG_X, e - l0, 0); }
                            c = (ys + 0.0) + go;
                            if (c > bx) { bx = c; px = pack_bp(PG_Y, e - l0, 0); }
                            c = (ms + ng) + open;
                            if (c > bx) { bx = c; px = pack_bp(PG_M, e - l0, 0); }
                        }
                    }
                    if (j > 0) {
                        const bool end_gap = (i == 0 || i == J.Lx - 1) && !no_terminal_edges;
                        const double ext = (double)(end_gap ? J.gE : J.ge);
                        for (int e = r0; e < r1; ++e) {
                            const int q = SM.esR[e & (EC - 1)];
                            double ms, xs, ys;
                            ring_load(J, d, mn1, mx1, in1, mn2, mx2, in2, i, q, ms, xs, ys);
                            const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                            double c = ys + ext;
                            if (c > by) { by = c; py = pack_bp(PG_Y, 0, e - r0); }
                            c = (xs + 0.0) + go;
                            if (c > by) { by = c; py = pack_bp(PG_X, 0, e - r0); }
                            c = (ms + ng) + open;
                            if (c > by) { by = c; py = pack_bp(PG_M, 0, e - r0); }
                        }
                    }
                    if (i > 0 && j > 0 && l1 > l0 && r1 > r0) {
                        const int ti = (SM.stL[i & (RW - 1)] & 0xffff) + (SM.stR[j & (RW - 1)] & 0xffff) * J.S;
                        const float smf = tab_lds ? SM.table[ti] : far_f32(J.table + ti);
                        const double tM = tng2 + (double)smf;
                        const double tX = tng1 + (double)smf;
                        for (int e1 = l0; e1 < l1; ++e1) {
                            const int p = SM.esL[e1 & (EC - 1)];
                            const double lw = (double)SM.ewL[e1 & (EC - 1)];
                            for (int e2 = r0; e2 < r1; ++e2) {
                                const int q = SM.esR[e2 & (EC - 1)];
                                const double rw = (double)SM.ewR[e2 & (EC - 1)];
                                double ms, xs, ys;
                                ring_load(J, d, mn1, mx1, in1, mn2, mx2, in2, p, q, ms, xs, ys);
                                double c = ((ms + tM) + lw) + rw;
                                if (c > bm) { bm = c; pm = pack_bp(PG_M, e1 - l0, e2 - r0); }
                                c = ((xs + tX) + lw) + rw;
                                if (c > bm) { bm = c; pm = pack_bp(PG_X, e1 - l0, e2 - r0); }
                                c = ((ys + tX) + lw) + rw;
                                if (c > bm) { bm = c; pm = pack_bp(PG_Y, e1 - l0, e2 - r0); }
                            }
                        }
                    }
                }
                SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_X] = bx;
                SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_Y] = by;
                SM.sc[d & (RK - 1)][i & (WMAX - 1)][PG_M] = bm;
                store_cell(J.sc, J.bp, base + (i - lo), bx, by, bm, px, py, pm);
            }
        }
        if (lane == 0) { SM.dmn[d & (RK - 1)] = lo; SM.dmx[d & (RK - 1)] = hi; SM.did[d & (RK - 1)] = wide ? -1 : d; }
        mn2 = mn1; mx2 = mx1; in2 = in1; mn1 = lo; mx1 = hi; in1 = !wide;
        g2 = g1; g1 = {lo, hi, base};
        __builtin_amdgcn_wave_barrier();
    }
    // let the loader finish whatever it was still allowed to prefetch
    if (lane == 0) { SM_PUT(prog_row, J.Lx); SM_PUT(prog_col, J.Ly); SM_PUT(prog_d, J.nd); }
}

// ---------------------------------------------------------------------------------------------
// End corner (iterate_bwd_edges_for_end_corner, VA:1440-1552, score_gap_close VA:2221-2255)
// and the pointer chase of backtrack_new_path (VA:1038-1189).  One lane per alignment: the
// chase is a serial dependency chain.  It emits the visited cells as (i, j, w) with
// w = the cell's own matrix in bits 0-1 and its two edge slots in bits 2-31 (the cell's
// `from` label is the next entry's matrix); skip columns and used-edge marks are derived
// from that list on the host.
__global__ void pg_end_and_trace(const PgDevJob *__restrict__ jobs) {
    if (threadIdx.x != 0) return;
    const View J = load_view(jobs + blockIdx.x);
    const double NI = neg_inf();
    const Diag none = {0, -1, 0};
    const int Lx = J.Lx, Ly = J.Ly;
    const int l0 = J.offL[Lx], l1 = J.offL[Lx + 1];
    const int r0 = J.offR[Ly], r1 = J.offR[Ly + 1];

    double best = NI;          // max->score
    int mat = -1, xi = -1, yi = -1, kl = -1, kr = -1;   // max->matrix, x_ind, y_ind, edge slots (-1 = none)
    if (l1 > l0 && r1 > r0) {
        const double ng = (double)J.ng;
        auto m_cand = [&](int e1, int e2) {
            const int p = J.srcL[e1], q = J.srcR[e2];
            const long long ix = hbm_index(J, -10, none, none, p, q);
            const double ms = ix >= 0 ? J.sc[3 * ix + PG_M] : NI;
            const double c = ((ms + ng) + (double)J.lwL[e1]) + (double)J.lwR[e2];
            if (c > best) { best = c; mat = PG_M; xi = p; yi = q; kl = e1 - l0; kr = e2 - r0; }
        };
        auto x_close = [&](int e1) {
            const int p = J.srcL[e1];
            const long long ix = hbm_index(J, -10, none, none, p, Ly - 1);
            const double c = (ix >= 0 ? J.sc[3 * ix + PG_X] : NI) + 0.0;
            if (c > best) { best = c; mat = PG_X; xi = p; kl = e1 - l0; kr = -1; yi = Ly - 1; }
        };
        auto y_close = [&](int e2) {
            const int q = J.srcR[e2];
            const long long ix = hbm_index(J, -10, none, none, Lx - 1, q);
            const double c = (ix >= 0 ? J.sc[3 * ix + PG_Y] : NI) + 0.0;
            if (c > best) { best = c; mat = PG_Y; yi = q; kr = e2 - r0; kl = -1; xi = Lx - 1; }
        };
        // The reference sets y_ind (x_ind) to the last column (row) exactly when the close
        // candidate wins (VA:1461-1475); folding that into x_close/y_close is equivalent
        // because `best_score` there always equals max->score before the call.
        m_cand(l0, r0); x_close(l0); y_close(r0);
        for (int e2 = r0 + 1; e2 < r1; ++e2) { m_cand(l0, e2); y_close(e2); }
        for (int e1 = l0 + 1; e1 < l1; ++e1) {
            m_cand(e1, r0); x_close(e1);
            for (int e2 = r0 + 1; e2 < r1; ++e2) { m_cand(e1, e2); y_close(e2); }
        }
    }
    J.endscore[0] = best;
    gint_w ec = J.endcell;
    ec[1] = mat; ec[2] = xi; ec[3] = yi; ec[4] = kl; ec[5] = kr;
    if (!(best > NI)) { ec[0] = 1; ec[6] = 0; return; }

    // ---- traceback: i,j jump straight to the predecessor cell; the host re-inserts the
    // skipped sites (insert_preexisting_gap, viterbi_alignment.h:146-193) ----
    int vit = mat, i = xi, j = yi, n = 0, status = 0;
    const int cap = Lx + Ly;
    gint_w tr = J.trace;
    while (!(i < 1 && j < 1)) {
        if (n >= cap || vit > 2 || vit < 0) { status = 2; break; }
        const long long ix = hbm_index(J, -10, none, none, i, j);
        if (ix < 0) { status = 2; break; }
        const unsigned b = J.bp[3 * ix + vit];
        tr[3 * n] = i; tr[3 * n + 1] = j; tr[3 * n + 2] = (int)((unsigned)vit | (b & ~3u));
        ++n;
        const unsigned from = b & 3u;
        const int k1 = (int)((b >> 2) & 32767u), k2 = (int)(b >> 17);
        if (from == PG_BP_NONE) { status = 2; break; }
        if (vit == PG_M) { i = J.srcL[J.offL[i] + k1]; j = J.srcR[J.offR[j] + k2]; }
        else if (vit == PG_X) { i = J.srcL[J.offL[i] + k1]; }
        else { j = J.srcR[J.offR[j] + k2]; }
        vit = (int)from;
    }
    ec[0] = status; ec[6] = n;
}
