// dp_kcommon.h -- device helpers shared by the fill kernels (dp_kernels.hip, dp_pipe.hip):
// the flattened job view, the bit-exact cell recurrence with operands in HBM, and the
// load/barrier primitives that keep HBM stores off the wavefront's critical path.
#pragma once
#include <hip/hip_runtime.h>
#include "dp_device.h"

#define PG_X 0
#define PG_Y 1
#define PG_M 2

namespace {

__device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }

__device__ __forceinline__ unsigned pack_bp(unsigned from, int k1, int k2, bool adjl, bool adjr) {
    return from | (adjl ? PG_BP_ADJL : 0u) | (adjr ? PG_BP_ADJR : 0u) | ((unsigned)k1 << 4) | ((unsigned)k2 << 18);
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
typedef int pg_i4 __attribute__((ext_vector_type(4)));
// read-only for the whole kernel and indexed uniformly: constant address space, so hipcc uses
// s_load (scalar cache, lgkmcnt) instead of a vector load that would queue behind the stores
typedef __attribute__((address_space(4))) const pg_i4 *cdesc_p;

// Job descriptor flattened to scalars (SGPRs): no arrays, never address-taken.
struct View {
    int Lx, Ly, nd, S;
    float go, ge, gE, ng;
    gint_p stL, offL, srcL; gfloat_p lwL;
    gint_p stR, offR, srcR; gfloat_p lwR;
    gfloat_p table;
    gint_p imin, imax; gll_p doff; cdesc_p dsc;
    gdouble_w sc;            // [cells][3]  X, Y, M
    gu32_w bp;               // [cells][3]
    gint_w trace, endcell; gdouble_w endscore;
    int n_bound; gint_p tb; gint_w ttab, segs;
};

__device__ __forceinline__ View load_view(const PgDevJob *__restrict__ j) {
    View v;
    v.Lx = j->Lx; v.Ly = j->Ly; v.nd = j->nd; v.S = j->S;
    v.go = j->go; v.ge = j->ge; v.gE = j->gE; v.ng = j->ng;
    v.stL = (gint_p)j->stL; v.offL = (gint_p)j->offL; v.srcL = (gint_p)j->srcL; v.lwL = (gfloat_p)j->lwL;
    v.stR = (gint_p)j->stR; v.offR = (gint_p)j->offR; v.srcR = (gint_p)j->srcR; v.lwR = (gfloat_p)j->lwR;
    v.table = (gfloat_p)j->table; v.imin = (gint_p)j->imin; v.imax = (gint_p)j->imax; v.doff = (gll_p)j->doff; v.dsc = (cdesc_p)j->dsc;
    v.sc = (gdouble_w)j->sc; v.bp = (gu32_w)j->bp;
    v.trace = (gint_w)j->trace; v.endcell = (gint_w)j->endcell; v.endscore = (gdouble_w)j->endscore;
    v.n_bound = j->n_bound; v.tb = (gint_p)j->tb; v.ttab = (gint_w)j->ttab; v.segs = (gint_w)j->segs;
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

// One DP cell, the three states of (i,j) and their back-pointers, in the reference's candidate order.
// Where the operands live is the caller's business:
//   fetch(p, q, xs, ys, ms)   scores of cell (p,q), -inf outside the band
//   edge_l(k, p, lw)          k-th bwd edge of left site i: start site and log weight (as double)
//   edge_r(k, q, rw)          the same for right site j
// nl / nr = number of bwd edges of the two sites, sm = the model's log score of the two states
// (used only when i > 0 && j > 0 and both sites have edges).
// (cell_any_t: the match terms tM = D(2*ng) + D(sm), tX = D(0+ng) + D(sm) come ready -- dp_pipe.hip keeps them in LDS)
template <class Fetch, class EdgeL, class EdgeR>
__device__ __forceinline__ void cell_any_t(const View &J, int i, int j, int nl, int nr, double tM, double tX, bool no_terminal_edges,
                                           bool reduced_terminal, Fetch fetch, EdgeL edge_l, EdgeR edge_r, double &bx,
                                           double &by, double &bm, unsigned &px, unsigned &py, unsigned &pm) {
    const double NI = neg_inf();
    bx = NI; by = NI; bm = NI;
    px = PG_BP_NONE; py = PG_BP_NONE; pm = PG_BP_NONE;
    if (i == 0 && j == 0) {
        bm = 0.0;                                   // initialise_array_corner, VA:725-736
        return;
    }
    const double go = (double)J.go, ng = (double)J.ng;
    // ---- X: gap in the right sequence, consumes left site i (VA:898-915) ----
    if (i > 0) {
        const bool end_gap = (j == 0 || j == J.Ly - 1) && !no_terminal_edges;   // VA:864-868
        const double ext = (double)(end_gap ? J.gE : J.ge);
        for (int k = 0; k < nl; ++k) {
            int p; double lw;
            edge_l(k, p, lw);
            double xs, ys, ms;
            fetch(p, j, xs, ys, ms);
            const double open = (reduced_terminal && p == 0) ? 0.0 : go;        // BA.h:490-513
            double c = xs + ext;                                                 // score_gap_ext
            if (c > bx) { bx = c; px = pack_bp(PG_X, k, 0, p == i - 1, false); }
            c = (ys + 0.0) + go;                                                 // score_gap_double
            if (c > bx) { bx = c; px = pack_bp(PG_Y, k, 0, p == i - 1, false); }
            c = (ms + ng) + open;                                                // score_gap_open
            if (c > bx) { bx = c; px = pack_bp(PG_M, k, 0, p == i - 1, false); }
        }
    }
    // ---- Y: gap in the left sequence, consumes right site j (VA:927-944) ----
    if (j > 0) {
        const bool end_gap = (i == 0 || i == J.Lx - 1) && !no_terminal_edges;   // VA:875-879
        const double ext = (double)(end_gap ? J.gE : J.ge);
        for (int k = 0; k < nr; ++k) {
            int q; double rw;
            edge_r(k, q, rw);
            double xs, ys, ms;
            fetch(i, q, xs, ys, ms);
            const double open = (reduced_terminal && q == 0) ? 0.0 : go;
            double c = ys + ext;
            if (c > by) { by = c; py = pack_bp(PG_Y, 0, k, false, q == j - 1); }
            c = (xs + 0.0) + go;
            if (c > by) { by = c; py = pack_bp(PG_X, 0, k, false, q == j - 1); }
            c = (ms + ng) + open;
            if (c > by) { by = c; py = pack_bp(PG_M, 0, k, false, q == j - 1); }
        }
    }
    // ---- M: both sites consumed (VA:956-963, 1353-1436) ----
    if (i > 0 && j > 0 && nl > 0 && nr > 0) {
        for (int k1 = 0; k1 < nl; ++k1) {
            int p; double lw;
            edge_l(k1, p, lw);
            for (int k2 = 0; k2 < nr; ++k2) {
                int q; double rw;
                edge_r(k2, q, rw);
                double xs, ys, ms;
                fetch(p, q, xs, ys, ms);
                double c = ((ms + tM) + lw) + rw;                                // score_m_match
                if (c > bm) { bm = c; pm = pack_bp(PG_M, k1, k2, p == i - 1, q == j - 1); }
                c = ((xs + tX) + lw) + rw;                                       // score_x_match
                if (c > bm) { bm = c; pm = pack_bp(PG_X, k1, k2, p == i - 1, q == j - 1); }
                c = ((ys + tX) + lw) + rw;                                       // score_y_match
                if (c > bm) { bm = c; pm = pack_bp(PG_Y, k1, k2, p == i - 1, q == j - 1); }
            }
        }
    }
}

template <class Fetch, class EdgeL, class EdgeR>
__device__ __forceinline__ void cell_any(const View &J, int i, int j, int nl, int nr, float sm, bool no_terminal_edges,
                                         bool reduced_terminal, Fetch fetch, EdgeL edge_l, EdgeR edge_r, double &bx,
                                         double &by, double &bm, unsigned &px, unsigned &py, unsigned &pm) {
    const double tM = (double)(2 * J.ng) + (double)sm;                          // VA:1364
    const double tX = (double)(0.0f + J.ng) + (double)sm;                       // VA:1366-1367
    cell_any_t(J, i, j, nl, nr, tM, tX, no_terminal_edges, reduced_terminal, fetch, edge_l, edge_r, bx, by, bm, px, py, pm);
}

// The same with every operand in HBM/L2 (graph arrays, model table, scores), written at `at`.
__device__ __forceinline__ void fill_cell_hbm(const View &J, int d, const Diag &d1, const Diag &d2, int i, int j,
                                              long long at, bool no_terminal_edges, bool reduced_terminal) {
    int l0 = 0, l1 = 0, r0 = 0, r1 = 0;
    if (i > 0) { l0 = J.offL[i]; l1 = J.offL[i + 1]; }
    if (j > 0) { r0 = J.offR[j]; r1 = J.offR[j + 1]; }
    float sm = 0.0f;
    if (i > 0 && j > 0 && l1 > l0 && r1 > r0) sm = J.table[J.stL[i] + J.stR[j] * J.S];       // VA:1363
    double bx, by, bm;
    unsigned px, py, pm;
    cell_any(J, i, j, l1 - l0, r1 - r0, sm, no_terminal_edges, reduced_terminal,
             [&](int p, int q, double &xs, double &ys, double &ms) {
                 const long long ix = hbm_index(J, d, d1, d2, p, q);
                 xs = ys = ms = neg_inf();
                 if (ix >= 0) { xs = J.sc[3 * ix + PG_X]; ys = J.sc[3 * ix + PG_Y]; ms = J.sc[3 * ix + PG_M]; }
             },
             [&](int k, int &p, double &lw) { p = J.srcL[l0 + k]; lw = (double)J.lwL[l0 + k]; },
             [&](int k, int &q, double &rw) { q = J.srcR[r0 + k]; rw = (double)J.lwR[r0 + k]; },
             bx, by, bm, px, py, pm);
    store_cell(J.sc, J.bp, at, bx, by, bm, px, py, pm);
}

// A job's status word (PgDevJob::fill_status): the FIRST report stays -- the strips of a job share their parent's word, and a
// later report (another strip's abort, the score check over a matrix an aborted fill left unfinished) must not hide it.  A row
// strip that found the strip above on another XCD (dp_pipe.hip, strip_feeder: tag 11) additionally sets PG_FILL_OTHER_XCD, which
// no other report clears: the host's re-run of the strips alone hinges on it.
__device__ __forceinline__ void report_fill_status(const PgDevJob *job, int status) {
    PG_GLOBAL int *w = (PG_GLOBAL int *)job->fill_status;
    status &= 0x1fffffff;                                           // (an abort tag is kind | wave << 4 | diagonal << 8)
    if ((status & 0xf) == 11 && status != PG_FILL_SCORE_MISMATCH) __hip_atomic_fetch_or(w, PG_FILL_OTHER_XCD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int expected = 0;
    __hip_atomic_compare_exchange_strong(w, &expected, status, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (a word that holds the sticky bit only takes the report's other bits as well)
    if (expected == PG_FILL_OTHER_XCD) __hip_atomic_compare_exchange_strong(w, &expected, status | PG_FILL_OTHER_XCD, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Loads the COMPUTE waves need only on rare paths (an edge reaching past the ring).  Issued as
// inline asm with their own wait so that the compiler's waitcnt insertion never places a
// vmcnt(0) -- which would also wait for every store in flight -- on the common path.
__device__ __forceinline__ float far_f32(PG_GLOBAL const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ pg_i4 far_desc(PG_GLOBAL const pg_i4 *p) {
    pg_i4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// the three scores of one cell (X, Y, M: 24 contiguous bytes), one wait for both loads
__device__ __forceinline__ void far_cell(PG_GLOBAL const double *p, double &xs, double &ys, double &ms) {
    typedef double pg_d2 __attribute__((ext_vector_type(2)));
    pg_d2 xy; double m;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(xy), "=&v"(m) : "v"(p) : "memory");
    xs = xy.x; ys = xy.y; ms = m;
}

// Workgroup barrier that orders LDS traffic only.  (__syncthreads() also waits for vmcnt(0),
// i.e. for this wave's HBM stores -- exactly what the steady state must not do.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// First-wins maximum of three candidates (first_is_bigger, basic_alignment.h:449-462, applied in
// candidate order to an incumbent of -inf): the value is the plain maximum, the winner is the
// first candidate equal to it, and nobody wins when all three are -inf.  Branch-free: two
// v_max_f64 for the value, three compares + three selects for the winner (a lone wave pays
// 5-8 cycles per instruction and ~20 per taken branch).  v_max returns +0 for (+0, -0) whatever
// the order, where a compare-and-select keeps the incumbent's sign; a score can only be -0.0
// when an input parameter is -0.0f, and the host keeps such jobs off the kernels using this.
__device__ __forceinline__ double first_max3(double c1, double c2, double c3, unsigned f1, unsigned f2, unsigned f3,
                                             unsigned &bp) {
    const double m23 = __builtin_fmax(c2, c3);
    const double m = __builtin_fmax(c1, m23);
    const bool w3 = c3 > c2;             // a later candidate wins only if strictly greater
    const bool w23 = m23 > c1;
    unsigned b = w3 ? f3 : f2;
    b = w23 ? b : f1;
    bp = (m > neg_inf()) ? b : PG_BP_NONE;
    return m;
}


} // namespace
