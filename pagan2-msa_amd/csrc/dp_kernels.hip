// dp_kernels.hip -- gfx950 kernels for the pairwise graph-vs-graph Viterbi DP.
//
// The recurrence (reference: Viterbi_alignment::compute_fwd_scores,
// src/main/viterbi_alignment.cpp:856-971, with iterate_bwd_edges_for_gap VA:1328-1349,
// iterate_bwd_edges_for_match VA:1353-1436 and score_* VA:2029-2219) is a scalar fp64
// max-plus over the incoming graph edges of the two sites -- no MFMA shape in it.  Every
// predecessor of cell (i,j) lies on an earlier anti-diagonal, so the fill sweeps d = i+j as
// a wavefront: one workgroup per alignment, one lane per in-band cell of the diagonal.
//
// Bit-exactness rules kept here (they decide the traceback through fp64 ties):
//   - every float parameter is promoted separately and added left to right, exactly as the
//     C++ expressions in the reference evaluate: (s + f1) + f2;
//   - `2*log_non_gap` and `0 + log_non_gap` are FLOAT operations (VA:1364-1367);
//   - a candidate replaces the incumbent only if strictly greater (first_is_bigger,
//     src/main/basic_alignment.h:449-462), candidates in the reference's order.
// Compiled with -ffp-contract=off; there are no multiplies to fuse on the fp64 path anyway.
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

// Diagonal descriptor hoisted once per step for the two diagonals nearly every edge lands on.
struct Diag { int mn, mx; long long off; };

struct CellCtx {
    const PgDevJob *J;
    int d;
    Diag d1, d2;          // diagonals d-1 and d-2
    // Linear index of cell (p,q), or -1 when it lies outside the tunnel
    // (Tunnel_slice::at returns the shared -inf entry there, src/utils/tunnel_matrix.h:85-98).
    __device__ __forceinline__ long long index(int p, int q) const {
        int dd = p + q;
        int mn, mx; long long off;
        if (dd == d - 1) { mn = d1.mn; mx = d1.mx; off = d1.off; }
        else if (dd == d - 2) { mn = d2.mn; mx = d2.mx; off = d2.off; }
        else { mn = J->imin[dd]; mx = J->imax[dd]; off = J->doff[dd]; }
        return (p >= mn && p <= mx) ? off + (p - mn) : -1;
    }
};

// One DP cell: the three states of (i,j).  Writes scores and back-pointers at `at`.
__device__ __forceinline__ void fill_cell(const PgDevJob &J, const CellCtx &cx, int i, int j, long long at,
                                          bool no_terminal_edges, bool reduced_terminal) {
    const double NI = neg_inf();
    double bx = NI, by = NI, bm = NI;
    unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
    const double *sX = J.sc[PG_X], *sY = J.sc[PG_Y], *sM = J.sc[PG_M];

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
                const long long ix = cx.index(p, j);
                double xs = NI, ys = NI, ms = NI;
                if (ix >= 0) { xs = sX[ix]; ys = sY[ix]; ms = sM[ix]; }
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
                const long long ix = cx.index(i, q);
                double xs = NI, ys = NI, ms = NI;
                if (ix >= 0) { xs = sX[ix]; ys = sY[ix]; ms = sM[ix]; }
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
                    const long long ix = cx.index(p, q);
                    double xs = NI, ys = NI, ms = NI;
                    if (ix >= 0) { xs = sX[ix]; ys = sY[ix]; ms = sM[ix]; }
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
    J.sc[PG_X][at] = bx; J.sc[PG_Y][at] = by; J.sc[PG_M][at] = bm;
    J.bp[PG_X][at] = px; J.bp[PG_Y][at] = py; J.bp[PG_M][at] = pm;
}

} // namespace

// Anti-diagonal wavefront fill, scores exchanged through HBM/L2 (every cell stays
// addressable because a graph edge may reach arbitrarily far back).  One workgroup per
// alignment; each diagonal is one grid-stride pass followed by a workgroup barrier whose
// release/acquire makes the stores visible to the next diagonal's loads.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pg_fill_wavefront(const PgDevJob *__restrict__ jobs, unsigned flags) {
    const PgDevJob J = jobs[blockIdx.x];
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    CellCtx cx;
    cx.J = &J;
    cx.d1 = {0, -1, 0};
    cx.d2 = {0, -1, 0};
    for (int d = 0; d < J.nd; ++d) {
        const int lo = J.imin[d], hi = J.imax[d];
        const long long base = J.doff[d];
        cx.d = d;
        for (int i = lo + (int)threadIdx.x; i <= hi; i += BLOCK)
            fill_cell(J, cx, i, d - i, base + (i - lo), no_terminal_edges, reduced_terminal);
        cx.d2 = cx.d1;
        cx.d1 = {lo, hi, base};
        __syncthreads();
    }
}

template __global__ void pg_fill_wavefront<64>(const PgDevJob *, unsigned);
template __global__ void pg_fill_wavefront<256>(const PgDevJob *, unsigned);
template __global__ void pg_fill_wavefront<1024>(const PgDevJob *, unsigned);

// End corner (iterate_bwd_edges_for_end_corner, VA:1440-1552, score_gap_close VA:2221-2255)
// and the pointer chase of backtrack_new_path (VA:1038-1189).  One lane per alignment: the
// chase is a serial dependency chain.  It emits the visited cells as (i, j, w) with
// w = the cell's own matrix in bits 0-1 and its two edge slots in bits 2-31 (the cell's
// `from` label is the next entry's matrix); skip columns and used-edge marks are derived
// from that list on the host.
__global__ void pg_end_and_trace(const PgDevJob *__restrict__ jobs) {
    if (threadIdx.x != 0) return;
    const PgDevJob J = jobs[blockIdx.x];
    const double NI = neg_inf();
    CellCtx cx; cx.J = &J; cx.d = -10; cx.d1 = {0, -1, 0}; cx.d2 = {0, -1, 0};
    const int Lx = J.Lx, Ly = J.Ly;
    const int l0 = J.offL[Lx], l1 = J.offL[Lx + 1];
    const int r0 = J.offR[Ly], r1 = J.offR[Ly + 1];

    double best = NI;          // max->score
    int mat = -1, xi = -1, yi = -1, kl = -1, kr = -1;   // max->matrix, x_ind, y_ind, edge slots (-1 = none)
    if (l1 > l0 && r1 > r0) {
        const double ng = (double)J.ng;
        auto m_cand = [&](int e1, int e2) {
            const int p = J.srcL[e1], q = J.srcR[e2];
            const long long ix = cx.index(p, q);
            const double ms = ix >= 0 ? J.sc[PG_M][ix] : NI;
            const double c = ((ms + ng) + (double)J.lwL[e1]) + (double)J.lwR[e2];
            if (c > best) { best = c; mat = PG_M; xi = p; yi = q; kl = e1 - l0; kr = e2 - r0; }
        };
        auto x_close = [&](int e1) {
            const int p = J.srcL[e1];
            const long long ix = cx.index(p, Ly - 1);
            const double c = (ix >= 0 ? J.sc[PG_X][ix] : NI) + 0.0;
            if (c > best) { best = c; mat = PG_X; xi = p; kl = e1 - l0; kr = -1; yi = Ly - 1; }
        };
        auto y_close = [&](int e2) {
            const int q = J.srcR[e2];
            const long long ix = cx.index(Lx - 1, q);
            const double c = (ix >= 0 ? J.sc[PG_Y][ix] : NI) + 0.0;
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
    int *ec = J.endcell;
    ec[1] = mat; ec[2] = xi; ec[3] = yi; ec[4] = kl; ec[5] = kr;
    if (!(best > NI)) { ec[0] = 1; ec[6] = 0; return; }

    // ---- traceback: i,j jump straight to the predecessor cell; the host re-inserts the
    // skipped sites (insert_preexisting_gap, viterbi_alignment.h:146-193) ----
    int vit = mat, i = xi, j = yi, n = 0, status = 0;
    const int cap = Lx + Ly;
    int *tr = J.trace;
    while (!(i < 1 && j < 1)) {
        if (n >= cap || vit > 2 || vit < 0) { status = 2; break; }
        const long long ix = cx.index(i, j);
        if (ix < 0) { status = 2; break; }
        const unsigned b = J.bp[vit][ix];
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
