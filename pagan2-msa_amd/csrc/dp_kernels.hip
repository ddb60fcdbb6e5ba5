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

#include "dp_kcommon.h"


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
// Back-pointer pass.  A back-pointer is a function of scores that are final once the fill has passed the cell:
// the first candidate, in the reference's order, that equals the state's maximum (first_is_bigger is strict,
// basic_alignment.h:449-462; candidates VA:1328-1349, 1396-1433, 2029-2219).  Nothing on the fill's dependency
// chain needs it, so the banded fill's hand-scheduled loop (dp_pipe.hip) computes and stores scores only and this
// kernel -- one thread per cell, every cell independent -- re-evaluates each cell's candidates from the stored
// scores of its predecessors, with the same expressions on the same doubles, and writes the 12 bytes.  HBM-bound:
// 24 B read (neighbours come out of L2) + 12 B written per cell.
// grid (ceil(max nd / diags_per_block), n_jobs, ceil(widest diagonal / PG_BP_CELLS)), block 256: wave w of block b takes the
// diagonals b.x * diags_per_block + w, + 4, ... (diags_per_block: PG_BP_DIAGS, fewer for small batches so that the grid fills the chip); its lanes the cells b.z * PG_BP_CELLS .. of each (a full matrix has diagonals of
// thousands of cells: one wave per diagonal left most of the chip idle).
// flags bit 8 (diagnostic): nothing is written; the recomputed scores AND back-pointers are compared with what the
// fill kernel stored, a difference sets fill_status to 0x7d (tests run this over whole matrices filled by the
// kernels that still write their own back-pointers).
__global__ __launch_bounds__(256) void pg_backptr(const PgDevJob *__restrict__ jobs, const int *__restrict__ which,
                                                  unsigned flags, int diags_per_block) {
    const PgDevJob *__restrict__ job = jobs + which[blockIdx.y];
    const int first = blockIdx.x * diags_per_block;
    if (first >= job->nd) return;
    const View J = load_view(job);
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const bool verify = flags & 0x100u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int end = first + diags_per_block < J.nd ? first + diags_per_block : J.nd;
    PG_GLOBAL const unsigned char *done = verify ? nullptr : (PG_GLOBAL const unsigned char *)job->bp_done;
    for (int d = first + wave; d < end; d += 4) {
        if (done && done[d / PG_FOLLOW_CHUNK]) continue;           // written behind the fill (dp_pipe.hip, pipe_follower)
        const pg_i4 cur = J.dsc[d];
        const int lo = cur.x, hi = cur.y;
        if (hi < lo) continue;
        const long long base = ((long long)cur.w << 32) | (unsigned)cur.z;
        Diag d1 = {0, -1, 0}, d2 = {0, -1, 0};
        if (d > 0) { const pg_i4 p = J.dsc[d - 1]; d1 = {p.x, p.y, ((long long)p.w << 32) | (unsigned)p.z}; }
        if (d > 1) { const pg_i4 p = J.dsc[d - 2]; d2 = {p.x, p.y, ((long long)p.w << 32) | (unsigned)p.z}; }
        const int c0 = lo + (int)blockIdx.z * PG_BP_CELLS, c1 = c0 + PG_BP_CELLS - 1 < hi ? c0 + PG_BP_CELLS - 1 : hi;
        for (int i = c0 + lane; i <= c1; i += 64) {
            const int j = d - i;
            const long long at = base + (i - lo);
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
            if (verify) {
                const bool same = __double_as_longlong(bx) == __double_as_longlong(J.sc[3 * at + PG_X]) &&
                                  __double_as_longlong(by) == __double_as_longlong(J.sc[3 * at + PG_Y]) &&
                                  __double_as_longlong(bm) == __double_as_longlong(J.sc[3 * at + PG_M]) &&
                                  px == J.bp[3 * at + PG_X] && py == J.bp[3 * at + PG_Y] && pm == J.bp[3 * at + PG_M];
                if (!same) *(PG_GLOBAL int *)job->fill_status = 0x7d;
            } else {
                typedef unsigned u3 __attribute__((ext_vector_type(3)));
                u3 b; b.x = px; b.y = py; b.z = pm;
                *(PG_GLOBAL u3 *)(J.bp + 3 * at) = b;
                // flags bit 9 (PG_FLAG_SCORE_CHECK): the cell's stored scores are what its predecessors' stored scores give, bit
                // for bit -- the recurrence holds at EVERY cell, so the whole matrix is the oracle's by induction from cell (0,0).
                // The fill kernels hand scores from wave to wave and from workgroup to workgroup (row strips) on landing rules;
                // a score that arrived through a stale read fails here, whether or not the traceback visits the cell.
                if (flags & PG_FLAG_SCORE_CHECK) {
                    const bool same = __double_as_longlong(bx) == __double_as_longlong(J.sc[3 * at + PG_X]) &&
                                      __double_as_longlong(by) == __double_as_longlong(J.sc[3 * at + PG_Y]) &&
                                      __double_as_longlong(bm) == __double_as_longlong(J.sc[3 * at + PG_M]);
                    if (!same) report_fill_status(job, PG_FILL_SCORE_MISMATCH);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Banded wavefront with the active band staged in LDS ("ring" kernel).
//
// One workgroup per alignment: 2 NW compute waves + one loader wave, one s_barrier per
// anti-diagonal.  Slot u = i % NT (NT = 64 NW) of the current diagonal is worked on by TWO
// threads: thread u (waves 0..NW-1, the "gap" role) computes the X and Y states of the cell,
// thread NT+u (waves NW..2NW-1, the "match" role) its M state.  The two roles read the same
// predecessors and write disjoint outputs, so splitting the states needs no combine step and
// takes ~40 % of the instructions off each wave -- the fill is instruction-issue bound (one
// wave issues a VALU instruction every 4-5 cycles), not bandwidth bound.  Up to NT cells are
// computed at once.  Per step (anti-diagonal d):
//   - the scores of the last RK diagonals live in an LDS ring  sc[d % RK][i % NT][X,Y,M];
//     every predecessor within RK diagonals is an LDS read guarded by that diagonal's
//     [imin,imax] interval;
//   - site data, bwd edges and the per-diagonal band index sit in LDS rings that the loader
//     wave refills with coalesced loads (64 sites / diagonals at a time) one step ahead;
//   - scores and back-pointers stream to HBM with coalesced stores nobody waits for: the
//     compute waves issue NO vector-memory load in the steady state (vmcnt returns in order,
//     a load would stall on every store still in flight -- measured 5.6 us per diagonal);
//   - every thread writes its slot of the diagonal's ring row -- scores inside the band, -inf
//     outside -- so reading a predecessor needs no band test, only "is that diagonal resident";
//   - a wave whose cells are all "simple" (both sites have one bwd edge, from the
//     predecessor site, weight 1) runs straight-line code: three 24-byte LDS reads, nine
//     candidates; otherwise each lane walks its (left edge, right edge) pairs in the
//     reference's order;
//   - a graph edge reaching >= RK diagonals back reads HBM (L1-bypassing loads); every wave
//     drains its stores once per RK/2 diagonals, so a cell that old has landed;
//   - a diagonal wider than NTW (a box between distant anchors) is computed from HBM operands
//     between two full drains and marked "not in the ring".
#define NW 4
#define NT (64 * NW)
// widest diagonal kept in the ring: leaves RK rows of slack on either side of the NT-slot window so
// that a predecessor row just outside an older diagonal's band can never alias a row inside it
#define NTW (NT - 16)
#define RK 16
#define RW 512
#define EC 2048
#define DR 256
// bit set in the cached state word of a site whose only bwd edge comes from its predecessor
// site with weight 1 (log-weight +0.0): adding that weight is an exact no-op
#define PG_SIMPLE 0x10000
#define PG_SPAN_SHIFT 17

struct RingSmem {
    double sc[RK][NT][3];                   // X, Y, M
    int stL[RW], ebL[RW], eeL[RW];
    int stR[RW], ebR[RW], eeR[RW];
    int esL[EC]; float ewL[EC];
    int esR[EC]; float ewR[EC];
    float table[256];
};

unsigned pg_ring_lds_bytes() { return (unsigned)sizeof(RingSmem); }

extern __shared__ __attribute__((aligned(16))) char pg_ring_lds[];
#define SM (*reinterpret_cast<RingSmem *>(pg_ring_lds))

namespace {


// Scores of cell (p,q) on an earlier diagonal; -inf outside the tunnel.  Every slot of a resident
// ring row is valid -- threads outside the band write -inf into theirs -- so a resident cell is
// three LDS reads with no band test.  `resident` has bit (dd % RK) set when diagonal dd (one of
// the last RK-1) went through the ring; anything else is read back from HBM.
__device__ __forceinline__ void ring_load(const View &J, int d, unsigned resident, int p, int q,
                                          double &xs, double &ys, double &ms) {
    const int dd = p + q, slot = dd & (RK - 1);
    if (d - dd < RK && ((resident >> slot) & 1u)) {
        xs = SM.sc[slot][p & (NT - 1)][PG_X];
        ys = SM.sc[slot][p & (NT - 1)][PG_Y];
        ms = SM.sc[slot][p & (NT - 1)][PG_M];
    } else {
        // left the ring (edge reaching >= RK diagonals back) or never entered it (wide diagonal):
        // two dependent HBM round trips -- the diagonal's packed descriptor, then the cell's 24 bytes
        const double NI = neg_inf();
        xs = NI; ys = NI; ms = NI;
        const pg_i4 ds = far_desc((PG_GLOBAL const pg_i4 *)J.dsc + dd);
        if (p >= ds.x && p <= ds.y) {
            const long long ix = (((long long)ds.w << 32) | (unsigned)ds.z) + (p - ds.x);
            far_cell(J.sc + 3 * ix, xs, ys, ms);
        }
    }
}

// Resident cell read with no residency test of its own (the caller established it): scores of
// (p,q), or -inf when `has` is false (a site's missing second edge).
__device__ __forceinline__ void ring_cell(int p, int q, bool has, double &xs, double &ys, double &ms) {
    const int slot = (p + q) & (RK - 1);
    const double rx = SM.sc[slot][p & (NT - 1)][PG_X];
    const double ry = SM.sc[slot][p & (NT - 1)][PG_Y];
    const double rm = SM.sc[slot][p & (NT - 1)][PG_M];
    const double NI = neg_inf();
    xs = has ? rx : NI; ys = has ? ry : NI; ms = has ? rm : NI;
}

// True when every active lane of the wave has at least one bwd edge on both sites and every
// predecessor diagonal its edges can reach is resident in the ring (so its item loop can read
// cells with ring_cell, without per-read residency branches).  The reach is bounded by the two
// sites' farthest edges: a pair (p,q) lies (i-p)+(j-q) <= spanL+spanR diagonals back.
__device__ __forceinline__ bool all_resident(int d, unsigned resident, bool active, int wi, int wj) {
    bool ok = true;
    if (active) {
        const int sl = (wi >> PG_SPAN_SHIFT) & 255, sr = (wj >> PG_SPAN_SHIFT) & 255;
        const int age = sl + sr;
        ok = sl > 0 && sr > 0 && age < RK;
        if (ok) {
            const unsigned run = ((1u << age) - 1u);                         // diagonals d-age .. d-1, modulo RK
            const int first = (d - age) & (RK - 1);
            const unsigned need = ((run << first) | (run >> (RK - first))) & ((1u << RK) - 1u);
            ok = (resident & need) == need;
        }
    }
    return __all(ok);
}

// ---- loader wave -------------------------------------------------------------------------
__device__ __forceinline__ void load_site_chunk(int first, int lane, int n, gint_p st, gint_p off, gint_p src, gfloat_p lw,
                                                int *cst, int *ceb, int *cee, int *ces, float *cew) {
    const int r = first + lane;
    if (r < n) {
        const int b = off[r], en = off[r + 1];
        int w = st[r] & 0xffff;
        if (r > 0 && en - b == 1 && src[b] == r - 1 && lw[b] == 0.0f) w |= PG_SIMPLE;
        // farthest bwd edge of the site, in sites (0 = no edge at all, saturates at 255): lets the compute
        // waves bound how far back a cell reaches without walking its edge list
        int span = 0;
        for (int e = b; e < en && e < b + 16; ++e) { const int sp = r - src[e]; span = sp > span ? sp : span; }
        if (en - b > 16 || span > 255) span = 255;
        w |= span << PG_SPAN_SHIFT;
        cst[r & (RW - 1)] = w; ceb[r & (RW - 1)] = b; cee[r & (RW - 1)] = en;
    }
    const int rend = first + 64 < n ? first + 64 : n;
    const int e0 = off[first], e1 = off[rend];
    for (int e = e0 + lane; e < e1; e += 64) { ces[e & (EC - 1)] = src[e]; cew[e & (EC - 1)] = lw[e]; }
}

// everything diagonal `dn` needs from the site/edge windows
__device__ __forceinline__ void loader_prepare(const View &J, int dn, int lane, int &rows, int &cols) {
    const pg_i4 ds = J.dsc[dn];
    const int lo = ds.x, hi = ds.y;
    if (hi - lo + 1 > NTW) return;                                // wide diagonals read HBM
    bool any = false;
    while (hi >= rows && rows < J.Lx) { load_site_chunk(rows, lane, J.Lx, J.stL, J.offL, J.srcL, J.lwL, SM.stL, SM.ebL, SM.eeL, SM.esL, SM.ewL); rows += 64; any = true; }
    while (dn - lo >= cols && cols < J.Ly) { load_site_chunk(cols, lane, J.Ly, J.stR, J.offR, J.srcR, J.lwR, SM.stR, SM.ebR, SM.eeR, SM.esR, SM.ewR); cols += 64; any = true; }
    if (any) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

} // namespace

template <bool TAB_LDS>
__global__ __launch_bounds__(2 * NT + 64) void pg_fill_ring(const PgDevJob *__restrict__ jobs, const int *__restrict__ which,
                                                            unsigned flags) {
    const View J = load_view(jobs + which[blockIdx.x]);
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (TAB_LDS) for (int k = tid; k < J.S * J.S; k += 2 * NT + 64) SM.table[k] = J.table[k];

    if (tid >= 2 * NT) {
        // ================= loader wave: stays one diagonal ahead of the compute waves =================
        int rows = 0, cols = 0;
        loader_prepare(J, 0, lane, rows, cols);
        lds_barrier();                                            // (B0) compute may start diagonal 0
        for (int d = 0; d < J.nd; ++d) {
            const pg_i4 ds = J.dsc[d];
            if (ds.y - ds.x + 1 > NTW) lds_barrier();             // (Bw) mirrors the compute waves' drain barrier
            if (d + 1 < J.nd) loader_prepare(J, d + 1, lane, rows, cols);
            lds_barrier();                                        // (Bd) end of diagonal d
        }
        return;
    }

    // ================= compute waves =================
    const bool match_role = tid >= NT;                            // wave-uniform: waves NW..2NW-1
    const int u = tid & (NT - 1);                                 // ring slot this thread works on
    const double NI = neg_inf();
    const double go = (double)J.go, ng = (double)J.ng, ge = (double)J.ge;
    const double tng2 = (double)(2 * J.ng), tng1 = (double)(0.0f + J.ng);
    unsigned resident = 0;                                        // bit (dd % RK): diagonal dd is in the ring
    pg_i4 nxt = J.dsc[0];
    lds_barrier();                                                // (B0)
    for (int d = 0; d < J.nd; ++d) {
        const pg_i4 cur = nxt;
        nxt = J.dsc[d + 1 < J.nd ? d + 1 : d];                    // scalar prefetch for the next step
        const int lo = cur.x, hi = cur.y;
        const long long base = ((long long)cur.w << 32) | (unsigned)cur.z;
        const bool wide = hi - lo + 1 > NTW;
        const unsigned slot_bit = 1u << (d & (RK - 1));
        if (wide) {
            // rare: a box between anchors wider than the ring.  Every wave drains its stores, then all
            // cells are computed from HBM/L2 operands, then drained again so that later diagonals
            // (which find "not in the ring") read landed data.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();                                        // (Bw)
            const pg_i4 p1 = J.dsc[d > 0 ? d - 1 : 0], p2 = J.dsc[d > 1 ? d - 2 : 0];
            const Diag g1 = {p1.x, d > 0 ? p1.y : p1.x - 1, ((long long)p1.w << 32) | (unsigned)p1.z};
            const Diag g2 = {p2.x, d > 1 ? p2.y : p2.x - 1, ((long long)p2.w << 32) | (unsigned)p2.z};
            for (int i = lo + tid; i <= hi; i += 2 * NT)
                fill_cell_hbm(J, d, g1, g2, i, d - i, base + (i - lo), no_terminal_edges, reduced_terminal);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            resident &= ~slot_bit;
        } else {
            const int off = (u - lo) & (NT - 1);                  // position of this thread's row in the band
            const int i = lo + off;
            const int j = d - i;
            const bool active = i <= hi;
            int wi = 0, wj = 0;
            bool simple = true;
            if (active) {
                wi = SM.stL[i & (RW - 1)]; wj = SM.stR[j & (RW - 1)];
                simple = (wi & wj & PG_SIMPLE) != 0;                      // implies i >= 1 and j >= 1
            }
            double bx = NI, by = NI, bm = NI;
            unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
            bool all_by_gap = false;                              // general path: the gap-role thread does all three states
            // interior: no first/last row or column on this diagonal, so every gap extension is the
            // normal one and every gap open pays the full penalty (BA.h:490-513, VA:864-879)
            const bool interior = lo >= 2 && hi <= J.Lx - 2 && d - hi >= 2 && d - lo <= J.Ly - 2;
            const unsigned prev2 = (1u << ((d - 1) & (RK - 1))) | (1u << ((d - 2) & (RK - 1)));
            if ((resident & prev2) == prev2 && interior && __all(simple)) {
                // ---- every cell of this wave is simple: straight-line code, no band tests ----
                // `+ 0.0` (log_gap_close, unit edge weights) is omitted: exact, no score is ever -0.0
                if (active && !match_role) {
                    const double *A = SM.sc[(d - 1) & (RK - 1)][(i - 1) & (NT - 1)];   // (i-1, j)   on d-1
                    const double *B = SM.sc[(d - 1) & (RK - 1)][i & (NT - 1)];         // (i, j-1)   on d-1
                    const double xA = A[PG_X], yA = A[PG_Y], mA = A[PG_M];
                    const double xB = B[PG_X], yB = B[PG_Y], mB = B[PG_M];
                    bx = first_max3(xA + ge, yA + go, (mA + ng) + go, PG_X | PG_BP_ADJL, PG_Y | PG_BP_ADJL, PG_M | PG_BP_ADJL, px);
                    by = first_max3(yB + ge, xB + go, (mB + ng) + go, PG_Y | PG_BP_ADJR, PG_X | PG_BP_ADJR, PG_M | PG_BP_ADJR, py);
                }
                if (active && match_role) {
                    const double *Cc = SM.sc[(d - 2) & (RK - 1)][(i - 1) & (NT - 1)];  // (i-1, j-1) on d-2
                    const double xC = Cc[PG_X], yC = Cc[PG_Y], mC = Cc[PG_M];
                    const int ti = (wi & 0xffff) + (wj & 0xffff) * J.S;
                    const float smf = TAB_LDS ? SM.table[ti] : far_f32(J.table + ti);
                    const double tM = tng2 + (double)smf, tX = tng1 + (double)smf;
                    bm = first_max3(mC + tM, xC + tX, yC + tX, PG_M | PG_BP_ADJL | PG_BP_ADJR, PG_X | PG_BP_ADJL | PG_BP_ADJR,
                                    PG_Y | PG_BP_ADJL | PG_BP_ADJR, pm);
                }
            } else if (interior && all_resident(d, resident, active, wi, wj)) {
                // ---- multi-edge sites, every predecessor in the ring, branch-free cell reads.  Gap role:
                // X candidates in left-list order, Y candidates in right-list order.  Match role: the
                // (left edge, right edge) pairs row-major, the reference's pair order (VA:1396-1433).
                int l0 = 0, nL = 0, r0 = 0, nR = 1;
                if (active) {
                    l0 = SM.ebL[i & (RW - 1)]; nL = SM.eeL[i & (RW - 1)] - l0;
                    r0 = SM.ebR[j & (RW - 1)]; nR = SM.eeR[j & (RW - 1)] - r0;
                }
                if (!match_role) {
                    const int n_items = active ? (nL > nR ? nL : nR) : 0;
                    for (int t = 0; __any(t < n_items); ++t) {
                        double xs, ys, ms, c;
                        if (t < nL && active) {                                      // X candidates of left edge t
                            const int p = SM.esL[(l0 + t) & (EC - 1)];
                            ring_cell(p, j, true, xs, ys, ms);
                            const double open = (reduced_terminal && p == 0) ? 0.0 : go;
                            const unsigned w = pack_bp(0, t, 0, p == i - 1, false);
                            c = xs + ge;            if (c > bx) { bx = c; px = w | PG_X; }
                            c = (ys + 0.0) + go;    if (c > bx) { bx = c; px = w | PG_Y; }
                            c = (ms + ng) + open;   if (c > bx) { bx = c; px = w | PG_M; }
                        }
                        if (t < nR && active) {                                      // Y candidates of right edge t
                            const int q = SM.esR[(r0 + t) & (EC - 1)];
                            ring_cell(i, q, true, xs, ys, ms);
                            const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                            const unsigned w = pack_bp(0, 0, t, false, q == j - 1);
                            c = ys + ge;            if (c > by) { by = c; py = w | PG_Y; }
                            c = (xs + 0.0) + go;    if (c > by) { by = c; py = w | PG_X; }
                            c = (ms + ng) + open;   if (c > by) { by = c; py = w | PG_M; }
                        }
                    }
                } else {
                    int n_items = 0;
                    double tM = 0, tX = 0;
                    if (active) {
                        n_items = nL * nR;
                        const int ti = (wi & 0xffff) + (wj & 0xffff) * J.S;
                        const float smf = TAB_LDS ? SM.table[ti] : far_f32(J.table + ti);
                        tM = tng2 + (double)smf; tX = tng1 + (double)smf;
                    }
                    int k1 = 0, k2 = 0;
                    for (int t = 0; __any(t < n_items); ++t) {
                        if (t < n_items) {
                            const int p = SM.esL[(l0 + k1) & (EC - 1)], q = SM.esR[(r0 + k2) & (EC - 1)];
                            const double lw = (double)SM.ewL[(l0 + k1) & (EC - 1)], rw = (double)SM.ewR[(r0 + k2) & (EC - 1)];
                            double xs, ys, ms, c;
                            ring_cell(p, q, true, xs, ys, ms);
                            const unsigned w = pack_bp(0, k1, k2, p == i - 1, q == j - 1);
                            c = ((ms + tM) + lw) + rw;  if (c > bm) { bm = c; pm = w | PG_M; }
                            c = ((xs + tX) + lw) + rw;  if (c > bm) { bm = c; pm = w | PG_X; }
                            c = ((ys + tX) + lw) + rw;  if (c > bm) { bm = c; pm = w | PG_Y; }
                            if (++k2 == nR) { k2 = 0; ++k1; }
                        }
                    }
                }
            } else {
                // ---- general (first/last rows and columns, sites without predecessors, edges reaching past
                // the ring): the gap-role thread computes all three states, walking its (left edge, right
                // edge) pairs row-major, which visits X candidates in left-list order, Y candidates in
                // right-list order and M candidates in the reference's pair order (VA:1396-1433) ----
                all_by_gap = true;
                if (!match_role) {
                    int l0 = 0, nL = 0, r0 = 0, nR = 0, n_items = 0;
                    double tM = 0, tX = 0, extX = 0, extY = 0;
                    if (active) {
                        if (i > 0) { l0 = SM.ebL[i & (RW - 1)]; nL = SM.eeL[i & (RW - 1)] - l0; }
                        if (j > 0) { r0 = SM.ebR[j & (RW - 1)]; nR = SM.eeR[j & (RW - 1)] - r0; }
                        if (i == 0 && j == 0) bm = 0.0;                        // initialise_array_corner, VA:725-736
                        else n_items = (nL > 0 ? nL : 1) * (nR > 0 ? nR : 1);
                        if (nL > 0 && nR > 0) {
                            const int ti = (wi & 0xffff) + (wj & 0xffff) * J.S;
                            const float smf = TAB_LDS ? SM.table[ti] : far_f32(J.table + ti);
                            tM = tng2 + (double)smf; tX = tng1 + (double)smf;
                        }
                        extX = (double)(((j == 0 || j == J.Ly - 1) && !no_terminal_edges) ? J.gE : J.ge);
                        extY = (double)(((i == 0 || i == J.Lx - 1) && !no_terminal_edges) ? J.gE : J.ge);
                    }
                    const int nRp = nR > 0 ? nR : 1;
                    int k1 = 0, k2 = 0;
                    for (int t = 0; __any(t < n_items); ++t) {
                        if (t < n_items) {
                            int p = 0, q = 0;
                            double lw = 0, rw = 0, xs, ys, ms, c;
                            if (nL > 0) { p = SM.esL[(l0 + k1) & (EC - 1)]; lw = (double)SM.ewL[(l0 + k1) & (EC - 1)]; }
                            if (nR > 0) { q = SM.esR[(r0 + k2) & (EC - 1)]; rw = (double)SM.ewR[(r0 + k2) & (EC - 1)]; }
                            if (nL > 0 && k2 == 0) {                                     // X candidates of left edge k1
                                ring_load(J, d, resident, p, j, xs, ys, ms);
                                const double open = (reduced_terminal && p == 0) ? 0.0 : go;
                                c = xs + extX;          if (c > bx) { bx = c; px = pack_bp(PG_X, k1, 0, p == i - 1, false); }
                                c = (ys + 0.0) + go;    if (c > bx) { bx = c; px = pack_bp(PG_Y, k1, 0, p == i - 1, false); }
                                c = (ms + ng) + open;   if (c > bx) { bx = c; px = pack_bp(PG_M, k1, 0, p == i - 1, false); }
                            }
                            if (nR > 0 && k1 == 0) {                                     // Y candidates of right edge k2
                                ring_load(J, d, resident, i, q, xs, ys, ms);
                                const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                                c = ys + extY;          if (c > by) { by = c; py = pack_bp(PG_Y, 0, k2, false, q == j - 1); }
                                c = (xs + 0.0) + go;    if (c > by) { by = c; py = pack_bp(PG_X, 0, k2, false, q == j - 1); }
                                c = (ms + ng) + open;   if (c > by) { by = c; py = pack_bp(PG_M, 0, k2, false, q == j - 1); }
                            }
                            if (nL > 0 && nR > 0) {                                      // M candidates of the pair
                                ring_load(J, d, resident, p, q, xs, ys, ms);
                                c = ((ms + tM) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_M, k1, k2, p == i - 1, q == j - 1); }
                                c = ((xs + tX) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_X, k1, k2, p == i - 1, q == j - 1); }
                                c = ((ys + tX) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_Y, k1, k2, p == i - 1, q == j - 1); }
                            }
                            if (++k2 == nRp) { k2 = 0; ++k1; }
                        }
                    }
                }
            }
            // EVERY slot of this diagonal's ring row is written (slot == u, since i = u mod NT): the
            // cell's scores inside the band, -inf outside it.  Each role commits the states it computed.
            gdouble_w orow = J.sc + 3 * (base + off);
            gu32_w brow = J.bp + 3 * (base + off);
            if (!match_role) {
                SM.sc[d & (RK - 1)][u][PG_X] = bx;
                SM.sc[d & (RK - 1)][u][PG_Y] = by;
                if (all_by_gap) SM.sc[d & (RK - 1)][u][PG_M] = bm;
                if (active) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    d2 xy; xy.x = bx; xy.y = by;
                    *(PG_GLOBAL d2 *)orow = xy;
                    u2 b2; b2.x = px; b2.y = py;
                    *(PG_GLOBAL u2 *)brow = b2;
                    if (all_by_gap) { orow[PG_M] = bm; brow[PG_M] = pm; }
                }
            } else if (!all_by_gap) {
                SM.sc[d & (RK - 1)][u][PG_M] = bm;
                if (active) { orow[PG_M] = bm; brow[PG_M] = pm; }
            }
            resident |= slot_bit;
        }
        // A cell read back from HBM (an edge reaching >= RK diagonals back) must have landed, whichever
        // wave stored it: every wave drains its stores once per RK/2 diagonals, so after the barriers
        // of the following diagonals nothing older than RK diagonals is still in flight.
        if ((d & (RK / 2 - 1)) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                            // (Bd)
    }
}

template __global__ void pg_fill_ring<true>(const PgDevJob *, const int *, unsigned);
template __global__ void pg_fill_ring<false>(const PgDevJob *, const int *, unsigned);

// ---------------------------------------------------------------------------------------------
// End corner and traceback.
//
// pg_end_corner: iterate_bwd_edges_for_end_corner (VA:1440-1552) with score_gap_close
// (VA:2221-2255); one lane per alignment.
//
// Traceback (backtrack_new_path, VA:1038-1189) is a pointer chase through the back-pointers:
// ~(Lx+Ly)/2..(Lx+Ly) dependent HBM reads if done by one lane (70-100 ms for 2 x 100 kb).
// Instead the path is cut at boundary diagonal pairs {k*PG_SEG, k*PG_SEG-1}, k = 1..K -- a move
// lowers d by 1 (gap) or 2 (match), so a path can only skip a pair through a long graph edge:
//   pg_trace_spec     every (cell, state) on every boundary chases to the next lower boundary,
//                     all in parallel, and records where it arrives and after how many cells;
//   pg_trace_compose  one lane per alignment hops from the end cell through those tables
//                     (K dependent reads instead of Lx+Ly), producing (start, length, offset)
//                     per segment; a long edge that jumps over a boundary is chased serially
//                     up to the next one;
//   pg_trace_emit     every segment is chased again in parallel, writing the visited cells
//                     (i, j, w) at its offset; w = the cell's own matrix in bits 0-1 and its
//                     back-pointer bits 2-31 (the cell's `from` label is the next entry's
//                     matrix).  Skip columns and used-edge marks are derived on the host.
namespace {

struct TNode { int i, j, vit; };

__device__ __forceinline__ bool trace_done(const TNode &n) { return n.i < 1 && n.j < 1; }

// One chase step: emit (i, j, w) for the cell `n` names and move to its predecessor.
// A chase is a chain of dependent loads (descriptor of the diagonal -> back-pointer -> edge list); the descriptors of the two
// diagonals a step usually moves to (dd-1: a gap move, dd-2: a match, both through an edge from the previous site) are
// requested together with the back-pointer and kept in `C`, so that a step costs one round trip instead of two.
struct TCache { int d0 = -1000; pg_i4 a, b, c; };     // descriptors {imin, imax, doff low, doff high} of d0, d0-1, d0-2
__device__ __forceinline__ bool trace_step(const View &J, TNode &n, int &w, TCache &C) {
    if (n.vit < 0 || n.vit > 2 || n.i < 0 || n.j < 0 || n.i >= J.Lx || n.j >= J.Ly) return false;
    const int dd = n.i + n.j;
    PG_GLOBAL const pg_i4 *gd = (PG_GLOBAL const pg_i4 *)(unsigned long long)J.dsc;
    const int back = C.d0 - dd;
    pg_i4 D;
    if (back == 0) { D = C.a; }
    else if (back == 1) { D = C.b; C.a = C.b; C.b = C.c; C.c = gd[dd >= 2 ? dd - 2 : 0]; C.d0 = dd; }
    else if (back == 2) { D = C.c; C.a = C.c; C.b = gd[dd >= 1 ? dd - 1 : 0]; C.c = gd[dd >= 2 ? dd - 2 : 0]; C.d0 = dd; }
    else { D = gd[dd]; C.a = D; C.b = gd[dd >= 1 ? dd - 1 : 0]; C.c = gd[dd >= 2 ? dd - 2 : 0]; C.d0 = dd; }
    const int mn = D.x, mx = D.y;
    if (n.i < mn || n.i > mx) return false;
    const long long ix = (((long long)D.w << 32) | (unsigned)D.z) + (n.i - mn);
    // the sites' list offsets are asked for together with the back-pointer (they do not depend on it): a move through a
    // long edge then costs two dependent round trips, not three
    const int oL = J.offL[n.i], oR = J.offR[n.j];
    const unsigned b = J.bp[3 * ix + n.vit];
    w = (int)((unsigned)n.vit | (b & ~3u));
    const unsigned from = b & 3u;
    if (from == PG_BP_NONE) return false;
    const int k1 = (int)((b >> 4) & 16383u), k2 = (int)(b >> 18);
    if (n.vit != PG_Y) n.i = (b & PG_BP_ADJL) ? n.i - 1 : J.srcL[oL + k1];
    if (n.vit != PG_X) n.j = (b & PG_BP_ADJR) ? n.j - 1 : J.srcR[oR + k2];
    n.vit = (int)from;
    return true;
}

// boundary a diagonal belongs to (0 = none): d == k*SEG or d == k*SEG - 1, 1 <= k <= K
__device__ __forceinline__ int boundary_of(int d, int K) {
    if (d <= 0) return 0;
    int k = 0;
    if (d % PG_SEG == 0) k = d / PG_SEG; else if ((d + 1) % PG_SEG == 0) k = (d + 1) / PG_SEG;
    return (k >= 1 && k <= K) ? k : 0;
}

// index of node n inside boundary k's table block
__device__ __forceinline__ int entry_index(const View &J, int k, const TNode &n) {
    const int D = k * PG_SEG, dd = n.i + n.j;
    if (dd == D) return 3 * (n.i - J.imin[D]) + n.vit;
    const int w = J.imax[D] - J.imin[D] + 1;
    return 3 * (w > 0 ? w : 0) + 3 * (n.i - J.imin[D - 1]) + n.vit;
}

enum { EXIT_ENTRY = 0, EXIT_MISS = 1, EXIT_DONE = 2 };

} // namespace

__global__ void pg_end_corner(const PgDevJob *__restrict__ jobs, const int *__restrict__ tiles_gave_up) {
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
    ec[1] = mat; ec[2] = xi; ec[3] = yi; ec[4] = kl; ec[5] = kr; ec[6] = 0; ec[7] = 0;
    ec[0] = (best > NI) ? 0 : 1;
    // the fill kernel gave up on a wait (dp_pipe.hip, poll_ge): no result; the status names the wait
    if (jobs[blockIdx.x].fill_status[0] != 0) ec[0] = 0x40000000 | jobs[blockIdx.x].fill_status[0];
    // the tiled fill (dp_tiles.hip, pg_fill_tiles_flow) abandoned a wait: the waves left the queue and tiles of ANY job of
    // the batch may be unfilled -- only the job whose tile was waiting carries a status of its own.  No job of the batch
    // may report what the arena held before.
    else if (tiles_gave_up && tiles_gave_up[0] != 0) ec[0] = 0x40000000 | 0x7e;
}

// grid (ceil(most table entries of a job / 128), n_jobs): one thread per table entry, i.e. per (cell, state) of a boundary
// pair -- a workgroup per boundary left the widest boundaries' lanes with ten and more chases one after the other, and the
// kernel as long as the longest of those queues
__global__ __launch_bounds__(128) void pg_trace_spec(const PgDevJob *__restrict__ jobs) {
    const View J = load_view(jobs + blockIdx.y);
    if (J.n_bound < 1 || J.endcell[0] != 0) return;
    const int eg = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (eg >= J.tb[J.n_bound + 1]) return;
    int k = 1;                                                     // the last boundary whose first entry is <= eg
    for (int hi = J.n_bound; k < hi;) { const int mid = (k + hi + 1) >> 1; if (J.tb[mid] <= eg) k = mid; else hi = mid - 1; }
    const int D = k * PG_SEG;
    const int mnA = J.imin[D], wA = max(J.imax[D] - mnA + 1, 0);
    const int mnB = J.imin[D - 1], wB = max(J.imax[D - 1] - mnB + 1, 0);
    const int n_entries = 3 * (wA + wB);
    gint_w tab = J.ttab + 8 * (long long)J.tb[k];
    {
        const int e = eg - J.tb[k];
        if (e >= n_entries) return;                                // (cannot happen: tb[k + 1] - tb[k] = n_entries)
        TNode n;
        if (e < 3 * wA) { n.i = mnA + e / 3; n.j = D - n.i; } else { n.i = mnB + (e - 3 * wA) / 3; n.j = D - 1 - n.i; }
        n.vit = e % 3;
        int steps = 0, kind = EXIT_DONE, w;
        bool ok = true;
        TCache tc;
        // the chase aims for the next lower boundary pair {low, low-1} (boundary kb; kb == 0: run to the start).  A long edge
        // may jump over a pair: the chase then goes on to the pair below (pg_trace_compose would otherwise have to walk
        // that stretch cell by cell, a chain of dependent reads -- most of its time at the top of a deep tree)
        // (only where a boundary is narrow -- a banded alignment: there the serial walk is what costs; on the thousands of
        // cells of a full matrix's boundary the longer chases of the entries that are NOT on the path cost more)
        const bool follow = n_entries <= 3 * 512;
        int kb = k - 1, low = (k - 1) * PG_SEG;
        for (;;) {
            if (trace_done(n)) { kind = EXIT_DONE; break; }
            const int dd = n.i + n.j;
            if (steps > 0 && kb >= 1 && dd <= low) {
                if (dd >= low - 1) { kind = EXIT_ENTRY; break; }
                if (!follow) { kind = EXIT_MISS; break; }
                while (kb >= 1 && dd < low - 1) { --kb; low -= PG_SEG; }
                if (kb >= 1 && dd <= low) { kind = EXIT_ENTRY; break; }           // (dd is low or low - 1 now)
            }
            if (steps >= 2 * PG_SEG) { kind = EXIT_MISS; break; }                 // far enough for one entry: pg_trace_compose walks on from here
            if (!trace_step(J, n, w, tc)) { ok = false; break; }
            ++steps;
        }
        // where the chase arrived, and -- when that is a cell of a lower boundary -- its entry there,
        // so that pg_trace_compose hops with one dependent read per boundary
        int next = -1;
        if (ok && kind == EXIT_ENTRY && n.vit >= 0 && n.vit <= 2) {
            const int dd = n.i + n.j, mn = J.imin[dd], mx = J.imax[dd];
            if (n.i >= mn && n.i <= mx) next = J.tb[kb] + entry_index(J, kb, n);
        }
        typedef int i4 __attribute__((ext_vector_type(4)));
        i4 a; a.x = n.i; a.y = n.j; a.z = (n.vit & 3) | (kind << 2); a.w = ok ? steps : -1;
        i4 b; b.x = next; b.y = 0; b.z = 0; b.w = 0;
        *(PG_GLOBAL i4 *)(tab + 8 * e) = a;
        *(PG_GLOBAL i4 *)(tab + 8 * e + 4) = b;
    }
}

__global__ void pg_trace_compose(const PgDevJob *__restrict__ jobs) {
    if (threadIdx.x != 0) return;
    const View J = load_view(jobs + blockIdx.x);
    gint_w ec = J.endcell;
    if (ec[0] != 0) return;
    TNode n; n.vit = ec[1]; n.i = ec[2]; n.j = ec[3];
    const int cap = J.Lx + J.Ly, seg_cap = 2 * J.n_bound + 8;
    int off = 0, nseg = 0, status = 0;
    gint_w tr = J.trace;
    long long cur = -1;                 // table entry of node n, when the previous hop delivered it
    TCache tc;
    while (!trace_done(n)) {
        if (off >= cap) { status = 2; break; }
        if (cur < 0) {
            const int k = boundary_of(n.i + n.j, J.n_bound);
            if (k > 0) {
                const int mnD = J.imin[n.i + n.j], mxD = J.imax[n.i + n.j];
                if (n.i < mnD || n.i > mxD || n.vit < 0 || n.vit > 2) { status = 2; break; }
                cur = (long long)J.tb[k] + entry_index(J, k, n);
            }
        }
        if (cur >= 0) {
            typedef int i4 __attribute__((ext_vector_type(4)));
            PG_GLOBAL const i4 *e = (PG_GLOBAL const i4 *)(J.ttab + 8 * cur);
            const i4 a = e[0], b = e[1];
            const int steps = a.w;
            if (steps <= 0 || nseg >= seg_cap || off + steps > cap) { status = 2; break; }
            gint_w sg = J.segs + 6 * nseg++;
            sg[0] = n.i; sg[1] = n.j; sg[2] = n.vit; sg[3] = steps; sg[4] = off;
            off += steps;
            n.i = a.x; n.j = a.y; n.vit = a.z & 3;
            cur = b.x;                  // -1: a long edge jumped over the next boundary, or the path ended
        } else {
            // not on a boundary pair (the first cells below the end corner, or a long edge that
            // jumped over a pair): chase serially until one is reached
            int w;
            const int ci = n.i, cj = n.j;
            if (!trace_step(J, n, w, tc)) { status = 2; break; }
            tr[3 * off] = ci; tr[3 * off + 1] = cj; tr[3 * off + 2] = w;
            ++off;
        }
    }
    ec[0] = status; ec[6] = off; ec[7] = nseg;
}

// grid (ceil(max_segments / 64), n_jobs), one lane per segment
__global__ __launch_bounds__(64) void pg_trace_emit(const PgDevJob *__restrict__ jobs) {
    const View J = load_view(jobs + blockIdx.y);
    if (J.endcell[0] != 0) return;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= J.endcell[7]) return;
    PG_GLOBAL const int *sg = J.segs + 6 * s;
    TNode n; n.i = sg[0]; n.j = sg[1]; n.vit = sg[2];
    const int steps = sg[3];
    gint_w tr = J.trace + 3 * (long long)sg[4];
    TCache tc;
    for (int t = 0; t < steps; ++t) {
        int w;
        const int ci = n.i, cj = n.j;
        if (!trace_step(J, n, w, tc)) { J.endcell[0] = 2; return; }     // cannot happen: pg_trace_spec walked the same cells
        tr[3 * t] = ci; tr[3 * t + 1] = cj; tr[3 * t + 2] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// The emitted path, checked cell by cell (always on; O(path), not O(cells)).
//
// A back-pointer is written off the fill's dependency chain -- by pg_backptr after the fill, or by pg_fill_pipe's follower
// workgroups WHILE it runs, from scores they read out of L2 on the strength of "a wave's stores of diagonal d have landed
// once it completed d + 8" (dp_pipe.hip).  A word computed from a score read too early would be a valid-looking pointer to
// the wrong predecessor: the host's replay validates indices, not choices.  So every cell the traceback visited is
// re-evaluated here from the stored scores with the comparing kernels' own function (cell_any: candidates in the
// reference's order, strict >, basic_alignment.h:449-462, VA:1038-1189 for what the walk reads) and both the visited
// state's score and its back-pointer must be what is stored, bit for bit.  A difference sets the job's status to
// PG_STATUS_PATH_CHECK; pagan_batch_fetch then runs the batch once more with pg_backptr writing every pointer and, if the
// difference stays, reports PAGAN_E_INTERNAL.
// grid (ceil(longest path / 256), n_jobs), one thread per visited cell.
__global__ __launch_bounds__(256) void pg_trace_check(const PgDevJob *__restrict__ jobs, unsigned flags) {
    const View J = load_view(jobs + blockIdx.y);
    if (J.endcell[0] != 0) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= J.endcell[6]) return;
    const int i = J.trace[3 * t], j = J.trace[3 * t + 1];
    const unsigned w = (unsigned)J.trace[3 * t + 2];
    const int vit = (int)(w & 3u);
    bool ok = vit <= 2 && i >= 0 && j >= 0 && i < J.Lx && j < J.Ly;
    if (ok) {
        const int d = i + j;
        const int lo = J.imin[d], hi = J.imax[d];
        ok = i >= lo && i <= hi;
        if (ok) {
            const long long at = J.doff[d] + (i - lo);
            const Diag none = {0, -1, 0};
            int l0 = 0, l1 = 0, r0 = 0, r1 = 0;
            if (i > 0) { l0 = J.offL[i]; l1 = J.offL[i + 1]; }
            if (j > 0) { r0 = J.offR[j]; r1 = J.offR[j + 1]; }
            float sm = 0.0f;
            if (i > 0 && j > 0 && l1 > l0 && r1 > r0) sm = J.table[J.stL[i] + J.stR[j] * J.S];       // VA:1363
            double b3[3];
            unsigned p3[3];
            cell_any(J, i, j, l1 - l0, r1 - r0, sm, (flags & 1u) != 0, !(flags & 2u),
                     [&](int p, int q, double &xs, double &ys, double &ms) {
                         const long long ix = hbm_index(J, -10, none, none, p, q);
                         xs = ys = ms = neg_inf();
                         if (ix >= 0) { xs = J.sc[3 * ix + PG_X]; ys = J.sc[3 * ix + PG_Y]; ms = J.sc[3 * ix + PG_M]; }
                     },
                     [&](int k, int &p, double &lw) { p = J.srcL[l0 + k]; lw = (double)J.lwL[l0 + k]; },
                     [&](int k, int &q, double &rw) { q = J.srcR[r0 + k]; rw = (double)J.lwR[r0 + k]; },
                     b3[PG_X], b3[PG_Y], b3[PG_M], p3[PG_X], p3[PG_Y], p3[PG_M]);
            double bs = b3[PG_X];
            unsigned ps = p3[PG_X];
            if (vit == PG_Y) { bs = b3[PG_Y]; ps = p3[PG_Y]; } else if (vit == PG_M) { bs = b3[PG_M]; ps = p3[PG_M]; }
            ok = __double_as_longlong(bs) == __double_as_longlong(J.sc[3 * at + vit]) && ps == J.bp[3 * at + vit] &&
                 (ps & ~3u) == (w & ~3u);
        }
    }
    if (!ok) J.endcell[0] = PG_STATUS_PATH_CHECK;
}

// Test hook (pagan_batch_debug_poke_bp): one back-pointer word overwritten between the fill and the traceback.
__global__ void pg_debug_poke_bp(const PgDevJob *__restrict__ jobs, int k, int i, int j, int vit, unsigned word) {
    const View J = load_view(jobs + k);
    const int d = i + j;
    if (threadIdx.x != 0 || d < 0 || d >= J.nd || vit < 0 || vit > 2) return;
    const int lo = J.imin[d], hi = J.imax[d];
    if (i < lo || i > hi) return;
    J.bp[3 * (J.doff[d] + (i - lo)) + vit] = word;
}
