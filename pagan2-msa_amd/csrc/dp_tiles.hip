// dp_tiles.hip -- gfx950 fill kernel for wide matrices (no band, or a band wider than dp_pipe.hip takes):
// the matrix is cut into PG_TILE x PG_TILE tiles, one wave per tile, one launch per tile anti-diagonal.
//
// Why: an anti-diagonal of a full matrix is thousands of cells wide -- more than one compute unit can
// take per step (2,000 cells x ~200 instructions is ~2.7 us of one CU's ALU time) -- while the matrix as a
// whole has work for every CU of the chip.  Tile (a,b) needs tiles (a',b') with a' <= a, b' <= b only
// (bwd edges point to earlier sites), so all tiles with the same a+b are independent: launch t runs
// them concurrently, one wave each, for all wide alignments of the batch at once.  Kernel boundaries
// order the launches, so there is no spinning between workgroups and nothing to get visibility wrong.
//
// Inside a tile the wave sweeps the tile's own anti-diagonals: lane r owns row i0 + r, step s handles
// column j0 + s - r.  The scores of the WHOLE tile plus a halo of eight rows above and eight columns left
// of it (loaded from HBM at the start) live in LDS, so every predecessor inside the tile or within eight
// sites of it is an LDS read; a bwd edge that reaches further back reads HBM (written by an earlier
// launch).  The prologue issues its loads in three dependent rounds, each in flight together (after a kernel
// boundary every first touch goes to memory).  Results go
// to HBM in the batch's diagonal-major layout (dp_device.h) as they are produced; the wave never waits
// for them: the only loads inside the step loop (operands from before the halo) are inline asm with their own
// wait (dp_kcommon.h: far_*) -- a load the compiler can see makes its waitcnt insertion put a vmcnt(0), a wait
// for every store in flight, at the loop header.
//
// Code paths per step, chosen wave-uniformly:
//   simple   every active cell's two sites have exactly one bwd edge, from the previous site: three LDS
//            cells, nine candidates, straight-line;
//   near     otherwise: cells whose sites have at most two bwd edges each, all starting inside the tile or
//            its halo -- eight LDS cells, 24 candidates, straight-line (an absent edge reads a -inf cell);
//   loops    a site with three or more bwd edges, or an edge from before the halo, anywhere in the wave: all
//            active lanes run the reference's loops (the straight-line block would come on top of them, the
//            loops cost what their longest trip costs): edges from LDS windows, the next one requested ahead,
//            (left, right) pairs as one flattened loop, operands from LDS or HBM.
// Back-pointers follow strict-greater in candidate order (first_is_bigger, basic_alignment.h:449-462); values
// are v_max_f64 maxima, which is why jobs with negative-zero parameters go to the comparing HBM wavefront.
//
// LDS: (72 x 72 + 1) cells x 24 B = 124,440 B + column records 2 KB + descriptors 4 KB + edge windows 9 KB +
// model scores 16 KB + table 1 KB = 157 KB of 160 -> one tile per CU.
// The row pitch of 72 cells makes the lanes of a step (stride 71 cells = 213 x 8 B, odd) hit distinct banks.
#include <hip/hip_runtime.h>
#include "dp_device.h"

#include "dp_kcommon.h"
#include <type_traits>

#define TS PG_TILE
#define TH 8                       // halo depth: rows above / columns left of the tile kept in LDS
#define TP (PG_TILE + TH)          // row pitch in cells; (TP - 1) * 3 is odd: the lanes of a step hit distinct 8-byte banks
#define TNULL ((TS + TH) * TP)     // a cell that stays -inf: stands in for the operand of an absent edge
#define TEC PG_TILE_EDGES           // bwd edges of the tile's 64 rows / of its 64 columns (the host keeps other jobs off this kernel)
#define TDB 128                    // descriptors kept for this many diagonals before the tile's first
#define THALO (TH * (TS + TH) + TH * TS)
#define THC 9                      // halo cells a lane has in flight at a time
#define TNL 10                     // far lines (TileSmem::farl)
#define TLINE_COL 0x40000000

namespace {

struct TileSmem {
    double sc[(TS + TH) * TP + 1][3];   // cell (p,q) at (q - j0 + TH) * TP + (p - i0 + TH); TNULL: a cell that stays -inf
    pg_i4 col[TS];                      // column j0 + k: state, first bwd edge (CSR index), edge count | flags, unused
    pg_i4 cole[TS];                     // its first two bwd edges: start site, log weight (float bits), start site, log weight
    pg_i4 dsc[TDB + 2 * TS];            // descriptors of anti-diagonals i0 + j0 - TDB ...: imin, imax, doff low, doff high
    int eL[TEC + 64][2], eR[TEC + 64][2];   // bwd edges of the tile's rows / columns: start site, log weight (float bits)
    // model log score of row i0 + r's state against column j0 + k's, sm[r][k] -- for tables that do not fit LDS.  A small table
    // (S <= 16: `table` below) is looked up directly, and its jobs use the 16 KB for FAR LINES (round 4): the cells of up to
    // TNL rows above / columns left of the halo that bwd edges of the tile's sites start in -- row p as (p, j0-1 .. j0+63),
    // column q as (i0-1 .. i0+63, q) -- loaded with the halo (blocks), so that an operand before the halo is an LDS read
    // instead of a round trip to L2 / HBM in the middle of a step (top of a 512-leaf tree: half of the generic steps made
    // such trips, three to four one after the other, a third of the fill).
    union { float sm[TS][TS]; double farl[TNL][TS + 4][3]; };
    float table[256];                   // the model table (S <= 16)
    unsigned char lineL[TEC + 64], lineR[TEC + 64];     // far line of a bwd edge of the tile's rows / columns (>= TNL: none)
    int line_key[TNL];                  // row lines: the row; column lines: the column | TLINE_COL
    int n_lines;
};

// What a lane keeps about a site: x = state, y = CSR index of its first bwd edge, z = number of bwd
// edges | SITE_SIMPLE (exactly one, from the previous site) | SITE_TWO (at most two) | SITE_FAR0/1 (edge
// 0 / 1 starts before the tile's halo); e = first two edges (start site, float bits of the log weight).
#define SITE_SIMPLE (1 << 30)
#define SITE_TWO (1 << 29)
#define SITE_FAR0 (1 << 28)           // the first / second bwd edge starts outside the LDS window (tile + halo)
#define SITE_FAR1 (1 << 27)
#define SITE_COUNT 0xffff             // PG_MAX_SLOT < 65536
struct SiteRec { pg_i4 r, e; };

// One candidate, in candidate order: it takes the back-pointer only if strictly greater (first_is_bigger,
// basic_alignment.h:449-462); the value is the maximum either way (v_max_f64: one instruction where a
// compare-and-select of a double costs two more).  v_max_f64 may flip the sign of a zero, so the host keeps jobs
// with a negative zero among their parameters off this kernel (dp_abi.hip: has_negative_zero).
// Round 3: SCORES ONLY.  A back-pointer is a function of scores that are final once the fill has passed the cell, and
// pg_backptr (dp_kernels.hip) re-derives all of them after the fill; here a candidate only raises its state's value, so the
// winner's code (and the compare + select that tracked it) is gone from every path of the tile step.
#define PG_TAKE(best, bp, c, code) do { best = __builtin_fmax(best, (c)); } while (0)

} // namespace

static_assert(sizeof(TileSmem) <= 160 * 1024, "TileSmem has to fit the 160 KB of LDS of a gfx950 compute unit");
unsigned pg_tiles_lds_bytes() { return (unsigned)sizeof(TileSmem); }

extern __shared__ __attribute__((aligned(16))) char pg_tiles_lds[];
#define TM (*reinterpret_cast<TileSmem *>(pg_tiles_lds))
#define TAT(p, q) (((q) - j0 + TH) * TP + ((p) - i0 + TH))

// A wait of the dataflow schedule: every wait is for a tile that was handed out earlier to a running wave, so it ends; the
// limit (seconds of polling) only turns a logic error into an error status of the tile's job (pg_end_corner reports it)
// instead of a hung GPU.  After one wave gave up every wait passes and the waves stop taking tiles.
#ifndef PG_FLOW_SPIN_LOG2
#define PG_FLOW_SPIN_LOG2 24
#endif
__device__ __forceinline__ void flow_wait(int *p, int need, int *giveup, int *status) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(16);
        if ((++spins & 63) == 0) {
            if (__hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (spins >= (1 << PG_FLOW_SPIN_LOG2)) {
                __hip_atomic_store(giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x == 0) *status = 0x7e;            // "a tile's wait for its neighbours never ended"
                return;
            }
        }
    }
}

// One tile {job, tile row a, tile column b, -} by one wave.
// LAG (pg_fill_tiles_flow on staircase tile sets): the tile starts while the tiles above (`up`) and to the left (`lf`) are
// still running, TLAG steps behind them.  Lane 0's cell of step s reads row i0-1 at column j0+s, which the tile above
// completes at its step s+63, and lane r's first cell (step r, column j0) reads row i0+r at column j0-1, which the tile to
// the left completes at its step r+63: so the steps [16k, 16k+16) need both neighbours' steps <= 16k+78 published, and from
// step 64 on the tile reads nothing a neighbour has not finished.  The halo is therefore loaded in blocks of TBLK steps'
// worth -- the next TBLK columns of the rows above, the next TBLK rows of the columns to the left -- each after a wait
// for the neighbours' progress (`prog[]`, published with an agent-scope release every TBLK steps), with L2-coherent loads
// (sc1: the cells may have been written by a wave of another XCD a moment ago).  A bwd edge that reaches before the halo
// reads cells (p,q) <= (i,j) of tiles above / to the left, complete at their step <= s+63 as well, or of tiles further
// back, which were TLAG steps ahead of those.
#define TBLK 16                    // (TH + TBLK) * TH + TBLK * TH <= 5 * 64: halo_block's five cells per lane
#define TDONE (1 << 20)
// TAB: the job's model table fits LDS (S <= 16) -- a compile-time fact of the body (the far lines and the table look-up exist
// for such jobs only; a protein walk paid for the other kind's branches and look-ups in every generic step)
template <bool LAG, int TB, bool TAB>
__device__ __noinline__ void tile_body(const PgDevJob *__restrict__ jobs, const pg_i4 T, unsigned flags, int *prog = nullptr,
                                       int self = -1, int up = -1, int lf = -1, int *giveup = nullptr) {
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const View J = load_view(jobs + T.x);
    const int i0 = T.y * TS, j0 = T.z * TS;
    const int r = (int)threadIdx.x;
    const int i = i0 + r;
#ifdef PG_TILE_STATS
    // diagnostic build (tools/build_stamps.sh): cycles and step counts summed over all tiles of the job at the
    // tail of its trace buffer: [0] tiles, [1] prologue, [2..4] steps simple / near / near + general, [5..7] their cycles
    unsigned long long st_n[3] = {0, 0, 0}, st_t[3] = {0, 0, 0};
    unsigned st_far = 0, st_far_steps = 0, st_far_max = 0, st_pairs_max = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    const double NI = neg_inf();
    constexpr bool tab_lds = TAB;        // DNA: 15 states; a protein table (211 x 211) stays in HBM/L2

    // ---- prologue: after a kernel boundary every first touch goes to memory (~1 us), so the loads are issued in
    // three dependent rounds, each round's in flight together, with the LDS-only work between issue and use ----
    PG_GLOBAL const pg_i4 *gdsc = (PG_GLOBAL const pg_i4 *)(unsigned long long)J.dsc;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int dbase = i0 + j0;
    // which halo cell a lane loads in slot u of chunk k0: TH rows above the tile (with the corner), then TH columns left
    auto halo_cell = [&](int k, int &p, int &q) {
        if (k < TH * (TS + TH)) { p = i0 - TH + k / (TS + TH); q = j0 - TH + k % (TS + TH); }
        else { const int k2 = k - TH * (TS + TH); q = j0 - TH + k2 / TS; p = i0 + k2 % TS; }
        return k < THALO && p >= 0 && q >= 0 && p < J.Lx && q < J.Ly;
    };
    // round 1: descriptors of the tile's diagonals, model table, the sites' CSR offsets and states, first halo descriptors
    pg_i4 dv[(TDB + 2 * TS) / 64];
#pragma unroll
    for (int u = 0; u < (TDB + 2 * TS) / 64; ++u) {
        const int d = dbase - TDB + r + 64 * u;
        dv[u] = gdsc[d >= 0 && d < J.nd ? d : 0];
    }
    float tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) tv[u] = J.table[tab_lds && r + 64 * u < J.S * J.S ? r + 64 * u : 0];
    const int jc = j0 + r;
    const bool cv = jc < J.Ly, rv = i < J.Lx;
    const int ca = J.offR[cv ? jc : 0], cb = J.offR[cv ? jc + 1 : 0], cst = J.stR[cv ? jc : 0];
    const int ra = J.offL[rv ? i : 0], rb = J.offL[rv ? i + 1 : 0], rst = J.stL[rv ? i : 0];
    const int eLend = J.offL[i0 + TS < J.Lx ? i0 + TS : J.Lx], eRend = J.offR[j0 + TS < J.Ly ? j0 + TS : J.Ly];
    int hp[THC], hat[THC];
    bool hv[THC];
    pg_i4 hd[THC];
#pragma unroll
    for (int u = 0; u < THC; ++u) {
        int q;
        hv[u] = halo_cell(r + 64 * u, hp[u], q);
        hat[u] = hv[u] ? TAT(hp[u], q) : TNULL;
        hd[u] = gdsc[hv[u] ? hp[u] + q : 0];
    }
    for (int k = r; k < (TS + TH) * TP + 1; k += 64) { TM.sc[k][0] = NI; TM.sc[k][1] = NI; TM.sc[k][2] = NI; }
#pragma unroll
    for (int u = 0; u < (TDB + 2 * TS) / 64; ++u) {
        const int d = dbase - TDB + r + 64 * u;
        TM.dsc[r + 64 * u] = d >= 0 && d < J.nd ? dv[u] : pg_i4{0, -1, 0, 0};
    }
    if (tab_lds) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (r + 64 * u < J.S * J.S) TM.table[r + 64 * u] = tv[u];
    }
    // round 2: the sites' first two bwd edges, the bwd edge windows, the first halo cells, the remaining halo descriptors
    SiteRec row, col;
    auto rec_issue = [&](SiteRec &o, bool valid, int a, int b, int st, gint_p src, gfloat_p lw) {
        o.r = pg_i4{valid ? st : 0, valid ? a : 0, valid ? b - a : 0, 0};
        const int n = o.r.z;
        o.e.x = src[n >= 1 ? a : 0]; o.e.y = __float_as_int(lw[n >= 1 ? a : 0]);
        o.e.z = src[n >= 2 ? a + 1 : 0]; o.e.w = __float_as_int(lw[n >= 2 ? a + 1 : 0]);
    };
    rec_issue(col, cv, ca, cb, cst, J.srcR, J.lwR);
    rec_issue(row, rv, ra, rb, rst, J.srcL, J.lwL);
    const int eL0 = __builtin_amdgcn_readfirstlane(ra), eR0 = __builtin_amdgcn_readfirstlane(ca);
    const int nL = eLend - eL0, nR = eRend - eR0;
    int es[4];
    float ew[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = r + 64 * u;
        es[u] = J.srcL[k < nL ? eL0 + k : 0]; ew[u] = J.lwL[k < nL ? eL0 + k : 0];
        es[2 + u] = J.srcR[k < nR ? eR0 + k : 0]; ew[2 + u] = J.lwR[k < nR ? eR0 + k : 0];
    }
    d2 hxy[THC];
    double hm[THC];
#pragma unroll
    for (int u = 0; u < THC; ++u) {
        hv[u] = !LAG && hv[u] && hp[u] >= hd[u].x && hp[u] <= hd[u].y;
        const long long ix = hv[u] ? (((long long)hd[u].w << 32) | (unsigned)hd[u].z) + (hp[u] - hd[u].x) : 0;
        if (!LAG) {
            hxy[u] = *(PG_GLOBAL const d2 *)(J.sc + 3 * ix);
            hm[u] = J.sc[3 * ix + 2];
        } else { hxy[u].x = NI; hxy[u].y = NI; hm[u] = NI; }
    }
    int gp[THC], gat[THC];
    bool gv[THC];
    pg_i4 gd[THC];
#pragma unroll
    for (int u = 0; u < THC; ++u) {
        int q;
        gv[u] = halo_cell(64 * THC + r + 64 * u, gp[u], q);
        gat[u] = gv[u] ? TAT(gp[u], q) : TNULL;
        gd[u] = gdsc[gv[u] ? gp[u] + q : 0];
    }
    // consume round 2
    auto rec_flags = [&](SiteRec &o, int s, int first) {
        const int n = o.r.z;
        if (n < 1) { o.e.x = 0; o.e.y = 0; }
        if (n < 2) { o.e.z = 0; o.e.w = 0; }
        if (n == 1 && o.e.x == s - 1) o.r.z |= SITE_SIMPLE;
        if (n <= 2) o.r.z |= SITE_TWO;
        if (n >= 1 && o.e.x < first - TH) o.r.z |= SITE_FAR0;
        if (n >= 2 && o.e.z < first - TH) o.r.z |= SITE_FAR1;
    };
    rec_flags(col, jc, j0);
    rec_flags(row, i, i0);
    if (!cv) col.r.z = 0;
    if (!rv) row.r.z = 0;
    TM.col[r] = col.r; TM.cole[r] = col.e;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = r + 64 * u;
        if (k < nL) { TM.eL[k][0] = es[u]; TM.eL[k][1] = __float_as_int(ew[u]); }
        if (k < nR) { TM.eR[k][0] = es[2 + u]; TM.eR[k][1] = __float_as_int(ew[2 + u]); }
    }
    for (int k = r + 128; k < nL && k < TEC; k += 64) { TM.eL[k][0] = J.srcL[eL0 + k]; TM.eL[k][1] = __float_as_int(J.lwL[eL0 + k]); }
    for (int k = r + 128; k < nR && k < TEC; k += 64) { TM.eR[k][0] = J.srcR[eR0 + k]; TM.eR[k][1] = __float_as_int(J.lwR[eR0 + k]); }
#pragma unroll
    for (int u = 0; u < THC; ++u)
        if (hv[u]) { TM.sc[hat[u]][0] = hxy[u].x; TM.sc[hat[u]][1] = hxy[u].y; TM.sc[hat[u]][2] = hm[u]; }
    // round 3: the remaining halo cells (and, for a table too large for LDS, the tile's model scores)
#pragma unroll
    for (int u = 0; u < THC; ++u) {
        gv[u] = !LAG && gv[u] && gp[u] >= gd[u].x && gp[u] <= gd[u].y;
        const long long ix = gv[u] ? (((long long)gd[u].w << 32) | (unsigned)gd[u].z) + (gp[u] - gd[u].x) : 0;
        if (!LAG) {
            hxy[u] = *(PG_GLOBAL const d2 *)(J.sc + 3 * ix);
            hm[u] = J.sc[3 * ix + 2];
        } else { hxy[u].x = NI; hxy[u].y = NI; hm[u] = NI; }
    }
    // the model's scores for the tile's 64 x 64 state pairs (VA:1363): from the table's LDS copy, or -- a
    // protein table is 211 x 211 floats -- from HBM/L2, sixteen loads in flight per lane
    if (!tab_lds) {
        const bool sv = i > 0 && i < J.Lx;
        for (int k0 = 0; k0 < TS; k0 += 16) {
            int at[16];
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) at[u] = TM.col[k0 + u].x;
#pragma unroll
            for (int u = 0; u < 16; ++u) at[u] = (sv && j0 + k0 + u > 0 && j0 + k0 + u < J.Ly) ? row.r.x + at[u] * J.S : 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = J.table[at[u]];
#pragma unroll
            for (int u = 0; u < 16; ++u) TM.sm[r][k0 + u] = v[u];
        }
    }
#pragma unroll
    for (int u = 0; u < THC; ++u)
        if (gv[u]) { TM.sc[gat[u]][0] = hxy[u].x; TM.sc[gat[u]][1] = hxy[u].y; TM.sc[gat[u]][2] = hm[u]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- far lines (small tables only: the memory is sm[][] otherwise) ----
    int n_lines = 0;
    if (tab_lds) {
        for (int k = r; k < TEC + 64; k += 64) { TM.lineL[k] = 255; TM.lineR[k] = 255; }
        if (r == 0) TM.n_lines = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // a line per bwd edge that starts before the halo, as long as the line's diagonals are in the descriptor window
        const int n_row_e = rv ? (row.r.z & SITE_COUNT) : 0, n_col_e = cv ? (col.r.z & SITE_COUNT) : 0;
        for (int k = 0; k < n_row_e && row.r.y - eL0 + k < TEC; ++k) {
            const int p = TM.eL[row.r.y - eL0 + k][0];
            if (p >= i0 - TH || p + j0 - 1 < dbase - TDB) continue;
            const int slot = __hip_atomic_fetch_add(&TM.n_lines, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < TNL) { TM.line_key[slot] = p; TM.lineL[row.r.y - eL0 + k] = (unsigned char)slot; }
        }
        for (int k = 0; k < n_col_e && col.r.y - eR0 + k < TEC; ++k) {
            const int q = TM.eR[col.r.y - eR0 + k][0];
            if (q >= j0 - TH || q + i0 - 1 < dbase - TDB) continue;
            const int slot = __hip_atomic_fetch_add(&TM.n_lines, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < TNL) { TM.line_key[slot] = q | TLINE_COL; TM.lineR[col.r.y - eR0 + k] = (unsigned char)slot; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        n_lines = TM.n_lines < TNL ? TM.n_lines : TNL;
        n_lines = __builtin_amdgcn_readfirstlane(n_lines);
        for (int k = r; k < n_lines * (TS + 4); k += 64) { double *c_ = &TM.farl[0][0][0] + 3 * k; c_[0] = NI; c_[1] = NI; c_[2] = NI; }
        if (!LAG) {
            // the tiles the lines lie in are finished: every cell now
            for (int e = r; e < n_lines * (TS + 1); e += 64) {
                const int line = e / (TS + 1), cc = e % (TS + 1), key = TM.line_key[line];
                const int p = (key & TLINE_COL) ? i0 - 1 + cc : key, q = (key & TLINE_COL) ? (key & ~TLINE_COL) : j0 - 1 + cc;
                if (p < 0 || q < 0 || p >= J.Lx || q >= J.Ly) continue;
                const pg_i4 F = TM.dsc[p + q - (dbase - TDB)];
                if (p < F.x || p > F.y) continue;
                const long long ix = (((long long)F.w << 32) | (unsigned)F.z) + (p - F.x);
                TM.farl[line][cc][0] = J.sc[3 * ix]; TM.farl[line][cc][1] = J.sc[3 * ix + 1]; TM.farl[line][cc][2] = J.sc[3 * ix + 2];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    }
    // the model's score of (row i0 + r, the column whose record is cw): a small table is looked up (its LDS copy), a large
    // one was gathered into sm[][] above
    // (a row or column that is no site of the matrix carries state 0 in its record, and its cells have no M candidate: any
    //  entry of the table does)
    auto model_score = [&](int jj_, int cstate) -> float {
        if (!tab_lds) return TM.sm[r][jj_];
        return TM.table[row.r.x + cstate * J.S];
    };

    const double go = (double)J.go, ng = (double)J.ng;
    const int nl = row.r.z & SITE_COUNT;
    const int p0 = row.e.x, p1 = row.e.z;
    const double lw0 = (double)__int_as_float(row.e.y), lw1 = (double)__int_as_float(row.e.w);
    const double extY = (double)(((i == 0 || i == J.Lx - 1) && !no_terminal_edges) ? J.gE : J.ge);   // VA:875-879
    const double openX0 = (reduced_terminal && p0 == 0) ? 0.0 : go, openX1 = (reduced_terminal && p1 == 0) ? 0.0 : go;  // BA.h:490-513
    const int s_last = (2 * TS - 2 < J.nd - 1 - dbase) ? 2 * TS - 2 : J.nd - 1 - dbase;
    // operands of a step are fetched one step ahead: the diagonal's descriptor and the column's record
    pg_i4 D = TM.dsc[TDB];
    pg_i4 c = pg_i4{0, 0, 0, 0}, ce = pg_i4{0, 0, 0, 0};
    if (r == 0) { c = TM.col[0]; ce = TM.cole[0]; }
#ifdef PG_TILE_STATS
    const unsigned long long st_loop = __builtin_amdgcn_s_memtime();
#endif
    // LAG: the halo cells of one block of steps (k = s / TB, s < TS): columns j0 + TB k ... of the TH rows above (the
    // first block also takes the corner), rows i0 + TB k ... of the TH columns to the left
    auto halo_block = [&](int k) {
        const int c0 = k == 0 ? -TH : TB * k, ncol = k == 0 ? TH + TB : TB;
        const int ntop = ncol * TH, nall = ntop + TB * TH;
        // at most five cells per lane ((TH + TB) * TH + TB * TH = 320): requested together, one wait
        bool hv_[5];
        int hat_[5];
        PG_GLOBAL const double *hp_[5];
        d2 hxy_[5];
        double hm_[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int e = r + 64 * u;
            int p, q;
            if (e < ntop) { p = i0 - TH + e / ncol; q = j0 + c0 + e % ncol; }
            else { const int e2 = e - ntop; q = j0 - TH + e2 / TB; p = i0 + TB * k + e2 % TB; }
            hv_[u] = e < nall && p >= 0 && q >= 0 && p < J.Lx && q < J.Ly;
            const pg_i4 F = TM.dsc[hv_[u] ? p + q - (dbase - TDB) : TDB];
            hv_[u] = hv_[u] && p >= F.x && p <= F.y;
            const long long ix = hv_[u] ? (((long long)F.w << 32) | (unsigned)F.z) + (p - F.x) : 0;
            hp_[u] = J.sc + 3 * ix;
            hat_[u] = hv_[u] ? TAT(p, q) : TNULL;
            hxy_[u].x = NI; hxy_[u].y = NI; hm_[u] = NI;
        }
        const unsigned long long k0 = __builtin_amdgcn_ballot_w64(hv_[0]), k1 = __builtin_amdgcn_ballot_w64(hv_[1]);
        const unsigned long long k2 = __builtin_amdgcn_ballot_w64(hv_[2]), k3 = __builtin_amdgcn_ballot_w64(hv_[3]);
        const unsigned long long k4 = __builtin_amdgcn_ballot_w64(hv_[4]);
        if ((k0 | k1 | k2 | k3 | k4) == 0) return;
        unsigned long long sv;
#define PG_HALO_LD(n) "s_and_b64 exec, %[sv], %[k" #n "]\n\tglobal_load_dwordx4 %[x" #n "], %[p" #n "], off sc1\n\tglobal_load_dwordx2 %[u" #n "], %[p" #n "], off offset:16 sc1\n\t"
        asm volatile("s_mov_b64 %[sv], exec\n\t" PG_HALO_LD(0) PG_HALO_LD(1) PG_HALO_LD(2) PG_HALO_LD(3) PG_HALO_LD(4)
                     "s_mov_b64 exec, %[sv]\n\ts_waitcnt vmcnt(0)"
                     : [x0] "+v"(hxy_[0]), [u0] "+v"(hm_[0]), [x1] "+v"(hxy_[1]), [u1] "+v"(hm_[1]), [x2] "+v"(hxy_[2]), [u2] "+v"(hm_[2]),
                       [x3] "+v"(hxy_[3]), [u3] "+v"(hm_[3]), [x4] "+v"(hxy_[4]), [u4] "+v"(hm_[4]), [sv] "=&s"(sv)
                     : [p0] "v"(hp_[0]), [p1] "v"(hp_[1]), [p2] "v"(hp_[2]), [p3] "v"(hp_[3]), [p4] "v"(hp_[4]),
                       [k0] "s"(k0), [k1] "s"(k1), [k2] "s"(k2), [k3] "s"(k3), [k4] "s"(k4)
                     : "memory");
#undef PG_HALO_LD
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (hv_[u]) { TM.sc[hat_[u]][0] = hxy_[u].x; TM.sc[hat_[u]][1] = hxy_[u].y; TM.sc[hat_[u]][2] = hm_[u]; }
    };
    // LAG: the far lines' cells of one block of steps -- a row line's next TB columns, a column line's next TB rows (the first
    // block also takes the cell at j0-1 / i0-1); they lie in tiles that are further along than the halo's (rows above the
    // halo complete before the halo's rows do, columns left of it before its columns), so the halo's wait covers them
    auto line_block = [&](int k) {
        if (n_lines == 0) return;
        const int per = TB + (k == 0 ? 1 : 0), nall = n_lines * per;
        bool hv_[3];
        int hl_[3], hc_[3];
        PG_GLOBAL const double *hp_[3];
        d2 hxy_[3];
        double hm_[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = r + 64 * u;
            const int line = e < nall ? e / per : 0, w = e % per;
            const int cc = (k == 0) ? w : TB * k + 1 + w;          // cell of the line: 0 .. TS
            const int key = TM.line_key[line];
            const int p = (key & TLINE_COL) ? i0 - 1 + cc : key, q = (key & TLINE_COL) ? (key & ~TLINE_COL) : j0 - 1 + cc;
            hv_[u] = e < nall && cc <= TS && p >= 0 && q >= 0 && p < J.Lx && q < J.Ly;
            const pg_i4 F = TM.dsc[hv_[u] ? p + q - (dbase - TDB) : TDB];
            hv_[u] = hv_[u] && p >= F.x && p <= F.y;
            const long long ix = hv_[u] ? (((long long)F.w << 32) | (unsigned)F.z) + (p - F.x) : 0;
            hp_[u] = J.sc + 3 * ix;
            hl_[u] = line; hc_[u] = cc;
            hxy_[u].x = NI; hxy_[u].y = NI; hm_[u] = NI;
        }
        const unsigned long long k0 = __builtin_amdgcn_ballot_w64(hv_[0]), k1 = __builtin_amdgcn_ballot_w64(hv_[1]);
        const unsigned long long k2 = __builtin_amdgcn_ballot_w64(hv_[2]);
        if ((k0 | k1 | k2) == 0) return;
        unsigned long long sv;
#define PG_LINE_LD(n) "s_and_b64 exec, %[sv], %[k" #n "]\n\tglobal_load_dwordx4 %[x" #n "], %[p" #n "], off sc1\n\tglobal_load_dwordx2 %[u" #n "], %[p" #n "], off offset:16 sc1\n\t"
        asm volatile("s_mov_b64 %[sv], exec\n\t" PG_LINE_LD(0) PG_LINE_LD(1) PG_LINE_LD(2)
                     "s_mov_b64 exec, %[sv]\n\ts_waitcnt vmcnt(0)"
                     : [x0] "+v"(hxy_[0]), [u0] "+v"(hm_[0]), [x1] "+v"(hxy_[1]), [u1] "+v"(hm_[1]), [x2] "+v"(hxy_[2]), [u2] "+v"(hm_[2]), [sv] "=&s"(sv)
                     : [p0] "v"(hp_[0]), [p1] "v"(hp_[1]), [p2] "v"(hp_[2]), [k0] "s"(k0), [k1] "s"(k1), [k2] "s"(k2)
                     : "memory");
#undef PG_LINE_LD
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (hv_[u]) { TM.farl[hl_[u]][hc_[u]][0] = hxy_[u].x; TM.farl[hl_[u]][hc_[u]][1] = hxy_[u].y; TM.farl[hl_[u]][hc_[u]][2] = hm_[u]; }
    };
    auto wait_prog = [&](int which, int need) {
        if (which < 0) return;
        flow_wait(&prog[which], need, giveup, jobs[T.x].fill_status);
    };
    for (int s = 0; s <= s_last; ++s) {
#ifdef PG_TILE_STATS
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
        int st_kind = -1;
#endif
        if (LAG && s < TS && (s & (TB - 1)) == 0) {
            const int need = s + TB - 1 + TS;                   // the neighbours' steps <= s + TB - 1 + 63 are published
#ifdef PG_TILE_STATS
            const unsigned long long lw0 = __builtin_amdgcn_s_memtime();
#endif
            wait_prog(up, need < 2 * TS - 1 ? need : TDONE);
            wait_prog(lf, need < 2 * TS - 1 ? need : TDONE);
#ifdef PG_TILE_STATS
            const unsigned long long lw1 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            halo_block(s / TB);
            line_block(s / TB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PG_TILE_STATS
            if (r == 0 && 3 * (J.Lx + J.Ly) >= 4096) {      // (long jobs only: the counters borrow the tail of the trace buffer) [13] waiting for the neighbours' progress, [14] acquire + halo block
                unsigned long long *out = (unsigned long long *)(jobs[T.x].trace + ((3 * (J.Lx + J.Ly) - 64) & ~1));
                atomicAdd(out + 13, lw1 - lw0); atomicAdd(out + 14, __builtin_amdgcn_s_memtime() - lw1);
            }
#endif
        }
        const pg_i4 Dn = TM.dsc[TDB + s + 1];
        const int jj = s - r;
        pg_i4 cn = pg_i4{0, 0, 0, 0}, cen = pg_i4{0, 0, 0, 0};
        if (jj + 1 >= 0 && jj + 1 < TS) { cn = TM.col[jj + 1]; cen = TM.cole[jj + 1]; }
        const int mn = D.x, mx = D.y;
        const bool active = jj >= 0 && jj < TS && i >= mn && i <= mx;
        if (__builtin_amdgcn_ballot_w64(active) != 0) {
            const long long off = ((long long)D.w << 32) | (unsigned)D.z;
            const int j = j0 + jj;
            const bool simple = (row.r.z & c.z & SITE_SIMPLE) != 0;
            const bool two = (row.r.z & c.z & SITE_TWO) != 0;
            const bool anyfar = ((row.r.z | c.z) & (SITE_FAR0 | SITE_FAR1)) != 0;
            double bx = NI, by = NI, bm = NI;
            unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
#ifdef PG_TILE_STATS
            st_kind = __builtin_amdgcn_ballot_w64(active && !simple) == 0 ? 0 : (__builtin_amdgcn_ballot_w64(active && !(two && !anyfar)) == 0 ? 1 : 2);
#endif
            if (__builtin_amdgcn_ballot_w64(active && !simple) == 0) {
                if (active) {
                    // (i-1,j), (i,j-1), (i-1,j-1)
                    const int a_up = TAT(i - 1, j), a_left = TAT(i, j - 1), a_diag = TAT(i - 1, j - 1);
                    const double ux = TM.sc[a_up][0], uy = TM.sc[a_up][1], um = TM.sc[a_up][2];
                    const double lx = TM.sc[a_left][0], ly = TM.sc[a_left][1], lm = TM.sc[a_left][2];
                    const double gx = TM.sc[a_diag][0], gy = TM.sc[a_diag][1], gm = TM.sc[a_diag][2];
                    const float sm = model_score(jj, c.x);   // VA:1363
                    const double rw = (double)__int_as_float(ce.y);
                    {
                        const bool end_gap = (j == J.Ly - 1) && !no_terminal_edges;          // VA:864-868 (j > 0 here)
                        const double ext = (double)(end_gap ? J.gE : J.ge);
                        PG_TAKE(bx, px, ux + ext, pack_bp(PG_X, 0, 0, true, false));
                        PG_TAKE(bx, px, (uy + 0.0) + go, pack_bp(PG_Y, 0, 0, true, false));
                        PG_TAKE(bx, px, (um + ng) + openX0, pack_bp(PG_M, 0, 0, true, false));
                    }
                    {
                        const double open = (reduced_terminal && j == 1) ? 0.0 : go;
                        PG_TAKE(by, py, ly + extY, pack_bp(PG_Y, 0, 0, false, true));
                        PG_TAKE(by, py, (lx + 0.0) + go, pack_bp(PG_X, 0, 0, false, true));
                        PG_TAKE(by, py, (lm + ng) + open, pack_bp(PG_M, 0, 0, false, true));
                    }
                    {
                        const double tM = (double)(2 * J.ng) + (double)sm;                   // VA:1364
                        const double tX = (double)(0.0f + J.ng) + (double)sm;                // VA:1366-1367
                        PG_TAKE(bm, pm, ((gm + tM) + lw0) + rw, pack_bp(PG_M, 0, 0, true, true));
                        PG_TAKE(bm, pm, ((gx + tX) + lw0) + rw, pack_bp(PG_X, 0, 0, true, true));
                        PG_TAKE(bm, pm, ((gy + tX) + lw0) + rw, pack_bp(PG_Y, 0, 0, true, true));
                    }
                }
            } else {
                const int nr = c.z & SITE_COUNT;
                const int q0 = ce.x, q1 = ce.z;
                const bool l0 = nl >= 1, l1 = nl >= 2, r0 = nr >= 1, r1 = nr >= 2;
                // one lane that needs the loops takes the whole step there: the loops cost what their longest trip
                // costs, the straight-line block would come on top
                const bool near = __builtin_amdgcn_ballot_w64(active && !(two && !anyfar)) == 0;
                // a simple site on at least one side of every cell of the wave (the usual case: two multi-edge sites in one
                // cell need a gap on both sides): the half-size block
                const bool one = ((row.r.z | c.z) & SITE_SIMPLE) != 0;
                const bool half = near && __builtin_amdgcn_ballot_w64(active && !one) == 0;
                if (active && half) {
                    // The simple side's gap state comes from its one (adjacent) edge; the other side's gap state and M from
                    // that side's at most two edges, paired with the simple side's one.  Which side is which differs between
                    // lanes: operands are selected per lane, the sums associate as in the full block below.
                    const bool left = !(row.r.z & SITE_SIMPLE);                      // the (possibly) multi-edge site is the left one
                    const double rw0 = (double)__int_as_float(ce.y), rw1 = (double)__int_as_float(ce.w);
                    const int g0 = left ? p0 : q0, g1 = left ? p1 : q1;              // where that site's edges start
                    const bool h0 = left ? l0 : r0, h1 = left ? l1 : r1;
                    const double gw0 = left ? lw0 : rw0, gw1 = left ? lw1 : rw1, sw = left ? rw0 : lw0;
                    const int aS = left ? TAT(i, j - 1) : TAT(i - 1, j);
                    const int aA0 = h0 ? (left ? TAT(g0, j) : TAT(i, g0)) : TNULL, aA1 = h1 ? (left ? TAT(g1, j) : TAT(i, g1)) : TNULL;
                    const int aB0 = h0 ? (left ? TAT(g0, j - 1) : TAT(i - 1, g0)) : TNULL;
                    const int aB1 = h1 ? (left ? TAT(g1, j - 1) : TAT(i - 1, g1)) : TNULL;
                    const double sx = TM.sc[aS][0], sy = TM.sc[aS][1], ss = TM.sc[aS][2];
                    const double a0x = TM.sc[aA0][0], a0y = TM.sc[aA0][1], a0m = TM.sc[aA0][2];
                    const double a1x = TM.sc[aA1][0], a1y = TM.sc[aA1][1], a1m = TM.sc[aA1][2];
                    const double b0x = TM.sc[aB0][0], b0y = TM.sc[aB0][1], b0m = TM.sc[aB0][2];
                    const double b1x = TM.sc[aB1][0], b1y = TM.sc[aB1][1], b1m = TM.sc[aB1][2];
                    const float sm = (i > 0 && j > 0) ? model_score(jj, c.x) : 0.0f;
                    const double extX = (double)(((j == 0 || j == J.Ly - 1) && !no_terminal_edges) ? J.gE : J.ge);   // VA:864-868
                    const unsigned adjG = left ? PG_BP_ADJL : PG_BP_ADJR, adjS = left ? PG_BP_ADJR : PG_BP_ADJL;
                    const unsigned selfG = left ? PG_X : PG_Y, crossG = left ? PG_Y : PG_X;
                    const int kshift = left ? 4 : 18;
                    const int nearG = left ? i - 1 : j - 1;                          // the site an adjacent edge of that side starts at
                    double gs = NI, ssb = NI;
                    unsigned pg = PG_BP_NONE, ps = PG_BP_NONE;
                    {   // the simple side's gap state: one operand, own state first (VA:898-915 / 927-944)
                        const double extS = left ? extY : extX;
                        const int srcS = left ? j - 1 : i - 1;
                        const double open = (reduced_terminal && srcS == 0) ? 0.0 : go;
                        PG_TAKE(ssb, ps, (left ? sy : sx) + extS, crossG | adjS);
                        PG_TAKE(ssb, ps, ((left ? sx : sy) + 0.0) + go, selfG | adjS);
                        PG_TAKE(ssb, ps, (ss + ng) + open, PG_M | adjS);
                    }
                    {   // the other side's gap state: per edge own state, the other gap state, M
                        const double extG = left ? extX : extY;
                        const unsigned e0 = g0 == nearG ? adjG : 0u, e1 = (g1 == nearG ? adjG : 0u) | (1u << kshift);
                        const double o0 = (reduced_terminal && g0 == 0) ? 0.0 : go, o1 = (reduced_terminal && g1 == 0) ? 0.0 : go;
                        PG_TAKE(gs, pg, (left ? a0x : a0y) + extG, e0 | selfG);
                        PG_TAKE(gs, pg, ((left ? a0y : a0x) + 0.0) + go, e0 | crossG);
                        PG_TAKE(gs, pg, (a0m + ng) + o0, e0 | PG_M);
                        PG_TAKE(gs, pg, (left ? a1x : a1y) + extG, e1 | selfG);
                        PG_TAKE(gs, pg, ((left ? a1y : a1x) + 0.0) + go, e1 | crossG);
                        PG_TAKE(gs, pg, (a1m + ng) + o1, e1 | PG_M);
                    }
                    {   // M: the (left edge, right edge) pairs in list order (VA:1396-1433)
                        const double tM = (double)(2 * J.ng) + (double)sm;                   // VA:1364
                        const double tX = (double)(0.0f + J.ng) + (double)sm;                // VA:1366-1367
                        const double lwA0 = left ? gw0 : sw, rwA0 = left ? sw : gw0, lwA1 = left ? gw1 : sw, rwA1 = left ? sw : gw1;
                        const unsigned e0 = (g0 == nearG ? adjG : 0u) | adjS, e1 = (g1 == nearG ? adjG : 0u) | (1u << kshift) | adjS;
                        PG_TAKE(bm, pm, ((b0m + tM) + lwA0) + rwA0, e0 | PG_M);
                        PG_TAKE(bm, pm, ((b0x + tX) + lwA0) + rwA0, e0 | PG_X);
                        PG_TAKE(bm, pm, ((b0y + tX) + lwA0) + rwA0, e0 | PG_Y);
                        PG_TAKE(bm, pm, ((b1m + tM) + lwA1) + rwA1, e1 | PG_M);
                        PG_TAKE(bm, pm, ((b1x + tX) + lwA1) + rwA1, e1 | PG_X);
                        PG_TAKE(bm, pm, ((b1y + tX) + lwA1) + rwA1, e1 | PG_Y);
                    }
                    bx = left ? gs : ssb; px = left ? pg : ps;
                    by = left ? ssb : gs; py = left ? ps : pg;
                }
                if (active && near && !half) {
                    // At most two bwd edges per site, all eight operand cells in LDS: straight-line, the cell of an
                    // absent edge is the -inf cell (its candidates never win).  Order: SURVEY.md Appendix A.
                    const double rw0 = (double)__int_as_float(ce.y), rw1 = (double)__int_as_float(ce.w);
                    const int aX0 = l0 ? TAT(p0, j) : TNULL, aX1 = l1 ? TAT(p1, j) : TNULL;
                    const int aY0 = r0 ? TAT(i, q0) : TNULL, aY1 = r1 ? TAT(i, q1) : TNULL;
                    const int aM00 = (l0 && r0) ? TAT(p0, q0) : TNULL, aM01 = (l0 && r1) ? TAT(p0, q1) : TNULL;
                    const int aM10 = (l1 && r0) ? TAT(p1, q0) : TNULL, aM11 = (l1 && r1) ? TAT(p1, q1) : TNULL;
                    float sm = 0.0f;
                    if (l0 && r0 && i > 0 && j > 0) sm = model_score(jj, c.x);
                    {
                        const double x0 = TM.sc[aX0][0], y0 = TM.sc[aX0][1], m0 = TM.sc[aX0][2];
                        const double x1 = TM.sc[aX1][0], y1 = TM.sc[aX1][1], m1 = TM.sc[aX1][2];
                        const bool end_gap = (j == 0 || j == J.Ly - 1) && !no_terminal_edges;    // VA:864-868
                        const double ext = (double)(end_gap ? J.gE : J.ge);
                        PG_TAKE(bx, px, x0 + ext, pack_bp(PG_X, 0, 0, p0 == i - 1, false));
                        PG_TAKE(bx, px, (y0 + 0.0) + go, pack_bp(PG_Y, 0, 0, p0 == i - 1, false));
                        PG_TAKE(bx, px, (m0 + ng) + openX0, pack_bp(PG_M, 0, 0, p0 == i - 1, false));
                        PG_TAKE(bx, px, x1 + ext, pack_bp(PG_X, 1, 0, p1 == i - 1, false));
                        PG_TAKE(bx, px, (y1 + 0.0) + go, pack_bp(PG_Y, 1, 0, p1 == i - 1, false));
                        PG_TAKE(bx, px, (m1 + ng) + openX1, pack_bp(PG_M, 1, 0, p1 == i - 1, false));
                    }
                    {
                        const double x0 = TM.sc[aY0][0], y0 = TM.sc[aY0][1], m0 = TM.sc[aY0][2];
                        const double x1 = TM.sc[aY1][0], y1 = TM.sc[aY1][1], m1 = TM.sc[aY1][2];
                        const double open0 = (reduced_terminal && q0 == 0) ? 0.0 : go, open1 = (reduced_terminal && q1 == 0) ? 0.0 : go;
                        PG_TAKE(by, py, y0 + extY, pack_bp(PG_Y, 0, 0, false, q0 == j - 1));
                        PG_TAKE(by, py, (x0 + 0.0) + go, pack_bp(PG_X, 0, 0, false, q0 == j - 1));
                        PG_TAKE(by, py, (m0 + ng) + open0, pack_bp(PG_M, 0, 0, false, q0 == j - 1));
                        PG_TAKE(by, py, y1 + extY, pack_bp(PG_Y, 0, 1, false, q1 == j - 1));
                        PG_TAKE(by, py, (x1 + 0.0) + go, pack_bp(PG_X, 0, 1, false, q1 == j - 1));
                        PG_TAKE(by, py, (m1 + ng) + open1, pack_bp(PG_M, 0, 1, false, q1 == j - 1));
                    }
                    {
                        const double tM = (double)(2 * J.ng) + (double)sm;                   // VA:1364
                        const double tX = (double)(0.0f + J.ng) + (double)sm;                // VA:1366-1367
#define PG_PAIR(a, k1, k2, lw, rw, pp, qq)                                                                          \
                        {                                                                                           \
                            const double x_ = TM.sc[a][0], y_ = TM.sc[a][1], m_ = TM.sc[a][2];                      \
                            PG_TAKE(bm, pm, ((m_ + tM) + lw) + rw, pack_bp(PG_M, k1, k2, pp == i - 1, qq == j - 1)); \
                            PG_TAKE(bm, pm, ((x_ + tX) + lw) + rw, pack_bp(PG_X, k1, k2, pp == i - 1, qq == j - 1)); \
                            PG_TAKE(bm, pm, ((y_ + tX) + lw) + rw, pack_bp(PG_Y, k1, k2, pp == i - 1, qq == j - 1)); \
                        }
                        PG_PAIR(aM00, 0, 0, lw0, rw0, p0, q0)
                        PG_PAIR(aM01, 0, 1, lw0, rw1, p0, q1)
                        PG_PAIR(aM10, 1, 0, lw1, rw0, p1, q0)
                        PG_PAIR(aM11, 1, 1, lw1, rw1, p1, q1)
#undef PG_PAIR
                    }
                    if (i == 0 && j == 0) bm = 0.0;                                          // initialise_array_corner, VA:725-736
                }
                if (!near) {
#ifdef PG_TILE_STATS
                    const unsigned far_before = st_far;
                    {
                        int pr_ = active ? (int)(nl > 0 ? nl : 1) * (int)(nr > 0 ? nr : 1) : 0;
                        for (int o_ = 32; o_ > 0; o_ >>= 1) pr_ = max(pr_, __shfl_xor(pr_, o_));
                        st_pairs_max += pr_;
                    }
#endif
                    // (compiled twice: with the far lines' look-ups for small-table jobs, without them for the others -- a protein
                    //  walk paid 10 % for look-ups it never uses)
                    auto loops = [&](auto with_lines) {
                        constexpr bool LINES = decltype(with_lines)::value;
                        // Any number of bwd edges, anywhere: the reference's loops, edges from the
                        // LDS windows (the next one requested while the current one is worked on), operand cells from
                        // the LDS window or, before the halo, from HBM.
                        // sl / sr: the far line of the left / right edge the operand comes through (>= TNL: none)
                        auto fetch = [&](int p, int q, int sl, int sr, double &xs, double &ys, double &ms) {
                            if (p >= i0 - TH && q >= j0 - TH) {
                                const int at = TAT(p, q);
                                xs = TM.sc[at][0]; ys = TM.sc[at][1]; ms = TM.sc[at][2];
                            } else if (LINES && sl < TNL && p < i0 - TH && q >= j0 - 1) {
                                const double *c_ = TM.farl[sl][q - (j0 - 1)];
                                xs = c_[0]; ys = c_[1]; ms = c_[2];
                            } else if (LINES && sr < TNL && q < j0 - TH && p >= i0 - 1) {
                                const double *c_ = TM.farl[sr][p - (i0 - 1)];
                                xs = c_[0]; ys = c_[1]; ms = c_[2];
                            } else {
#ifdef PG_TILE_STATS
                                ++st_far;
#endif
                                const int w = p + q - (dbase - TDB);
                                const pg_i4 F = w >= 0 ? TM.dsc[w] : far_desc(((PG_GLOBAL const pg_i4 *)(unsigned long long)J.dsc) + (p + q));
                                xs = ys = ms = NI;
#ifndef PG_TILE_EXP_NOFAR                                       // (timing experiment, wrong results: what would the generic steps cost without their far loads?)
                                if (p >= F.x && p <= F.y) {
                                    const long long ix = (((long long)F.w << 32) | (unsigned)F.z) + (p - F.x);
                                    far_cell(J.sc + 3 * ix, xs, ys, ms);
                                }
#endif
                            }
                        };
                        typedef int i2 __attribute__((ext_vector_type(2)));
                        const int eLi = row.r.y - eL0, eRi = c.y - eR0;
                        if (i == 0 && j == 0) bm = 0.0;                                      // initialise_array_corner, VA:725-736
                        if (nl > 0) {                                                        // X (VA:898-915); nl > 0 implies i > 0
                            const bool end_gap = (j == 0 || j == J.Ly - 1) && !no_terminal_edges;
                            const double ext = (double)(end_gap ? J.gE : J.ge);
                            // two stages: while the candidates of edge k are taken, the cell of edge k+1 is on its way
                            i2 en = *(const i2 *)TM.eL[eLi + 1];
                            int p = TM.eL[eLi][0];
                            double xs, ys, ms;
                            fetch(p, j, LINES ? TM.lineL[eLi] : 255, 255, xs, ys, ms);
                            for (int k = 0; k < nl; ++k) {
                                const int pn = en.x;
                                double nx = NI, ny = NI, nm = NI;
                                if (k + 1 < nl) fetch(pn, j, LINES ? TM.lineL[eLi + k + 1] : 255, 255, nx, ny, nm);
                                en = *(const i2 *)TM.eL[eLi + k + 2];
                                const double open = (reduced_terminal && p == 0) ? 0.0 : go;
                                const unsigned base = ((unsigned)k << 4) | (p == i - 1 ? PG_BP_ADJL : 0u);
                                PG_TAKE(bx, px, xs + ext, base | PG_X);
                                PG_TAKE(bx, px, (ys + 0.0) + go, base | PG_Y);
                                PG_TAKE(bx, px, (ms + ng) + open, base | PG_M);
                                p = pn; xs = nx; ys = ny; ms = nm;
                            }
                        }
                        if (nr > 0) {                                                        // Y (VA:927-944)
                            i2 en = *(const i2 *)TM.eR[eRi + 1];
                            int q = TM.eR[eRi][0];
                            double xs, ys, ms;
                            fetch(i, q, 255, LINES ? TM.lineR[eRi] : 255, xs, ys, ms);
                            for (int k = 0; k < nr; ++k) {
                                const int qn = en.x;
                                double nx = NI, ny = NI, nm = NI;
                                if (k + 1 < nr) fetch(i, qn, 255, LINES ? TM.lineR[eRi + k + 1] : 255, nx, ny, nm);
                                en = *(const i2 *)TM.eR[eRi + k + 2];
                                const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                                const unsigned base = ((unsigned)k << 18) | (q == j - 1 ? PG_BP_ADJR : 0u);
                                PG_TAKE(by, py, ys + extY, base | PG_Y);
                                PG_TAKE(by, py, (xs + 0.0) + go, base | PG_X);
                                PG_TAKE(by, py, (ms + ng) + open, base | PG_M);
                                q = qn; xs = nx; ys = ny; ms = nm;
                            }
                        }
                        if (nl > 0 && nr > 0) {                                              // M (VA:956-963, 1353-1436)
                            const float sm = model_score(jj, c.x);
                            const double tM = (double)(2 * J.ng) + (double)sm;               // VA:1364
                            const double tX = (double)(0.0f + J.ng) + (double)sm;            // VA:1366-1367
                            // the (left edge, right edge) pairs row-major as ONE loop: the next pair's edges are requested
                            // before the current pair's cell
                            const int pairs = nl * nr;
                            int k1 = 0, k2 = 0;
                            i2 e1 = *(const i2 *)TM.eL[eLi], e2 = *(const i2 *)TM.eR[eRi];
                            for (int t = 0; t < pairs; ++t) {
                                const int p = e1.x, q = e2.x;
                                const double lw = (double)__int_as_float(e1.y), rw = (double)__int_as_float(e2.y);
                                const unsigned base = ((unsigned)k1 << 4) | ((unsigned)k2 << 18) | (p == i - 1 ? PG_BP_ADJL : 0u) |
                                                      (q == j - 1 ? PG_BP_ADJR : 0u);
                                const int sl_ = LINES ? TM.lineL[eLi + k1] : 255, sr_ = LINES ? TM.lineR[eRi + k2] : 255;
                                const bool wrap = k2 + 1 == nr;
                                k2 = wrap ? 0 : k2 + 1;
                                k1 = wrap ? k1 + 1 : k1;
                                e1 = *(const i2 *)TM.eL[eLi + k1];
                                e2 = *(const i2 *)TM.eR[eRi + k2];
                                double xs, ys, ms;
                                fetch(p, q, sl_, sr_, xs, ys, ms);
                                PG_TAKE(bm, pm, ((ms + tM) + lw) + rw, base | PG_M);
                                PG_TAKE(bm, pm, ((xs + tX) + lw) + rw, base | PG_X);
                                PG_TAKE(bm, pm, ((ys + tX) + lw) + rw, base | PG_Y);
                            }
                        }
                    };
                    if (active) loops(std::integral_constant<bool, TAB>());
#ifdef PG_TILE_STATS
                    {   // fetches from beyond the LDS window in this step: the wave's largest count, and whether there was any
                        int f_ = (int)(st_far - far_before);
                        for (int o_ = 32; o_ > 0; o_ >>= 1) f_ = max(f_, __shfl_xor(f_, o_));
                        st_far_max += f_; st_far_steps += f_ > 0;
                    }
#endif
                }
            }
            if (active) {
                const int at = TAT(i, j);
                TM.sc[at][0] = bx; TM.sc[at][1] = by; TM.sc[at][2] = bm;
                {   // 24 B of scores (the 12 B of back-pointers are pg_backptr's)
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    gdouble_w o_ = J.sc + 3 * (off + (i - mn));
                    d2 xy; xy.x = bx; xy.y = by;
                    *(PG_GLOBAL d2 *)o_ = xy;
                    o_[2] = bm;
                }
                (void)px; (void)py; (void)pm;
            }
        }
        D = Dn; c = cn; ce = cen;
        asm volatile("" ::: "memory");
        // (the first progress anybody waits for is TS + TB - 1)
        if (LAG && (s & (TB - 1)) == TB - 1 && s >= TS && s < s_last) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // (waits for the wave's stores, writes the L2 back)
            if (r == 0) __hip_atomic_store(&prog[self], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef PG_TILE_STATS
        if (st_kind >= 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - st_a;
            if (st_kind == 0) { ++st_n[0]; st_t[0] += dt; } else if (st_kind == 1) { ++st_n[1]; st_t[1] += dt; } else { ++st_n[2]; st_t[2] += dt; }
        }
#endif
    }
#ifdef PG_TILE_STATS
    if (r == 0 && 3 * (J.Lx + J.Ly) >= 4096) {
        unsigned long long *out = (unsigned long long *)(jobs[T.x].trace + ((3 * (J.Lx + J.Ly) - 64) & ~1));
        atomicAdd(out + 0, 1ull);
        atomicAdd(out + 1, st_loop - st_begin);
        for (int k = 0; k < 3; ++k) { atomicAdd(out + 2 + k, st_n[k]); atomicAdd(out + 5 + k, st_t[k]); }
        // [15] generic steps with a fetch from beyond the LDS window, [16] such fetches of the busiest lane summed over those steps,
        // [17] (left, right) edge pairs of the busiest lane summed over the generic steps
        atomicAdd(out + 15, (unsigned long long)st_far_steps); atomicAdd(out + 16, (unsigned long long)st_far_max); atomicAdd(out + 17, (unsigned long long)st_pairs_max);
        atomicAdd(out + 8, __builtin_amdgcn_s_memtime() - st_begin);
    }
#endif
}

// tiles[blockIdx.x] = {job, tile row a, tile column b, -}: one launch per tile anti-diagonal (PAGAN_DP_TILES=launches)
__global__ __launch_bounds__(64) void pg_fill_tiles(const PgDevJob *__restrict__ jobs, const int *__restrict__ tiles,
                                                    unsigned flags) {
    const pg_i4 T = ((cdesc_p)tiles)[blockIdx.x];
    if (jobs[T.x].S <= 16) tile_body<false, TBLK, true>(jobs, T, flags); else tile_body<false, TBLK, false>(jobs, T, flags);
}

// ---- dataflow schedule: ONE launch for all tile anti-diagonals of the batch ----
// A launch per anti-diagonal leaves compute units idle whenever a diagonal's tile count is not a multiple of their number
// (one tile per CU: the tile fills LDS) and at every diagonal's tail: the root of a 512-leaf tree ran at 42 % of the
// tile time x tiles / CUs bound.  Here one persistent wave per CU takes tiles from a queue in anti-diagonal order
// (`flow[0]`); tile (a,b) of diagonal t starts once its neighbours (a-1,b), (a,b-1) and (a-1,b-1) are done (`done[]`, list
// positions from the host; -1: not in the band).  Where the tiles of every job form a staircase -- each tile row a
// contiguous run of columns, first and last column never falling, consecutive rows touching (the host checks; any band
// the anchors make does) -- that is enough: by induction every tile (a',b') <= (a,b) is done then, i.e. every tile a bwd
// edge of this tile's cells can reach.  Otherwise (`use_water` 1) a tile also waits until every tile of the diagonals <= t-2
// is done (`fin[t']` counts finished tiles; a wave keeps a watermark).  `use_water` 0: staircase jobs, tiles start 80 steps
// behind their neighbours (tile_body<true>); 2: staircase jobs, a tile waits for its neighbours to finish -- the host picks
// it when the batch has more than 1.75 x as many tiles per anti-diagonal as the chip has compute units (the lag buys
// nothing then and its progress flags and block-wise halo cost 15 %).
// A wave only ever waits for tiles that were handed out before its own -- to waves that are running -- so the queue drains.
// Visibility across the XCDs' L2s: the finishing wave's stores are released at agent scope before its flags are set, the
// starting wave acquires at agent scope after it has seen them (the compiler's gfx950 memory model: write-back of the
// L2 / invalidate); the flags themselves are agent-scope atomics.
// tiles: 4 * (n_tiles + 1) ints of {job, a, b, list position of (a-1,b) or -1}, then n_tiles list positions of (a,b-1) or
// -1, the same for (a-1,b-1), then n_diag + 1 first-tile offsets per diagonal.  flow (zeroed by the host before every
// launch): [0] next tile, [1 .. n_diag] finished tiles per diagonal, then n_tiles progress words (steps published; TDONE:
// finished), then the give-up flag (flow_wait).
__global__ __launch_bounds__(64) void pg_fill_tiles_flow(const PgDevJob *__restrict__ jobs, const int *__restrict__ tiles,
                                                         int n_tiles, int n_diag, int *__restrict__ flow, unsigned flags,
                                                         int use_water) {
    const int *left = tiles + 4 * ((size_t)n_tiles + 1);
    const int *diag = left + n_tiles;
    const int *first = diag + n_tiles;
    int *fin = flow + 1, *done = flow + 1 + n_diag;
    int water = 0;                                             // every diagonal < water is complete
    if (flags & 0x200u) {                                      // diagnostic (PAGAN_DP_DEBUG_FLAGS=0x200): the waves of XCD 0 only -- do the
        unsigned x;                                            // reads of other tiles' cells cost what they cost because they cross XCDs?
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        if ((x & 15u) != 0) return;
    }
    int *giveup = done + n_tiles;                              // set by a wave whose wait ran into its limit: everybody leaves
    for (;;) {
        int idx = 0;
        if (threadIdx.x == 0) idx = __hip_atomic_fetch_add(&flow[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= n_tiles) break;
        const pg_i4 T = ((cdesc_p)tiles)[idx];
        const int t = T.y + T.z;
        auto wait_ge = [&](int *p, int need) { flow_wait(p, need, giveup, jobs[T.x].fill_status); };
        if (__hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
#ifdef PG_TILE_STATS
        const unsigned long long fs0 = __builtin_amdgcn_s_memtime();
#endif
        if (use_water == 1) for (; water <= t - 2; ++water) wait_ge(&fin[water], first[water + 1] - first[water]);
        const int lf = left[idx], dg = diag[idx];
#ifdef PG_TILE_STATS
        const unsigned long long fs1 = __builtin_amdgcn_s_memtime();
#endif
        const bool lag = use_water == 0 || use_water == 3;
        if (!lag) {
            if (T.w >= 0) wait_ge(&done[T.w], TDONE);
            if (lf >= 0) wait_ge(&done[lf], TDONE);
            if (use_water == 2 && dg >= 0) wait_ge(&done[dg], TDONE);
        } else if (T.w < 0 && lf < 0 && dg >= 0) wait_ge(&done[dg], TDONE);    // (the band enters through the corner)
#ifdef PG_TILE_STATS
        const unsigned long long fs2 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const bool small_table = jobs[T.x].S <= 16;
        if (!lag) { if (small_table) tile_body<false, TBLK, true>(jobs, T, flags); else tile_body<false, TBLK, false>(jobs, T, flags); }
        else if (use_water == 3) {
            if (small_table) tile_body<true, 8, true>(jobs, T, flags, done, idx, T.w, lf, giveup);
            else tile_body<true, 8, false>(jobs, T, flags, done, idx, T.w, lf, giveup);
        } else {
            if (small_table) tile_body<true, TBLK, true>(jobs, T, flags, done, idx, T.w, lf, giveup);
            else tile_body<true, TBLK, false>(jobs, T, flags, done, idx, T.w, lf, giveup);
        }
#ifdef PG_TILE_STATS
        const unsigned long long fs3 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // (waits for the wave's stores, writes the L2 back)
#ifdef PG_TILE_STATS
        if (threadIdx.x == 0) {      // [9] waiting for whole diagonals, [10] for the neighbours, [11] acquire + tile, [12] release
            const View Jv = load_view(jobs + T.x);
            if (3 * (Jv.Lx + Jv.Ly) >= 4096) {
            unsigned long long *out = (unsigned long long *)(jobs[T.x].trace + ((3 * (Jv.Lx + Jv.Ly) - 64) & ~1));
            atomicAdd(out + 9, fs1 - fs0); atomicAdd(out + 10, fs2 - fs1); atomicAdd(out + 11, fs3 - fs2);
            atomicAdd(out + 12, __builtin_amdgcn_s_memtime() - fs3);
            }
        }
#endif
        if (threadIdx.x == 0) {
            __hip_atomic_store(&done[idx], TDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&fin[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

