// dp_pipe.hip -- banded fill as a register wavefront, pipelined across waves (gfx950).
//
// Same recurrence and bit-exactness rules as dp_kernels.hip (reference:
// Viterbi_alignment::compute_fwd_scores, src/main/viterbi_alignment.cpp:856-971, with
// iterate_bwd_edges_for_gap VA:1328-1349 and iterate_bwd_edges_for_match VA:1353-1436).
// What changes is how one anti-diagonal hands its scores to the next.  The fill is a chain of
// ~2e5 dependent diagonals of 25-250 cells; its speed is the latency of ONE diagonal step, and
// a lone wave issues one instruction per ~4-5 cycles, so that latency is the number of
// instructions a wave executes per step.  The ring kernel (dp_kernels.hip) pays an LDS write ->
// s_barrier -> LDS read round trip per step on top; here the hand-over stays in registers:
//
//   - lane T of the PNT = 256 compute lanes owns the rows i with i % 256 == T, one row at a
//     time, and keeps its row's scores of the previous diagonal in VGPRs.  Cell (i,j) needs
//     (i,j-1): the lane's own registers; (i-1,j): lane T-1's registers, one DPP wave shift;
//     (i-1,j-1): what that shift delivered one step earlier.  No LDS read, no barrier.
//   - lane 0 of a wave takes row i-1 from lane 63 of the wave "upstream" through an LDS ring
//     sc[d % PRK][row % 256][X,Y,M] that every awake lane also writes each step (-inf outside
//     the band), and that multi-edge sites read their older predecessors from;
//   - there is NO workgroup barrier.  Each wave publishes the last diagonal it completed in
//     LDS; a wave with cells on diagonal d waits (cached flag, re-read only when stale) until
//     its upstream neighbour has completed d-1, and before a wave overwrites a ring row it
//     waits until its downstream neighbour has completed the last diagonal that still reads
//     that row (worked out by the host per diagonal: 18 steps of slack in simple stretches, 2
//     where long edges are about).  The band is a diagonal stripe, so the waves form a pipeline: the wave
//     holding the top of the band leads, the others follow a step behind;
//   - a wave whose 64 rows are nowhere near the band SLEEPS: the host lists, per wave, the
//     diagonal intervals in which it must be awake (rows within reach of the band, plus PRK
//     steps afterwards so that its ring columns are flushed to -inf).  A sleeping wave publishes
//     "completed everything up to my next wake-up" and costs nothing; with a 25-70 cell band
//     one or two waves carry the step and nobody waits for idle ones;
//   - the host classifies every diagonal (dp_abi.hip, classify_diagonals): 0 = all cells
//     simple (straight-line code), 1 = multi-edge sites whose predecessors are all in the
//     ring (straight-line code for the simple lanes, straight-line blocks for cells with at most two bwd
//     edges per site -- half size when no cell of the wave has two multi-edge sites --, an item loop for
//     the others), 2 = same
//     with edges reaching past the ring (those cells come from L2; every wave keeps all but
//     its last 24 stores retired, so a diagonal 8 steps behind every wave has landed), 3 =
//     general (first/last rows and columns, the steps after a wide
//     diagonal: all awake waves rendezvous and drain), 4 = wider than the lanes (every lane takes
//     several rows, cells come from L2), 5 = wider than the record windows (graph arrays from HBM too);
//   - a loader wave stages 16-byte site records (state, flags, first two bwd edges), the bwd
//     edge lists and the recent diagonal descriptors in LDS windows ahead of the slowest wave;
//   - scores and back-pointers stream to HBM with stores nobody waits for; compute waves
//     issue no vector-memory load outside class 2-4 diagonals.
#include "dp_kcommon.h"
#include <type_traits>

#define PNW 4
#define PNT (64 * PNW)
#define PNTW PG_PIPE_WIDTH       // widest diagonal handled in registers
#define PRK PG_PIPE_RING         // ring depth in diagonals
#define PAGE PG_PIPE_REACH       // a reader may reach PAGE-1 diagonals back
#define PRW 512                  // site-record window (sites)
#define PEC PG_PIPE_EDGE_CAP     // bwd-edge window (edges)
#define PDR 128                  // descriptor window (diagonals)
#define PDR_REACH 60             // oldest diagonal looked up in it
#define PLOOK 64                 // diagonals the loader looks ahead of the slowest wave
#define PLAND 8                  // stores of diagonal d have landed once the storing wave completed d+PLAND
#define PNA PG_PIPE_ASSIST       // assist waves
#define PST PG_PIPE_STAGE        // staging slots
#define PBLOCK (PNT + 64 * PNA + 64)     // compute waves, assist waves, loader wave
static_assert(PNTW == PNT - PAGE && PRK >= PAGE && PG_PIPE_WINDOW + 80 <= PRW, "kernel geometry out of step with dp_device.h");
static_assert(PNA == 3 && PST == 3 && PAGE >= PLAND + 4, "assist wave a stages the diagonals d % 3 == a into slot a, at most two diagonals ahead; far cells have landed");
#define PFAR_POOL ((PNT * 4) / 24)  // cells of 24 bytes in one staging slot of spm
// staged back-pointer words: the regular bits of a back-pointer plus what the compute wave needs to merge
#define PS_ONLY 0x80000000u      // the site has no edge from the previous site: the staged value IS the state's value
#define PS_FIRST 0x40000000u     // the staged winner precedes the previous-site edge in the list: it wins a tie
#define PS_BP 0x3fffffffu
#ifndef PG_SPIN_LIMIT_LOG2
#define PG_SPIN_LIMIT_LOG2 25
#endif
#define PSPIN_LIMIT (1 << PG_SPIN_LIMIT_LOG2)     // seconds of polling: far beyond any legitimate wait (a wave sleeping through a long gap)

// site record, word x
#define PR_SIMPLE 0x10000        // one bwd edge, from the previous site, log-weight 0
#define PR_NE_SHIFT 17           // 7 bits: number of bwd edges (host keeps sites with > 126 off this kernel)
// bits 24-31 of a site record's word x (small model tables only; the loader sets them, the hand-scheduled loop reads them)
#define PR_SRC 0x1000000         // the site is the START of an edge that reads a far history line: its lane appends its cell to line (x >> 25) & 3
#define PR_THREE 0x20000000       // (with PR_TWO) THREE bwd edges, one from the previous site, all inside the ring's reach: the lanes evaluate the third
                                 // one -- entry 2 of the site's list in the edge window, which the loader permutes to (previous-site edge, other, other) --
                                 // in a third pass of their class 1 blocks on the diagonals the host flags (descriptor word 4, bit 19)
#define PR_TWO 0x40000000        // two bwd edges, exactly one of them from the previous site -- NORMALISED: that edge in slot 0, the other one in
                                 // slot 1 whatever the list's order (the fill computes values only: tools/gen_hot_asm.py, c1_issue)
#define PR_FAR 0x80000000u       // ... and the other edge reaches past the ring: its operands come from history line (x >> 27) & 3.  The loop tests
                                 // "other edge in the ring" as ONE signed compare, x > 0x3fffffff: PR_TWO set, PR_FAR (the sign) clear, whatever the
                                 // bits below say

struct PipeSmem {
    double sc[PRK][PNT][3];      // X, Y, M
    // what the assist waves hand over, per staging slot (d % PST) and lane of the compute waves (row % 256):
    // best candidate of X / Y over the bwd edges that do NOT come from the previous site, M over all edge pairs.
    // (Directly behind the ring: a wide run uses ring + staging arrays as ONE wide ring -- the assist waves stage nothing while
    //  the compute waves are inside a wide run: every diagonal they could stage lies behind it and its general steps.)
    double sx[PST][PNT], sy[PST][PNT], sM[PST][PNT];
    unsigned spx[PST][PNT], spy[PST][PNT], spm[PST][PNT];
    pg_i4 recL[PRW], recR[PRW];  // x: state | PR_SIMPLE | n_edges << 17 | PR_TWO, y: dist0 | dist1 << 16, z/w: log-weights 0/1
    pg_i4 dring[PDR];            // lo, hi, score byte offset (64 bit) of diagonal d at [d % PDR]
    int ebL[PRW], ebR[PRW];      // first bwd edge of the site (edge numbering of the graph)
    int esL[PEC], esR[PEC];
    float ewL[PEC], ewR[PEC];
    // small model tables (S*S <= 256, TAB_LDS): the match terms of every state pair, ready to add --
    // tab2[a + b*S] = { D(2*ng) + D(s(a,b)), D(0+ng) + D(s(a,b)) } (VA:1363-1367), computed once per workgroup with the expressions every
    // other path uses; large tables: the model scores the assist waves gather per staging slot
    union { double tab2[256][2]; float ssm[PST][PNT]; };
    // Far histories (round 5; dp_abi.hip, plan_far_hist): line q holds the last 64 cells of ONE row (indexed by column % 64)
    // or ONE column (indexed by row % 64) that is the start site of an edge reaching past the ring -- written every step by
    // the lane that computes the cell, read k steps later by the lane of the edge's end site, in place of a trip to L2.
    // Who reads and who writes which line is in the site records (PR_FAR / PR_SRC, set by the loader from the host's flag bytes).
    double hist[PG_HIST_SLOTS][64][3];
    double null_cell[4];         // -inf, -inf, -inf: what a missing second edge reads
    int progress[PNW];           // last diagonal each compute wave completed (or sleeps through)
    int arrived[PNW];            // last rendezvous diagonal each compute wave drained for
    int wflag[8];                // seven-wave wide runs (wide_run7): last diagonal each of the 4 compute + 3 assist waves completed
    int warrived[8];             // ... and the run boundary (first diagonal on entry, the one behind the run on exit) each has drained for
    int loaded[3];               // rows / columns / diagonal descriptors staged by the loader
    int abort_flag;
    int assist_done[PNA];        // last diagonal each assist wave has staged
    int as_list[PNA][64];        // rows of the multi-edge cells of the diagonal an assist wave is working on, compacted
    const int *pdsc;             // row strips: the parent job's dsc array (null otherwise): far_ask's view of the whole band
#ifdef PG_PIPE_STATS
    long long far_limit;         // (statistics builds) bytes of the job's score array: a far read outside it sets the abort flag instead
#endif
};

static_assert(sizeof(PipeSmem) <= 160 * 1024, "PipeSmem has to fit the 160 KB of LDS of a gfx950 compute unit");
unsigned pg_pipe_lds_bytes() { return (unsigned)sizeof(PipeSmem); }
unsigned pg_pipe_block() { return PBLOCK; }

// A STATIC allocation (gfx950 takes 160 KB; the launch asks for no dynamic LDS): the kernel's run functions and the assist
// waves are functions of their own, and in a non-kernel function every access to DYNAMIC LDS first loads the allocation's base
// address from a table in memory -- a scalar load and a drained LDS queue per access (142 sites in pipe_assist alone).
__shared__ __attribute__((aligned(16))) PipeSmem pg_pipe_smem;
#define PM pg_pipe_smem

namespace {

typedef int pg_i8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const pg_i8 *cdesc8_p;
typedef __attribute__((address_space(4))) const int *cint_p;

// Acquire, so that the compiler keeps the LDS reads a flag guards (ring cells, records, descriptors) below the read
// of the flag; in hardware a wave's LDS operations execute in order anyway.  The stores sit behind a release fence
// restricted to LDS: an unrestricted release would also wait for the wave's HBM stores, which the flags do not
// cover (the far reads have their own rule).
__device__ __forceinline__ int flag_peek(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int flag_load(const int *p) { return __builtin_amdgcn_readfirstlane(flag_peek(p)); }
__device__ __forceinline__ void flag_store(int *p, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // LDS only: the writes a flag publishes stay above its store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The hot loop's flag store: the writes a flag publishes are this wave's own LDS writes, which the hardware executes
// in order; the inline asm keeps the compiler from moving them across and from waiting for them (the fence above ends
// up as an s_waitcnt on every LDS operation in flight).
__device__ __forceinline__ void flag_store_inorder(int *p, int v) {
    const unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) int *)p;
    asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory");
}

// Spin until *p >= need.  Every wait in this kernel is for a wave that is not waiting for the
// caller (see the header); the spin limit only turns a logic error into an error status
// instead of a hung GPU.
// `tag` (nonzero) names the wait in the error status: kind | wave << 4 | diagonal << 8.
__device__ __forceinline__ int poll_ge(const int *p, int need, int tag) {
    int v = flag_load(p);
    int spins = 0;
#pragma nounroll
    while (v < need) {
        if (spins < 64) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
        if (++spins > PSPIN_LIMIT || flag_load(&PM.abort_flag) != 0) {
            if (flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, tag);
            return 0x7ffffff0;
        }
        v = flag_load(p);
    }
    return v;
}
// the same as a function of its own: the run functions call it only when a cached flag stops covering a step
__device__ __noinline__ int poll_ge_out(const int *p, int need, int tag) { return poll_ge(p, need, tag); }
#define PTAG(kind) ((kind) | (wave << 4) | (d << 8))
// (statistics builds) the strips of a job share its trace buffer: only the strip named in the debug flags' bits 12-15 records
#define PG_STATS_MINE(job, flags) (!(job)->is_strip || (job)->strip_row0 / PG_STRIP_ROWS == (int)(((flags) >> 12) & 15u))
#ifdef PG_PIPE_STATS
#define POLLX(p, need, kind) ([&] { const long long t0_ = __builtin_readcyclecounter(); const int v_ = __builtin_amdgcn_readfirstlane(poll_ge_out(p, need, PTAG(kind))); st_poll_t[kind] += __builtin_readcyclecounter() - t0_; ++st_poll_n[kind]; return v_; }())
#define POLL(p, need, kind) ([&] { const long long t0_ = __builtin_readcyclecounter(); const int v_ = poll_ge(p, need, PTAG(kind)); st_poll_t[kind] += __builtin_readcyclecounter() - t0_; ++st_poll_n[kind]; return v_; }())
#else
#define POLLX(p, need, kind) __builtin_amdgcn_readfirstlane(poll_ge_out(p, need, PTAG(kind)))
#define POLL(p, need, kind) poll_ge(p, need, PTAG(kind))
#endif

__device__ __forceinline__ double dpp_shr1(double v, double lane0) {
    // lane n takes lane n-1's v; lane 0 keeps `lane0` (wave_shr:1 leaves the destination of a lane
    // without a source untouched when bound_ctrl is off)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- loader wave -------------------------------------------------------------------------
template <bool LEFT>
__device__ __forceinline__ void load_rec_chunk(int first, int lane, int n, gint_p st, gint_p off, gint_p src, gfloat_p lw, bool strip, bool norm,
                                               PG_GLOBAL const unsigned char *hf) {
    pg_i4 *rec = LEFT ? PM.recL : PM.recR;
    int *eb = LEFT ? PM.ebL : PM.ebR;
    int *es = LEFT ? PM.esL : PM.esR;
    float *ew = LEFT ? PM.ewL : PM.ewR;
    const int r = first + lane;
    // two round trips for the usual chunk: offsets and states, then the sites' first two edges together with the chunk's
    // edge list for the LDS window; a third (and on) only where a site has more than two edges
    int b = 0, en = 0, w = 0;
    unsigned flag = 0;                                             // far histories: the site's flag byte (0 where the job has none)
    if (r < n) { b = off[r]; en = off[r + 1]; w = st[r] & 0xffff; if (hf) flag = hf[r]; }
    const int ne = en - b;
    const int rend = first + 64 < n ? first + 64 : n;
    const int e0 = __builtin_amdgcn_readfirstlane(b), e1 = __builtin_amdgcn_readlane(en, rend - 1 - first);
    int d0 = 0, d1 = 0;
    float w0 = 0.0f, w1 = 0.0f;
    if (ne >= 1) { d0 = r - src[b]; w0 = lw[b]; }
    if (ne >= 2) { d1 = r - src[b + 1]; w1 = lw[b + 1]; }
    for (int e = e0 + lane; e < e1; e += 128) {
        const bool two = e + 64 < e1;
        const int s0 = src[e];
        const float x0 = lw[e];
        int s1 = 0;
        float x1 = 0.0f;
        if (two) { s1 = src[e + 64]; x1 = lw[e + 64]; }
        es[e & (PEC - 1)] = s0; ew[e & (PEC - 1)] = x0;
        if (two) { es[(e + 64) & (PEC - 1)] = s1; ew[(e + 64) & (PEC - 1)] = x1; }
    }
    if (r < n) {
        if (r > 0 && ne == 1 && d0 == 1 && w0 == 0.0f) w |= PR_SIMPLE;
        w |= (ne < 127 ? ne : 127) << PR_NE_SHIFT;
        if (norm && ne == 2 && (d0 == 1) != (d1 == 1)) {
            if (d0 != 1) { const int td = d0; d0 = d1; d1 = td; const float tw = w0; w0 = w1; w1 = tw; }
            w |= PR_TWO;
        }
        if (norm && ne == 3) {
            // three edges, exactly one from the previous site, none past the ring's reach (dp_abi.hip, SiteFeat::is_three -- the same
            // test): the site's list, in the record AND in the edge window, becomes (previous-site edge, other, other); values do not
            // depend on a list's order, and every reader of the window sees the same permuted list (this lane's LDS writes follow
            // the chunk's in the wave's order)
            int d2 = r - src[b + 2];
            float w2 = lw[b + 2];
            const int n_adj = (d0 == 1) + (d1 == 1) + (d2 == 1);
            const int far_ = d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2);
            if (n_adj == 1 && far_ <= PAGE - 2) {
                if (d1 == 1) { const int td = d0; d0 = d1; d1 = td; const float tw = w0; w0 = w1; w1 = tw; }
                else if (d2 == 1) { const int td = d0; d0 = d2; d2 = td; const float tw = w0; w0 = w2; w2 = tw; }
                es[b & (PEC - 1)] = r - d0; ew[b & (PEC - 1)] = w0;
                es[(b + 1) & (PEC - 1)] = r - d1; ew[(b + 1) & (PEC - 1)] = w1;
                es[(b + 2) & (PEC - 1)] = r - d2; ew[(b + 2) & (PEC - 1)] = w2;
                w |= PR_TWO | PR_THREE;
            }
        }
        pg_i4 v;
        v.x = w; v.y = (d0 < 65535 ? d0 : 65535) | ((d1 < 65535 ? d1 : 65535) << 16);
        v.z = __float_as_int(w0); v.w = __float_as_int(w1);
        // row strips: the first site (no bwd edge) is recorded as a SIMPLE one -- one edge of weight 1 from a site before it.
        // Whatever a cell of row 0 (column 0) reads of the row (column) before it is -inf (a ring column no wave writes, a lane
        // outside the band, a row outside the band for far_ask), which is what the general rules give its X and M (Y and M);
        // its y-gap (x-gap) chain is the straight code's at the terminal rate; the diagonals 0 and 1 and the cells that meet
        // M(0,0) through an edge from site 0 are general steps (dp_abi.hip, plan_strips), and those read no record of site 0.
        if (strip && r == 0) { v.x = (w & 0xffff) | PR_SIMPLE | (1 << PR_NE_SHIFT); v.y = 1; v.z = 0; v.w = 0; }
        // far histories (dp_abi.hip, plan_far_hist; flag byte: bit 7 reader + bits 0-1 its line, bit 6 writer + bits 4-5 its line)
        if (flag & 0x80u) v.x |= (int)(PR_FAR | ((flag & 3u) << 27));
        if (flag & 0x40u) v.x |= (int)(PR_SRC | (((flag >> 4) & 3u) << 25));
        rec[r & (PRW - 1)] = v;
        eb[r & (PRW - 1)] = b;
    }
}

// The site-record windows are filled as far ahead as they allow -- a record may replace the one PRW sites before it once
// the slowest wave's diagonal has left that site behind (a band's first row and first column never fall) -- and at least as
// far as the diagonal PLOOK ahead needs; the descriptor window follows PLOOK diagonals ahead.  One chunk of rows and one of
// columns per round, each published as it lands.
#define PREC_BACK 72              // 64 (a chunk) + 8: the records up to 8 sites before the slowest diagonal's first stay
// follow[0] = 1 + the last diagonal whose scores have landed in L2 (PgDevJob::follow; what the follower workgroups wait
// for): a wave's stores of diagonal d have landed once it completed d + PLAND; published every 16 diagonals or so
// (wt, row strips: written through -- the strip below may poll from another XCD)
__device__ __forceinline__ void publish_landed(PG_GLOBAL int *follow, int lane, int landed, bool wt = false) {
    if (follow && lane == 0) {
        if (wt) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(follow), "v"(landed + 1) : "memory");
        else asm volatile("global_store_dword %0, %1, off" :: "v"(follow), "v"(landed + 1) : "memory");
    }
}
// Row strips (`strip`: the PgDevJob, null otherwise): records from the strip's first halo row / first column on, the
// descriptor window from the PARENT's array (whole-band rows per diagonal, offsets in cells there), everything from d_first.
__device__ __forceinline__ void pipe_loader(const View &J, cdesc8_p psc, int lane, PG_GLOBAL int *follow, const PgDevJob *strip, bool norm,
                                            PG_GLOBAL const unsigned char *hfl, PG_GLOBAL const unsigned char *hfr) {
    int rows = 0, cols = 0, diags = 0, published = -1;
    PG_GLOBAL const pg_i4 *pdsc = nullptr;
    if (strip) {
        rows = strip->strip_row0 >= 64 ? strip->strip_row0 - 64 : 0;
        cols = strip->col_first;
        diags = strip->d_first >= 64 ? strip->d_first - 64 : 0;
        published = strip->d_first - 1;
        pdsc = (PG_GLOBAL const pg_i4 *)strip->pdsc;
    }
    for (;;) {
        int pmin = flag_load(&PM.progress[0]);
        for (int w = 1; w < PNW; ++w) { const int p = flag_load(&PM.progress[w]); pmin = p < pmin ? p : pmin; }
        if (flag_load(&PM.abort_flag) != 0) { publish_landed(follow, lane, J.nd - 1, strip != nullptr); return; }   // (the job fails: the followers need not wait)
        // A compute wave's flag for the LAST diagonal goes up as after any other step -- with its stores still in flight -- so
        // the tail is published like the middle, PLAND behind the slowest wave, and the last diagonals only once every wave
        // has said "drained for good": progress = nd, stored behind the s_waitcnt vmcnt(0) that ends its last interval.
        if (pmin >= J.nd) { publish_landed(follow, lane, J.nd - 1, strip != nullptr); return; }
        if (pmin - PLAND >= published + (strip ? 8 : PG_FOLLOW_CHUNK)) { published = pmin - PLAND; publish_landed(follow, lane, published, strip != nullptr); }
        if (pmin + 1 >= J.nd) { __builtin_amdgcn_s_sleep(8); continue; }                          // nothing left to stage
        const int dcur = pmin + 1;                                 // the slowest wave may be computing this one
        const int da = dcur + PLOOK < J.nd - 1 ? dcur + PLOOK : J.nd - 1;
        const pg_i8 ds = psc[da], dc = psc[dcur];
        int want_rows = ds.y + 5, want_cols = da - ds.x + 4;       // rows <= hi+4, columns <= jmax+3 of diagonal da
        if (dc.y >= dc.x) {
            // as far ahead as the windows allow -- and no further: a record may only replace the one PRW sites before it, and those
            // of the slowest wave's diagonal (from PREC_BACK - 64 sites before its first row / column on) are still read.  (Round 5:
            // the look-ahead of PLOOK diagonals used to win over this bound; with diagonals of up to PG_PIPE_WINDOW cells it must not.)
            const int far_rows = dc.x + PRW - PREC_BACK, far_cols = dcur - dc.y + PRW - PREC_BACK;
            want_rows = far_rows;
            want_cols = far_cols;
            if (dc.y - dc.x + 1 > PG_PIPE_WINDOW) {
                // a diagonal wider than the windows (class 5) takes its records from L2, but its waves still wait for "rows <= hi + 3
                // loaded": just those, so that the first diagonal the windows cover again finds the records before its first row
                const pg_i8 dn = psc[dcur + 1 < J.nd ? dcur + 1 : dcur];
                const int near_rows = dn.y + 5, near_cols = dcur + 1 - dn.x + 4;
                want_rows = want_rows > near_rows ? want_rows : near_rows;
                want_cols = want_cols > near_cols ? want_cols : near_cols;
            }
        }
        want_rows = want_rows < J.Lx ? want_rows : J.Lx;
        want_cols = want_cols < J.Ly ? want_cols : J.Ly;
        bool any = false;
        // descriptors of the diagonals up to da: what a far read needs to find an old cell
        // (entries older than dcur - PDR + PLOOK are overwritten; readers look back PDR_REACH at most)
        if (diags <= da) {
            while (diags <= da) {
                const int t = diags + lane;
                if (t <= da) {
                    pg_i4 v;
                    if (pdsc) {
                        v = pdsc[t];
                        const long long bo = 24ll * (((long long)v.w << 32) | (unsigned)v.z);
                        v.z = (int)(bo & 0xffffffffLL); v.w = (int)(bo >> 32);
                    } else v = *((PG_GLOBAL const pg_i4 *)psc + 2 * t);
                    PM.dring[t & (PDR - 1)] = v;
                }
                diags = diags + 64 < da + 1 ? diags + 64 : da + 1;
            }
            flag_store(&PM.loaded[2], diags);
            any = true;
        }
        if (rows < want_rows) {
            load_rec_chunk<true>(rows, lane, J.Lx, J.stL, J.offL, J.srcL, J.lwL, strip != nullptr, norm, hfl);
            rows += 64; any = true;
            flag_store(&PM.loaded[0], rows);
        }
        if (cols < want_cols) {
            load_rec_chunk<false>(cols, lane, J.Ly, J.stR, J.offR, J.srcR, J.lwR, strip != nullptr, norm, hfr);
            cols += 64; any = true;
            flag_store(&PM.loaded[1], cols);
        }
        if (!any) __builtin_amdgcn_s_sleep(8);
    }
}

// ---- compute-wave helpers ------------------------------------------------------------------
// bwd edge k of a site: distance back in sites and log-weight.  The first two come with the record.
template <bool LEFT>
__device__ __forceinline__ void edge_at(const pg_i4 &rec, int k, int site, int &dist, double &lw) {
    if (k == 0) { dist = rec.y & 0xffff; lw = (double)__int_as_float(rec.z); }
    else if (k == 1) { dist = (int)((unsigned)rec.y >> 16); lw = (double)__int_as_float(rec.w); }
    else {
        const int e = (LEFT ? PM.ebL : PM.ebR)[site & (PRW - 1)] + k;
        dist = site - (LEFT ? PM.esL : PM.esR)[e & (PEC - 1)];
        lw = (double)(LEFT ? PM.ewL : PM.ewR)[e & (PEC - 1)];
    }
}

// Cells from L2, several at once: every load AND the wait sit in ONE asm statement.  (Requesting in one statement and
// waiting in a later one leaves registers with a load in flight visible to the compiler, which may copy or spill them
// at a join or under pressure -- the copy reads the old content, and the load lands in a register that holds something
// else by then.  It showed as a memory fault that came and went with unrelated code.)  A lane that does not want one
// of the cells passes need = false: it loads the arena's first cell, which is always readable, and keeps what it had.
typedef double pg_d2 __attribute__((ext_vector_type(2)));
struct FarAsk { bool need; long long boff; };                     // byte offset of the cell in the job's score array
// (one cell: its loads under the lanes that want it; the other lanes keep what the registers hold)
#define PG_FAR_LD(xy, m_, a, mk) "s_and_b64 exec, %[sv], %[" #mk "]\n\tglobal_load_dwordx4 %[" #xy "], %[" #a "], off sc1\n\t" \
                                 "global_load_dwordx2 %[" #m_ "], %[" #a "], off offset:16 sc1\n\t"
__device__ __forceinline__ void far_fetch4(gdouble_w sc, const FarAsk &a0, const FarAsk &a1, const FarAsk &a2, const FarAsk &a3,
                                           pg_d2 &xy0, double &m0, pg_d2 &xy1, double &m1, pg_d2 &xy2, double &m2, pg_d2 &xy3, double &m3) {
#ifdef PG_PIPE_STATS
    {
        const long long lim = PM.far_limit;
        const bool bad = (a0.need && (a0.boff < 0 || a0.boff + 24 > lim)) || (a1.need && (a1.boff < 0 || a1.boff + 24 > lim)) ||
                         (a2.need && (a2.boff < 0 || a2.boff + 24 > lim)) || (a3.need && (a3.boff < 0 || a3.boff + 24 > lim));
        if (__builtin_amdgcn_ballot_w64(bad) != 0) { if (PM.abort_flag == 0) PM.abort_flag = 0x7f000004; return; }
    }
#endif
    const unsigned long long k0 = __builtin_amdgcn_ballot_w64(a0.need), k1 = __builtin_amdgcn_ballot_w64(a1.need);
    const unsigned long long k2 = __builtin_amdgcn_ballot_w64(a2.need), k3 = __builtin_amdgcn_ballot_w64(a3.need);
    if ((k0 | k1 | k2 | k3) == 0) return;
    PG_GLOBAL const char *b = (PG_GLOBAL const char *)sc;
    PG_GLOBAL const char *p0 = b + a0.boff, *p1 = b + a1.boff, *p2 = b + a2.boff, *p3 = b + a3.boff;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 PG_FAR_LD(x0, u0, p0, k0) PG_FAR_LD(x1, u1, p1, k1) PG_FAR_LD(x2, u2, p2, k2) PG_FAR_LD(x3, u3, p3, k3)
                 "s_mov_b64 exec, %[sv]\n\ts_waitcnt vmcnt(0)"
                 : [x0] "+v"(xy0), [u0] "+v"(m0), [x1] "+v"(xy1), [u1] "+v"(m1), [x2] "+v"(xy2), [u2] "+v"(m2), [x3] "+v"(xy3), [u3] "+v"(m3),
                   [sv] "=&s"(sv)
                 : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [k0] "s"(k0), [k1] "s"(k1), [k2] "s"(k2), [k3] "s"(k3)
                 : "memory");
}
__device__ __forceinline__ void far_fetch8(gdouble_w sc, const FarAsk (&a)[8], pg_d2 (&xy)[8], double (&m)[8]) {
#ifdef PG_PIPE_STATS
    {
        const long long lim = PM.far_limit;
        bool bad = false;
        for (int k = 0; k < 8; ++k) bad = bad || (a[k].need && (a[k].boff < 0 || a[k].boff + 24 > lim));
        if (__builtin_amdgcn_ballot_w64(bad) != 0) { if (PM.abort_flag == 0) PM.abort_flag = 0x7f000008; return; }
    }
#endif
    const unsigned long long k0 = __builtin_amdgcn_ballot_w64(a[0].need), k1 = __builtin_amdgcn_ballot_w64(a[1].need);
    const unsigned long long k2 = __builtin_amdgcn_ballot_w64(a[2].need), k3 = __builtin_amdgcn_ballot_w64(a[3].need);
    const unsigned long long k4 = __builtin_amdgcn_ballot_w64(a[4].need), k5 = __builtin_amdgcn_ballot_w64(a[5].need);
    const unsigned long long k6 = __builtin_amdgcn_ballot_w64(a[6].need), k7 = __builtin_amdgcn_ballot_w64(a[7].need);
    if ((k0 | k1 | k2 | k3 | k4 | k5 | k6 | k7) == 0) return;
    PG_GLOBAL const char *b = (PG_GLOBAL const char *)sc;
    PG_GLOBAL const char *p0 = b + a[0].boff, *p1 = b + a[1].boff, *p2 = b + a[2].boff, *p3 = b + a[3].boff;
    PG_GLOBAL const char *p4 = b + a[4].boff, *p5 = b + a[5].boff, *p6 = b + a[6].boff, *p7 = b + a[7].boff;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 PG_FAR_LD(x0, u0, p0, k0) PG_FAR_LD(x1, u1, p1, k1) PG_FAR_LD(x2, u2, p2, k2) PG_FAR_LD(x3, u3, p3, k3)
                 PG_FAR_LD(x4, u4, p4, k4) PG_FAR_LD(x5, u5, p5, k5) PG_FAR_LD(x6, u6, p6, k6) PG_FAR_LD(x7, u7, p7, k7)
                 "s_mov_b64 exec, %[sv]\n\ts_waitcnt vmcnt(0)"
                 : [x0] "+v"(xy[0]), [u0] "+v"(m[0]), [x1] "+v"(xy[1]), [u1] "+v"(m[1]), [x2] "+v"(xy[2]), [u2] "+v"(m[2]),
                   [x3] "+v"(xy[3]), [u3] "+v"(m[3]), [x4] "+v"(xy[4]), [u4] "+v"(m[4]), [x5] "+v"(xy[5]), [u5] "+v"(m[5]),
                   [x6] "+v"(xy[6]), [u6] "+v"(m[6]), [x7] "+v"(xy[7]), [u7] "+v"(m[7]), [sv] "=&s"(sv)
                 : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
                   [k0] "s"(k0), [k1] "s"(k1), [k2] "s"(k2), [k3] "s"(k3), [k4] "s"(k4), [k5] "s"(k5), [k6] "s"(k6), [k7] "s"(k7)
                 : "memory");
}
// the L2 address of cell (p, d - age), or "not wanted" when it lies outside the band (the caller keeps -inf)
__device__ __forceinline__ FarAsk far_ask(cdesc8_p psc, int d, int age, int p) {
    const int dd = d - age;
    // (an operand of a real cell lies on a diagonal >= 0; after an abort the waves run on without waiting for the loader, a
    //  site record may be another site's, and a distance from it may reach before the first diagonal: no descriptor is read for that)
    if (dd < 0) { FarAsk none = {false, 0}; return none; }
    pg_i4 ds;
    if (age <= PDR_REACH) ds = PM.dring[dd & (PDR - 1)];
    else if (const int *pd = PM.pdsc) {
        // a row strip: the strip's own descriptors hold its rows only; an operand may lie in a strip above (same arrays)
        ds = far_desc((PG_GLOBAL const pg_i4 *)pd + dd);
        const long long bo = 24ll * (((long long)ds.w << 32) | (unsigned)ds.z);
        ds.z = (int)(bo & 0xffffffffLL); ds.w = (int)(bo >> 32);
    } else ds = far_desc((PG_GLOBAL const pg_i4 *)psc + 2 * dd);
    FarAsk f;
    f.need = p >= ds.x && p <= ds.y;
    f.boff = (((long long)ds.w << 32) | (unsigned)ds.z) + 24ll * (p - ds.x);
    return f;
}

// Up to three earlier cells at once (the X, Y and M operands of one (left edge, right edge) item), each `age`
// diagonals back in row p.  FAR = false: every age is 1..PAGE-1 and those diagonals are in the ring (class 1).
// FAR = true: any earlier cell; `resmask` has bit a set when diagonal d-a went through the lanes, the others are
// requested from L2/HBM together -- after the caller made sure they have landed -- through the descriptor
// window (or the descriptor array itself) and waited for once.
struct CellAsk { bool need; int age, p; };
template <bool FAR>
__device__ __forceinline__ void old_cells3(gdouble_w sc, cdesc8_p psc, int d, int slot, unsigned resmask, const CellAsk &a0,
                                           const CellAsk &a1, const CellAsk &a2, double (&v)[3][3]) {
    const double NI = neg_inf();
    pg_d2 xy0 = {NI, NI}, xy1 = {NI, NI}, xy2 = {NI, NI};
    double m0 = NI, m1 = NI, m2 = NI;
    auto fetch = [&](const CellAsk &a, pg_d2 &xy, double &m) -> FarAsk {
        FarAsk f = {false, 0};
        if (!a.need) return f;
        if (!FAR || (a.age < PAGE && ((resmask >> a.age) & 1u))) {
            int s = slot - a.age;
            s += s < 0 ? PRK : 0;
            xy.x = PM.sc[s][a.p & (PNT - 1)][PG_X];
            xy.y = PM.sc[s][a.p & (PNT - 1)][PG_Y];
            m = PM.sc[s][a.p & (PNT - 1)][PG_M];
        } else {
            f = far_ask(psc, d, a.age, a.p);
        }
        return f;
    };
    const FarAsk f0 = fetch(a0, xy0, m0), f1 = fetch(a1, xy1, m1), f2 = fetch(a2, xy2, m2);
    if (FAR) {
        const FarAsk none = {false, 0};
        pg_d2 dq = {NI, NI};
        double dm = NI;
        far_fetch4(sc, f0, f1, f2, none, xy0, m0, xy1, m1, xy2, m2, dq, dm);
    }
    v[0][0] = xy0.x; v[0][1] = xy0.y; v[0][2] = m0;
    v[1][0] = xy1.x; v[1][1] = xy1.y; v[1][2] = m1;
    v[2][0] = xy2.x; v[2][1] = xy2.y; v[2][2] = m2;
}

// The part of a multi-edge cell (class 1/2: interior) that does not need the previous diagonal, evaluated by an ASSIST
// wave up to two diagonals ahead of the compute waves:
//   X: the candidates of every left bwd edge that does NOT start at the previous site (an edge from row-1 reads the
//      cell (row-1, j) of diagonal d-1: that one stays with the compute wave, whose straight-line code is exactly its
//      three candidates -- gap candidates ignore edge weights, VA:2116-2219);
//   Y: the same for the right site;
//   M: every (left edge, right edge) pair, row-major (VA:1396-1433): all of them read diagonals <= d-2.
// Candidates in list order, strict > (first wins).  The compute wave merges: X/Y by value, ties by list position
// (PS_FIRST: the staged winner's edge precedes the previous-site edge; PS_ONLY: there is no previous-site edge, the
// staged value is the state's value); M is taken as it is.  adjacent-edge slots travel in unused bits of the staged
// words so that the compute wave's back-pointer names the right list slot.  This is the general form (any number of
// edges, sites without edges); assist1_cell / assist2_cell are its straight-line special cases.
template <bool FAR>
__device__ __forceinline__ void assist_cell(gdouble_w sc, cdesc8_p psc, int d, int slot, unsigned resmask, const pg_i4 &rL,
                                            const pg_i4 &cR, int row, int j, bool reduced_terminal, double go, double gex, double gey,
                                            double ng, double tM, double tX, double &ex, double &ey, double &em,
                                            unsigned &px, unsigned &py, unsigned &pm) {
    const double NI = neg_inf();
    ex = NI; ey = NI; em = NI; px = PG_BP_NONE; py = PG_BP_NONE; pm = PG_BP_NONE;
    const int nL = (rL.x >> PR_NE_SHIFT) & 127, nR = (cR.x >> PR_NE_SHIFT) & 127;
    int adjL = -1, adjR = -1, winL = -1, winR = -1, dL, dR;
    double lw, rw;
    for (int k = 0; k < nL; ++k) { edge_at<true>(rL, k, row, dL, lw); if (dL == 1) adjL = k; }
    for (int k = 0; k < nR; ++k) { edge_at<false>(cR, k, j, dR, rw); if (dR == 1) adjR = k; }
    const int n_items = nL * nR;
    int k1 = 0, k2 = 0;
    if (n_items > 0) { edge_at<true>(rL, 0, row, dL, lw); edge_at<false>(cR, 0, j, dR, rw); }
    for (int t = 0; t < n_items; ++t) {
        double v[3][3], c;
        const bool do_x = k2 == 0 && dL != 1, do_y = k1 == 0 && dR != 1;
        const CellAsk ax = {do_x, dL, row - dL}, ay = {do_y, dR, row}, am = {true, dL + dR, row - dL};
        old_cells3<FAR>(sc, psc, d, slot, resmask, ax, ay, am, v);
        if (do_x) {                                                   // X candidates of left edge k1
            const double open = (reduced_terminal && row == dL) ? 0.0 : go;
            const unsigned w = pack_bp(0, k1, 0, false, false);
            c = v[0][0] + gex;            if (c > ex) { ex = c; px = w | PG_X; winL = k1; }
            c = (v[0][1] + 0.0) + go;    if (c > ex) { ex = c; px = w | PG_Y; winL = k1; }
            c = (v[0][2] + ng) + open;   if (c > ex) { ex = c; px = w | PG_M; winL = k1; }
        }
        if (do_y) {                                                   // Y candidates of right edge k2
            const double open = (reduced_terminal && j == dR) ? 0.0 : go;
            const unsigned w = pack_bp(0, 0, k2, false, false);
            c = v[1][1] + gey;            if (c > ey) { ey = c; py = w | PG_Y; winR = k2; }
            c = (v[1][0] + 0.0) + go;    if (c > ey) { ey = c; py = w | PG_X; winR = k2; }
            c = (v[1][2] + ng) + open;   if (c > ey) { ey = c; py = w | PG_M; winR = k2; }
        }
        {                                                             // M candidates of the pair
            const unsigned w = pack_bp(0, k1, k2, dL == 1, dR == 1);
            c = ((v[2][2] + tM) + lw) + rw;  if (c > em) { em = c; pm = w | PG_M; }
            c = ((v[2][0] + tX) + lw) + rw;  if (c > em) { em = c; pm = w | PG_X; }
            c = ((v[2][1] + tX) + lw) + rw;  if (c > em) { em = c; pm = w | PG_Y; }
        }
        if (++k2 == nR) { k2 = 0; ++k1; if (k1 < nL) edge_at<true>(rL, k1, row, dL, lw); }
        edge_at<false>(cR, k2, j, dR, rw);
    }
    // X: regular bits 0-17 (from, k1 << 4), the previous-site edge's slot in bits 18-24; Y: regular bits 0-3 and 18-24
    // (k2 << 18), the previous-site edge's slot in bits 4-10
    px |= (unsigned)(adjL < 0 ? 0 : adjL) << 18;
    py |= (unsigned)(adjR < 0 ? 0 : adjR) << 4;
    if (adjL < 0) px |= PS_ONLY; else if (winL >= 0 && winL < adjL) px |= PS_FIRST;
    if (adjR < 0) py |= PS_ONLY; else if (winR >= 0 && winR < adjR) py |= PS_FIRST;
}

// One candidate of a cell state: replaces the incumbent only if strictly greater (first wins).
__device__ __forceinline__ void cand(double c, unsigned f, double &best, unsigned &bp) {
    const bool gt = c > best;
    bp = gt ? f : bp;
    best = __builtin_fmax(best, c);
}

// Multi-edge cell whose sites have at most two bwd edges each (nearly all of them: a site after a gap has
// the edge from its predecessor and the one that skips the gap), every predecessor in the ring: the up
// to 2 + 2 + 4 cells are read at once and the up to 24 candidates evaluated in the reference's order
// (X by left edge, Y by right edge, M by (left, right) pair, row-major) as straight-line code -- one
// LDS latency and independent instruction streams instead of one dependent loop iteration per pair.
// A missing second edge reads the all -inf null cell, whose candidates can never win.
template <bool FAR>
__device__ __forceinline__ void multi2_cell(gdouble_w sc, cdesc8_p psc, int d, unsigned resmask, int slot, const pg_i4 &rL,
                                            const pg_i4 &cR, int row, int j, bool reduced_terminal,
                                            double go, double gex, double gey, double ng, double tM, double tX, double &bx, double &by,
                                            double &bm, unsigned &px, unsigned &py, unsigned &pm) {
    const double NI = neg_inf();
    const bool l1 = ((rL.x >> PR_NE_SHIFT) & 127) > 1, r1 = ((cR.x >> PR_NE_SHIFT) & 127) > 1;
    const int dL0 = rL.y & 0xffff, dL1 = (int)((unsigned)rL.y >> 16), dR0 = cR.y & 0xffff, dR1 = (int)((unsigned)cR.y >> 16);
    const double lw0 = (double)__int_as_float(rL.z), lw1 = (double)__int_as_float(rL.w);
    const double rw0 = (double)__int_as_float(cR.z), rw1 = (double)__int_as_float(cR.w);
    double xa_x, xa_y, xa_m, xb_x, xb_y, xb_m, ya_x, ya_y, ya_m, yb_x, yb_y, yb_m;
    double m00x, m00y, m00m, m01x, m01y, m01m, m10x, m10y, m10m, m11x, m11y, m11m;
    if (!FAR) {
        // every cell is in the ring: branch-free reads, a missing edge reads the null cell
        auto cell = [&](int age, int p, bool present, double &xs, double &ys, double &ms) {
            int s = slot - age;
            s += s < 0 ? PRK : 0;
            const double *c = present ? &PM.sc[s][p & (PNT - 1)][0] : &PM.null_cell[0];
            xs = c[PG_X]; ys = c[PG_Y]; ms = c[PG_M];
        };
        cell(dL0, row - dL0, true, xa_x, xa_y, xa_m);
        cell(dL1, row - dL1, l1, xb_x, xb_y, xb_m);
        cell(dR0, row, true, ya_x, ya_y, ya_m);
        cell(dR1, row, r1, yb_x, yb_y, yb_m);
        cell(dL0 + dR0, row - dL0, true, m00x, m00y, m00m);
        cell(dL0 + dR1, row - dL0, r1, m01x, m01y, m01m);
        cell(dL1 + dR0, row - dL1, l1, m10x, m10y, m10m);
        cell(dL1 + dR1, row - dL1, l1 && r1, m11x, m11y, m11m);
    } else {
        // some cells have left the ring: those are requested from L2 together and waited for once
        pg_d2 q[8];
        double qm[8];
        FarAsk fa[8];
        auto cell = [&](int k, int age, int p, bool present) {
            q[k].x = NI; q[k].y = NI; qm[k] = NI;
            fa[k].need = false; fa[k].boff = 0;
            if (!present) return;
            if (age < PAGE && ((resmask >> age) & 1u)) {
                int s = slot - age;
                s += s < 0 ? PRK : 0;
                q[k].x = PM.sc[s][p & (PNT - 1)][PG_X]; q[k].y = PM.sc[s][p & (PNT - 1)][PG_Y]; qm[k] = PM.sc[s][p & (PNT - 1)][PG_M];
            } else {
                fa[k] = far_ask(psc, d, age, p);
            }
        };
        cell(0, dL0, row - dL0, true);
        cell(1, dL1, row - dL1, l1);
        cell(2, dR0, row, true);
        cell(3, dR1, row, r1);
        cell(4, dL0 + dR0, row - dL0, true);
        cell(5, dL0 + dR1, row - dL0, r1);
        cell(6, dL1 + dR0, row - dL1, l1);
        cell(7, dL1 + dR1, row - dL1, l1 && r1);
        far_fetch8(sc, fa, q, qm);
        xa_x = q[0].x; xa_y = q[0].y; xa_m = qm[0];  xb_x = q[1].x; xb_y = q[1].y; xb_m = qm[1];
        ya_x = q[2].x; ya_y = q[2].y; ya_m = qm[2];  yb_x = q[3].x; yb_y = q[3].y; yb_m = qm[3];
        m00x = q[4].x; m00y = q[4].y; m00m = qm[4];  m01x = q[5].x; m01y = q[5].y; m01m = qm[5];
        m10x = q[6].x; m10y = q[6].y; m10m = qm[6];  m11x = q[7].x; m11y = q[7].y; m11m = qm[7];
    }
    const unsigned aL0 = dL0 == 1 ? PG_BP_ADJL : 0u, aL1 = dL1 == 1 ? PG_BP_ADJL : 0u;
    const unsigned aR0 = dR0 == 1 ? PG_BP_ADJR : 0u, aR1 = dR1 == 1 ? PG_BP_ADJR : 0u;
    bx = NI; by = NI; bm = NI; px = PG_BP_NONE; py = PG_BP_NONE; pm = PG_BP_NONE;
    {   // X: gap in the right sequence, candidates per left edge (VA:898-915)
        // (a class 1 diagonal lies PAGE rows and columns inside the matrix: no edge in reach starts at site 0)
        const double o0 = (FAR && reduced_terminal && row == dL0) ? 0.0 : go, o1 = (FAR && reduced_terminal && row == dL1) ? 0.0 : go;
        cand(xa_x + gex, aL0 | PG_X, bx, px); cand((xa_y + 0.0) + go, aL0 | PG_Y, bx, px); cand((xa_m + ng) + o0, aL0 | PG_M, bx, px);
        const unsigned w1 = aL1 | (1u << 4);
        cand(xb_x + gex, w1 | PG_X, bx, px); cand((xb_y + 0.0) + go, w1 | PG_Y, bx, px); cand((xb_m + ng) + o1, w1 | PG_M, bx, px);
    }
    {   // Y: gap in the left sequence, candidates per right edge (VA:927-944)
        const double o0 = (FAR && reduced_terminal && j == dR0) ? 0.0 : go, o1 = (FAR && reduced_terminal && j == dR1) ? 0.0 : go;
        cand(ya_y + gey, aR0 | PG_Y, by, py); cand((ya_x + 0.0) + go, aR0 | PG_X, by, py); cand((ya_m + ng) + o0, aR0 | PG_M, by, py);
        const unsigned w1 = aR1 | (1u << 18);
        cand(yb_y + gey, w1 | PG_Y, by, py); cand((yb_x + 0.0) + go, w1 | PG_X, by, py); cand((yb_m + ng) + o1, w1 | PG_M, by, py);
    }
    {   // M: (left edge, right edge) pairs row-major (VA:1396-1433)
        unsigned w = aL0 | aR0;
        cand(((m00m + tM) + lw0) + rw0, w | PG_M, bm, pm); cand(((m00x + tX) + lw0) + rw0, w | PG_X, bm, pm);
        cand(((m00y + tX) + lw0) + rw0, w | PG_Y, bm, pm);
        w = aL0 | aR1 | (1u << 18);
        cand(((m01m + tM) + lw0) + rw1, w | PG_M, bm, pm); cand(((m01x + tX) + lw0) + rw1, w | PG_X, bm, pm);
        cand(((m01y + tX) + lw0) + rw1, w | PG_Y, bm, pm);
        w = aL1 | aR0 | (1u << 4);
        cand(((m10m + tM) + lw1) + rw0, w | PG_M, bm, pm); cand(((m10x + tX) + lw1) + rw0, w | PG_X, bm, pm);
        cand(((m10y + tX) + lw1) + rw0, w | PG_Y, bm, pm);
        w = aL1 | aR1 | (1u << 4) | (1u << 18);
        cand(((m11m + tM) + lw1) + rw1, w | PG_M, bm, pm); cand(((m11x + tX) + lw1) + rw1, w | PG_X, bm, pm);
        cand(((m11y + tX) + lw1) + rw1, w | PG_Y, bm, pm);
    }
}

__device__ __forceinline__ double in_vgpr(double x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Commit of one step: EVERY lane writes its column of the ring row (-inf outside the band); cells
// inside the band go to HBM, 24 B of scores + 12 B of back-pointers at the diagonal's offset.
// A cell's three scores to the score matrix.  WT (row strips): written through to memory (sc1) -- the strip below, and a strip
// whose far operand this is, may run on another XCD, whose L2 is not this one's (strip_feeder).  Two instructions either way
// (the s_waitcnt vmcnt(n) behind a step counts on that).
template <bool WT>
__device__ __forceinline__ void store_scores(PG_GLOBAL char *p, double bx, double by, double bm) {
    pg_d2 xy; xy.x = bx; xy.y = by;
    if (WT) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %0, %2, off offset:16 sc1" :: "v"(p), "v"(xy), "v"(bm) : "memory");
    } else {
        *(PG_GLOBAL pg_d2 *)p = xy;
        *(PG_GLOBAL double *)(p + 16) = bm;
    }
}
template <bool WT>
__device__ __forceinline__ void commit_cell(gdouble_w sc_out, gu32_w bp_out, const pg_i8 &cur, int slot, int tid, int off,
                                            bool active, double bx, double by, double bm, unsigned px, unsigned py, unsigned pm) {
    PM.sc[slot][tid][PG_X] = bx;
    PM.sc[slot][tid][PG_Y] = by;
    PM.sc[slot][tid][PG_M] = bm;
    if (active) {
        typedef unsigned u3 __attribute__((ext_vector_type(3)));
        const long long soff = ((long long)cur.w << 32) | (unsigned)cur.z;     // 24 * first cell of the diagonal
        PG_GLOBAL char *srow = (PG_GLOBAL char *)sc_out + soff;
        PG_GLOBAL char *brow = (PG_GLOBAL char *)bp_out + (soff >> 1);
        store_scores<WT>(srow + 24u * (unsigned)off, bx, by, bm);
        u3 b3; b3.x = px; b3.y = py; b3.z = pm;
        *(PG_GLOBAL u3 *)(brow + 12u * (unsigned)off) = b3;
    }
    // all but this wave's last 24 stores (8 steps' worth) have retired: what a far read relies on
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}


// ---- helpers of the compute waves' own multi-edge arithmetic (hot_run) ----
// First-wins maximum of three candidates WITHOUT the "-inf means no back-pointer" rule (the caller applies it once per
// state): value = plain maximum, `from` = the first candidate equal to it.
__device__ __forceinline__ double fmax3_from(double c1, double c2, double c3, unsigned f1, unsigned f2, unsigned f3, unsigned &from) {
    const double m23 = __builtin_fmax(c2, c3);
    const double m = __builtin_fmax(c1, m23);
    const bool w3 = c3 > c2;
    const bool w23 = m23 > c1;
    const unsigned b = w3 ? f3 : f2;
    from = w23 ? b : f1;
    return m;
}
// (va, pa) against (vb, pb): b replaces a if strictly greater, or equal and listed first (first_is_bigger is strict,
// basic_alignment.h:449-462, so the earlier candidate keeps a tie)
__device__ __forceinline__ void take_better(double &va, unsigned &pa, double vb, unsigned pb, bool b_first) {
    const bool t = (vb > va) | ((vb == va) & b_first);
    va = t ? vb : va;
    pa = t ? pb : pa;
}
// the three scores of the ring cell at byte offset `off` of the ring (X, Y, M)
__device__ __forceinline__ void ring_cell(int off, double &x, double &y, double &m) {
    const double *c = (const double *)((const char *)&PM.sc[0][0][0] + off);
    x = c[PG_X]; y = c[PG_Y]; m = c[PG_M];
}
// wide diagonals (class 4, model table in LDS) reuse the ring's memory as PWK rows of PWPOS positions: a lane
// holds up to two rows of such a diagonal; a cell reads at most PWAGE diagonals back in it, older operands come from L2
// (round 5: 12 rows of 384 positions over the ring AND the assist waves' staging arrays behind it -- before, 7 rows of 512 in the
//  ring alone: a cell read five diagonals back at most, and nearly every wide step had an operand 6 .. 12 back that cost a trip to
//  L2.  A wide diagonal has at most PG_PIPE_WINDOW = 352 cells, and what a step reads of an earlier diagonal lies at most
//  2 * PWAGE rows above this diagonal's first row: 384 positions tell those rows apart.  Rows 384 and more past the first row --
//  a lane's second row can be -- are not written: they would land on the positions of rows in the band.)
//  Runs with a diagonal of more than PG_PIPE_WINDOW_A = 352 cells -- up to PG_PIPE_WINDOW = 432 -- take 9 rows of 512 positions.)
#define PWK_A 12
#define PWPOS_A 384
#define PWK_B 9
#define PWPOS_B 512
static_assert(PWK_B * PWPOS_B * 24 <= PRK * PNT * 24 + PST * PNT * 36 && PG_PIPE_WINDOW + 2 * (PWK_B - 2) + 8 <= PWPOS_B, "the wider wide ring");
#define PWK 12
#define PWPOS 384
#define PWAGE 10
#define PWROW_BYTES (PWPOS * 24)
static_assert(offsetof(PipeSmem, sx) == sizeof(double) * PRK * PNT * 3 && offsetof(PipeSmem, recL) - offsetof(PipeSmem, sx) == (size_t)PST * PNT * (3 * 8 + 3 * 4),
              "the staging arrays directly behind the ring");
static_assert(PWK * PWROW_BYTES <= PRK * PNT * 24 + PST * PNT * 36 && PWAGE + 2 <= PWK && PG_PIPE_WINDOW_A + 2 * PWAGE + 8 <= PWPOS, "wide ring inside ring + staging arrays");
#define PRING_BYTES (PRK * PNT * 24)
#define PROW_BYTES (PNT * 24)
// byte offset (inside the ring) of the ring row `age` diagonals before the row at `sb`
__device__ __forceinline__ int ring_back(int sb, int age) {
    const int x = sb - age * PROW_BYTES;
    return x < 0 ? x + PRING_BYTES : x;
}
} // namespace

// A class 5 step: out of line -- it is rare, and the kernel is as large as the instruction cache.
__device__ __attribute__((noinline)) void widest_step(const PgDevJob *job, cdesc8_p psc, int d, int lo, int hi, int tid,
                                                      bool no_terminal_edges, bool reduced_terminal, int stride = PNT) {
    const View J = load_view(job);
    const pg_i8 cur = psc[d];
    const pg_i8 p1 = psc[d > 0 ? d - 1 : 0], p2 = psc[d > 1 ? d - 2 : 0];
    const Diag g1 = {p1.x, d > 0 ? p1.y : p1.x - 1, ((long long)p1.s6 << 32) | (unsigned)p1.s5};
    const Diag g2 = {p2.x, d > 1 ? p2.y : p2.x - 1, ((long long)p2.s6 << 32) | (unsigned)p2.s5};
    const long long base = ((long long)cur.s6 << 32) | (unsigned)cur.s5;
    for (int i = lo + tid; i <= hi; i += stride)
        fill_cell_hbm(J, d, g1, g2, i, d - i, base + (i - lo), no_terminal_edges, reduced_terminal);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// assist_cell for sites with one or two bwd edges each (nearly all multi-edge cells: a site after a gap has the edge
// from its predecessor and the one that skips the gap): the up to 2 + 2 + 4 operand cells are read at once and the
// candidates evaluated as straight-line code in the reference's order.  An absent operand -- a missing second edge, or
// a gap candidate of the edge from the previous site, which the compute wave evaluates -- reads the all -inf null cell
// and cannot win.  With at most two edges the tie rule is simple: the staged X (Y) winner precedes the previous-site
// edge exactly when that edge is the SECOND one.
template <bool FAR>
__device__ __forceinline__ void assist2_cell(gdouble_w sc, cdesc8_p psc, int d, unsigned resmask, int slot, const pg_i4 &rL,
                                             const pg_i4 &cR, int row, int j, bool reduced_terminal, double go, double gex, double gey,
                                             double ng, double tM, double tX, double &ex, double &ey, double &em,
                                             unsigned &px, unsigned &py, unsigned &pm) {
    const double NI = neg_inf();
    const bool l1 = ((rL.x >> PR_NE_SHIFT) & 127) > 1, r1 = ((cR.x >> PR_NE_SHIFT) & 127) > 1;
    const int dL0 = rL.y & 0xffff, dL1 = (int)((unsigned)rL.y >> 16), dR0 = cR.y & 0xffff, dR1 = (int)((unsigned)cR.y >> 16);
    const double lw0 = (double)__int_as_float(rL.z), lw1 = (double)__int_as_float(rL.w);
    const double rw0 = (double)__int_as_float(cR.z), rw1 = (double)__int_as_float(cR.w);
    const bool xa_on = dL0 != 1, xb_on = l1 && dL1 != 1, ya_on = dR0 != 1, yb_on = r1 && dR1 != 1;
    double xa_x, xa_y, xa_m, xb_x, xb_y, xb_m, ya_x, ya_y, ya_m, yb_x, yb_y, yb_m;
    double m00x, m00y, m00m, m01x, m01y, m01m, m10x, m10y, m10m, m11x, m11y, m11m;
    if (!FAR) {
        auto cell = [&](int age, int p, bool present, double &xs, double &ys, double &ms) {
            int s_ = slot - age;
            s_ += s_ < 0 ? PRK : 0;
            const double *c = present ? &PM.sc[s_][p & (PNT - 1)][0] : &PM.null_cell[0];
            xs = c[PG_X]; ys = c[PG_Y]; ms = c[PG_M];
        };
        cell(dL0, row - dL0, xa_on, xa_x, xa_y, xa_m);
        cell(dL1, row - dL1, xb_on, xb_x, xb_y, xb_m);
        cell(dR0, row, ya_on, ya_x, ya_y, ya_m);
        cell(dR1, row, yb_on, yb_x, yb_y, yb_m);
        cell(dL0 + dR0, row - dL0, true, m00x, m00y, m00m);
        cell(dL0 + dR1, row - dL0, r1, m01x, m01y, m01m);
        cell(dL1 + dR0, row - dL1, l1, m10x, m10y, m10m);
        cell(dL1 + dR1, row - dL1, l1 && r1, m11x, m11y, m11m);
    } else {
        pg_d2 q[8];
        double qm[8];
        FarAsk fa[8];
        auto cell = [&](int k, int age, int p, bool present) {
            q[k].x = NI; q[k].y = NI; qm[k] = NI;
            fa[k].need = false; fa[k].boff = 0;
            if (!present) return;
            if (age < PAGE && ((resmask >> age) & 1u)) {
                int s_ = slot - age;
                s_ += s_ < 0 ? PRK : 0;
                q[k].x = PM.sc[s_][p & (PNT - 1)][PG_X]; q[k].y = PM.sc[s_][p & (PNT - 1)][PG_Y]; qm[k] = PM.sc[s_][p & (PNT - 1)][PG_M];
            } else {
                fa[k] = far_ask(psc, d, age, p);
            }
        };
        cell(0, dL0, row - dL0, xa_on);
        cell(1, dL1, row - dL1, xb_on);
        cell(2, dR0, row, ya_on);
        cell(3, dR1, row, yb_on);
        cell(4, dL0 + dR0, row - dL0, true);
        cell(5, dL0 + dR1, row - dL0, r1);
        cell(6, dL1 + dR0, row - dL1, l1);
        cell(7, dL1 + dR1, row - dL1, l1 && r1);
        far_fetch8(sc, fa, q, qm);
        xa_x = q[0].x; xa_y = q[0].y; xa_m = qm[0];  xb_x = q[1].x; xb_y = q[1].y; xb_m = qm[1];
        ya_x = q[2].x; ya_y = q[2].y; ya_m = qm[2];  yb_x = q[3].x; yb_y = q[3].y; yb_m = qm[3];
        m00x = q[4].x; m00y = q[4].y; m00m = qm[4];  m01x = q[5].x; m01y = q[5].y; m01m = qm[5];
        m10x = q[6].x; m10y = q[6].y; m10m = qm[6];  m11x = q[7].x; m11y = q[7].y; m11m = qm[7];
    }
    const unsigned aL0 = dL0 == 1 ? PG_BP_ADJL : 0u, aL1 = dL1 == 1 ? PG_BP_ADJL : 0u;
    const unsigned aR0 = dR0 == 1 ? PG_BP_ADJR : 0u, aR1 = dR1 == 1 ? PG_BP_ADJR : 0u;
    ex = NI; ey = NI; em = NI; px = PG_BP_NONE; py = PG_BP_NONE; pm = PG_BP_NONE;
    {   // X: the left edges that do not start at the previous site (VA:898-915); an edge in reach may start at site 0 (class 2)
        const double o0 = (FAR && reduced_terminal && row == dL0) ? 0.0 : go, o1 = (FAR && reduced_terminal && row == dL1) ? 0.0 : go;
        cand(xa_x + gex, PG_X, ex, px); cand((xa_y + 0.0) + go, PG_Y, ex, px); cand((xa_m + ng) + o0, PG_M, ex, px);
        const unsigned w1 = 1u << 4;
        cand(xb_x + gex, w1 | PG_X, ex, px); cand((xb_y + 0.0) + go, w1 | PG_Y, ex, px); cand((xb_m + ng) + o1, w1 | PG_M, ex, px);
    }
    {   // Y: the same for the right edges (VA:927-944)
        const double o0 = (FAR && reduced_terminal && j == dR0) ? 0.0 : go, o1 = (FAR && reduced_terminal && j == dR1) ? 0.0 : go;
        cand(ya_y + gey, PG_Y, ey, py); cand((ya_x + 0.0) + go, PG_X, ey, py); cand((ya_m + ng) + o0, PG_M, ey, py);
        const unsigned w1 = 1u << 18;
        cand(yb_y + gey, w1 | PG_Y, ey, py); cand((yb_x + 0.0) + go, w1 | PG_X, ey, py); cand((yb_m + ng) + o1, w1 | PG_M, ey, py);
    }
    {   // M: (left edge, right edge) pairs row-major (VA:1396-1433)
        unsigned w = aL0 | aR0;
        cand(((m00m + tM) + lw0) + rw0, w | PG_M, em, pm); cand(((m00x + tX) + lw0) + rw0, w | PG_X, em, pm);
        cand(((m00y + tX) + lw0) + rw0, w | PG_Y, em, pm);
        w = aL0 | aR1 | (1u << 18);
        cand(((m01m + tM) + lw0) + rw1, w | PG_M, em, pm); cand(((m01x + tX) + lw0) + rw1, w | PG_X, em, pm);
        cand(((m01y + tX) + lw0) + rw1, w | PG_Y, em, pm);
        w = aL1 | aR0 | (1u << 4);
        cand(((m10m + tM) + lw1) + rw0, w | PG_M, em, pm); cand(((m10x + tX) + lw1) + rw0, w | PG_X, em, pm);
        cand(((m10y + tX) + lw1) + rw0, w | PG_Y, em, pm);
        w = aL1 | aR1 | (1u << 4) | (1u << 18);
        cand(((m11m + tM) + lw1) + rw1, w | PG_M, em, pm); cand(((m11x + tX) + lw1) + rw1, w | PG_X, em, pm);
        cand(((m11y + tX) + lw1) + rw1, w | PG_Y, em, pm);
    }
    // `cand` keeps the incumbent's back-pointer while nothing beats -inf: an all -inf state ends as PG_BP_NONE
    const int adjL = dL0 == 1 ? 0 : ((l1 && dL1 == 1) ? 1 : -1), adjR = dR0 == 1 ? 0 : ((r1 && dR1 == 1) ? 1 : -1);
    px |= (unsigned)(adjL < 0 ? 0 : adjL) << 18;
    py |= (unsigned)(adjR < 0 ? 0 : adjR) << 4;
    px |= adjL < 0 ? PS_ONLY : ((adjL == 1 && ex > NI) ? PS_FIRST : 0u);
    py |= adjR < 0 ? PS_ONLY : ((adjR == 1 && ey > NI) ? PS_FIRST : 0u);
}

// assist_cell for a cell with ONE multi-edge site (one to three bwd edges) opposite a simple site -- what a diagonal
// usually holds: two multi-edge sites in one cell need a gap in both children at the same place.  The simple side's gap
// state is the compute wave's; staged are the multi-edge side's gap state over its edges that do not start at the
// previous site (operands (i - dL, j) for a left site, (i, j - dR) for a right one) and M over the pairs (edge k, the
// simple site's edge), k in list order, operands one diagonal further back.  THREE: some cell of the batch has a third
// edge (it comes from the LDS edge window, the first two travel in the site record).
template <bool FAR, bool THREE>
__device__ __forceinline__ void assist1_cell(gdouble_w sc, cdesc8_p psc, int d, unsigned resmask, int slot, const pg_i4 &rL,
                                             const pg_i4 &cR, int row, int j, bool reduced_terminal, double go, double gex, double gey,
                                             double ng, double tM, double tX, double &eg, double &em, unsigned &pg,
                                             unsigned &pm, bool &left) {
    const double NI = neg_inf();
    left = !(rL.x & PR_SIMPLE);                             // the multi-edge site is the left one
    const double gs = left ? gex : gey;                     // the gap state's extension rate (X: by column, Y: by row)
    const pg_i4 m = left ? rL : cR;
    const int site = left ? row : j;
    const int ne = (m.x >> PR_NE_SHIFT) & 127;
    const bool has1 = ne > 1, has2 = THREE && ne > 2;
    const int d0 = m.y & 0xffff, d1 = (int)((unsigned)m.y >> 16);
    const double w0 = (double)__int_as_float(m.z), w1 = (double)__int_as_float(m.w);
    int d2 = 2;
    double w2 = 0.0;
    if (THREE && has2) {
        const int e = (left ? PM.ebL : PM.ebR)[site & (PRW - 1)] + 2;
        d2 = site - (left ? PM.esL : PM.esR)[e & (PEC - 1)];
        w2 = (double)(left ? PM.ewL : PM.ewR)[e & (PEC - 1)];
    }
    // The simple side's edge weight is 0, and `+ 0.0` changes nothing but the sign of a zero, which no score of a job
    // on this kernel has (has_negative_zero, dp_abi.hip): ((s + t) + lw) + rw = (s + t) + w either way.
    const int ga0 = left ? row - d0 : row, ga1 = left ? row - d1 : row, ga2 = left ? row - d2 : row;               // gap operands
    const int mb0 = left ? row - d0 : row - 1, mb1 = left ? row - d1 : row - 1, mb2 = left ? row - d2 : row - 1;   // M operands
    const bool g0 = d0 != 1, g1 = has1 && d1 != 1, g2 = has2 && d2 != 1;
    double a0x, a0y, a0m, a1x, a1y, a1m, a2x = NI, a2y = NI, a2m = NI, b0x, b0y, b0m, b1x, b1y, b1m, b2x = NI, b2y = NI, b2m = NI;
    if (!FAR) {
        auto cell = [&](int age, int p, bool present, double &xs, double &ys, double &ms) {
            int s_ = slot - age;
            s_ += s_ < 0 ? PRK : 0;
            const double *c = present ? &PM.sc[s_][p & (PNT - 1)][0] : &PM.null_cell[0];
            xs = c[PG_X]; ys = c[PG_Y]; ms = c[PG_M];
        };
        cell(d0, ga0, g0, a0x, a0y, a0m);
        cell(d1, ga1, g1, a1x, a1y, a1m);
        cell(d0 + 1, mb0, true, b0x, b0y, b0m);
        cell(d1 + 1, mb1, has1, b1x, b1y, b1m);
        if (THREE) { cell(d2, ga2, g2, a2x, a2y, a2m); cell(d2 + 1, mb2, has2, b2x, b2y, b2m); }
    } else {
        pg_d2 q[8];
        double qm[8];
        FarAsk fa[8];
        q[6].x = NI; q[6].y = NI; qm[6] = NI; q[7] = q[6]; qm[7] = NI;
        fa[6].need = false; fa[6].boff = 0; fa[7] = fa[6];
        auto cell = [&](int k, int age, int p, bool present) {
            q[k].x = NI; q[k].y = NI; qm[k] = NI;
            fa[k].need = false; fa[k].boff = 0;
            if (!present) return;
            if (age < PAGE && ((resmask >> age) & 1u)) {
                int s_ = slot - age;
                s_ += s_ < 0 ? PRK : 0;
                q[k].x = PM.sc[s_][p & (PNT - 1)][PG_X]; q[k].y = PM.sc[s_][p & (PNT - 1)][PG_Y]; qm[k] = PM.sc[s_][p & (PNT - 1)][PG_M];
            } else {
                fa[k] = far_ask(psc, d, age, p);
            }
        };
        cell(0, d0, ga0, g0);
        cell(1, d1, ga1, g1);
        cell(2, d0 + 1, mb0, true);
        cell(3, d1 + 1, mb1, has1);
        cell(4, d2, ga2, g2);
        cell(5, d2 + 1, mb2, has2);
        far_fetch8(sc, fa, q, qm);
        a0x = q[0].x; a0y = q[0].y; a0m = qm[0];  a1x = q[1].x; a1y = q[1].y; a1m = qm[1];
        b0x = q[2].x; b0y = q[2].y; b0m = qm[2];  b1x = q[3].x; b1y = q[3].y; b1m = qm[3];
        a2x = q[4].x; a2y = q[4].y; a2m = qm[4];  b2x = q[5].x; b2y = q[5].y; b2m = qm[5];
    }
    // an edge of the multi-edge site that starts at site 0 opens a gap for free (BA.h:490-513); class 1 diagonals lie
    // PAGE rows and columns inside the matrix
    const double o0 = (FAR && reduced_terminal && site == d0) ? 0.0 : go, o1 = (FAR && reduced_terminal && site == d1) ? 0.0 : go;
    const double o2 = (FAR && reduced_terminal && site == d2) ? 0.0 : go;
    const unsigned adj = left ? PG_BP_ADJL : PG_BP_ADJR, other = left ? PG_BP_ADJR : PG_BP_ADJL;
    const unsigned kk = left ? (1u << 4) : (1u << 18);
    const unsigned self = left ? PG_X : PG_Y, cross = left ? PG_Y : PG_X;
    {   // the multi-edge side's gap state: per edge own state, the other gap state, M (VA:898-915 / 927-944)
        eg = NI; pg = PG_BP_NONE;
        const double s0 = left ? a0x : a0y, c0 = left ? a0y : a0x, s1 = left ? a1x : a1y, c1 = left ? a1y : a1x;
        cand(s0 + gs, self, eg, pg); cand(c0 + go, cross, eg, pg); cand((a0m + ng) + o0, PG_M, eg, pg);
        cand(s1 + gs, kk | self, eg, pg); cand(c1 + go, kk | cross, eg, pg); cand((a1m + ng) + o1, kk | PG_M, eg, pg);
        if (THREE) {
            const double s2 = left ? a2x : a2y, c2 = left ? a2y : a2x;
            cand(s2 + gs, 2 * kk | self, eg, pg); cand(c2 + go, 2 * kk | cross, eg, pg); cand((a2m + ng) + o2, 2 * kk | PG_M, eg, pg);
        }
    }
    {   // M: the (left edge, right edge) pairs in list order (VA:1396-1433)
        em = NI; pm = PG_BP_NONE;
        unsigned w = (d0 == 1 ? adj : 0u) | other;
        cand((b0m + tM) + w0, w | PG_M, em, pm); cand((b0x + tX) + w0, w | PG_X, em, pm); cand((b0y + tX) + w0, w | PG_Y, em, pm);
        w = (d1 == 1 ? adj : 0u) | kk | other;
        cand((b1m + tM) + w1, w | PG_M, em, pm); cand((b1x + tX) + w1, w | PG_X, em, pm); cand((b1y + tX) + w1, w | PG_Y, em, pm);
        if (THREE) {
            w = (d2 == 1 ? adj : 0u) | 2 * kk | other;
            cand((b2m + tM) + w2, w | PG_M, em, pm); cand((b2x + tX) + w2, w | PG_X, em, pm); cand((b2y + tX) + w2, w | PG_Y, em, pm);
        }
    }
    // merge information: the previous-site edge's slot, whether the staged winner precedes it, whether it exists at all
    const int adjs = d0 == 1 ? 0 : ((has1 && d1 == 1) ? 1 : ((has2 && d2 == 1) ? 2 : -1));
    const int win = (int)((pg >> (left ? 4 : 18)) & 127u);
    pg |= (unsigned)(adjs < 0 ? 0 : adjs) << (left ? 18 : 4);
    pg |= adjs < 0 ? PS_ONLY : ((eg > NI && win < adjs) ? PS_FIRST : 0u);
}

// ---- assist waves ----------------------------------------------------------------------------------------------
// Assist wave `a` stages, for every interior diagonal d with d % PNA == a (= its staging slot) that holds a multi-edge
// site (class 1 or 2), what assist_cell computes for the diagonal's multi-edge cells -- compacted, lane = cell -- at the
// compute lanes' positions (row % 256).  Nothing of it depends on diagonal d-1, so the wave computes d as soon as every
// compute wave has completed d-2 and is done, as a rule, before the compute waves merge it near the end of their step d:
// the multi-edge arithmetic, its LDS round trips and -- class 2 -- the L2 reads of cells that have left the ring run on a
// SIMD of their own.  With a model table too large for LDS the wave also gathers every cell's model score (classes
// 0..2) from L2, so that the compute waves never issue a load behind their stores.
// Ring safety: the compute waves cannot pass d before this wave has published d, a cell reads at most PAGE-1
// diagonals back, and step d' overwrites diagonal d'-PRK only: everything diagonal d reads is still in place.
// ---- assist waves: the general staging code --------------------------------------------------------------------------
// Scan of a diagonal's band for multi-edge cells, their records and model scores, every candidate that does not read diagonal
// d-1 in the reference's order, the staging stores.  pipe_assist<false> (large tables) runs it inline for every diagonal;
// pipe_assist<true> keeps the band's multi-edge sites in its lanes and stages values only, and calls this -- out of line,
// so that its registers do not weigh on the usual path -- for the few diagonals that path does not take.
template <bool TAB_LDS>
struct AssistGen {
    gdouble_w sc_out;
    cdesc8_p psc;
    gfloat_p table;
    int *list;
    double go, ge, ng, tng2, tng1;
    int S, lane;
    bool reduced_terminal;
    // row strips: a cell of the first / last column (row) extends its x-gap (y-gap) state at the terminal rate; nowhere else
    // does a cell of those reach this code (term_on = false)
    double gE;
    int Lx, Ly;
    bool term_on;
    // classification of one multi-edge cell: 1 one multi-edge site with <= 2 edges, 2 ... with 3, 3 two multi-edge sites
    // with <= 2 edges each, 4 anything else
    __device__ __forceinline__ int classify(const pg_i4 &rL, const pg_i4 &cR) const {
        const int nl = (rL.x >> PR_NE_SHIFT) & 127, nr = (cR.x >> PR_NE_SHIFT) & 127;
        const bool sL = rL.x & PR_SIMPLE, sR = cR.x & PR_SIMPLE;
        const int ne = sL ? nr : nl;
        if ((sL || sR) && (unsigned)(ne - 1) < 3u) return ne == 3 ? 2 : 1;
        if ((unsigned)(nl - 1) < 2u && (unsigned)(nr - 1) < 2u) return 3;
        return 4;
    }
    // the ring-dependent part for one batch of cells held in registers
    __device__ __forceinline__ void compute(int d, int cls, unsigned resmask, bool on, int row, int j, int kind, const pg_i4 &rL, const pg_i4 &cR,
                                            double tM, double tX) const {
        const int slot = d % PRK, stg = d % PST, at = row & (PNT - 1);
        const double gxc = (term_on && (j == 0 || j == Ly - 1)) ? gE : ge, gyc = (term_on && (row == 0 || row == Lx - 1)) ? gE : ge;
        if (__builtin_amdgcn_ballot_w64(on && kind >= 3) == 0) {
            // the usual batch: one multi-edge site per cell
            double eg = 0, em = 0;
            unsigned pg = 0, pm = 0;
            bool left = true;
            const bool three = __builtin_amdgcn_ballot_w64(on && kind == 2) != 0;
            if (on) {
                if (cls == 2) {
                    if (three) assist1_cell<true, true>(sc_out, psc, d, resmask, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                    else assist1_cell<true, false>(sc_out, psc, d, resmask, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                } else {
                    if (three) assist1_cell<false, true>(sc_out, psc, d, 0u, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                    else assist1_cell<false, false>(sc_out, psc, d, 0u, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                }
                if (left) { PM.sx[stg][at] = eg; PM.spx[stg][at] = pg; } else { PM.sy[stg][at] = eg; PM.spy[stg][at] = pg; }
                PM.sM[stg][at] = em;
                if (!TAB_LDS) PM.spm[stg][at] = pm;                // (small tables: spm's memory is the far-cell pool)
            }
        } else if (on) {
            double ex, ey, em;
            unsigned px, py, pm;
            if (kind == 4) {
                if (cls == 2) assist_cell<true>(sc_out, psc, d, slot, resmask, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, ex, ey, em, px, py, pm);
                else assist_cell<false>(sc_out, psc, d, slot, 0u, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, ex, ey, em, px, py, pm);
            } else if (kind == 2) {
                bool left;
                double eg;
                unsigned pg;
                if (cls == 2) assist1_cell<true, true>(sc_out, psc, d, resmask, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                else assist1_cell<false, true>(sc_out, psc, d, 0u, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, eg, em, pg, pm, left);
                ex = ey = eg; px = py = pg;                        // the compute wave reads the multi-edge side's only
            } else {
                if (cls == 2) assist2_cell<true>(sc_out, psc, d, resmask, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, ex, ey, em, px, py, pm);
                else assist2_cell<false>(sc_out, psc, d, 0u, slot, rL, cR, row, j, reduced_terminal, go, gxc, gyc, ng, tM, tX, ex, ey, em, px, py, pm);
            }
            PM.sx[stg][at] = ex; PM.sy[stg][at] = ey; PM.sM[stg][at] = em;
            PM.spx[stg][at] = px; PM.spy[stg][at] = py;
            if (!TAB_LDS) PM.spm[stg][at] = pm;
        }
    }
    // one batch from the compacted list into registers
    __device__ __forceinline__ void fetch_batch(int d, int stg, int n, bool &on, int &row, int &j, int &kind, pg_i4 &rL, pg_i4 &cR, double &tM, double &tX) const {
        on = lane < n;
        row = 0; j = 0; kind = 0; tM = 0; tX = 0;
        if (on) {
            row = list[lane]; j = d - row;
            rL = PM.recL[row & (PRW - 1)]; cR = PM.recR[j & (PRW - 1)];
            const int ti = (rL.x & 0xffff) + (cR.x & 0xffff) * S;
            if (TAB_LDS) { tM = PM.tab2[ti & 255][0]; tX = PM.tab2[ti & 255][1]; }
            else { const float sm = PM.ssm[stg][row & (PNT - 1)]; tM = tng2 + (double)sm; tX = tng1 + (double)sm; }
            kind = classify(rL, cR);
        }
    }
    // scan of a diagonal's band: model scores for large tables, the multi-edge cells compacted into `list`.  Returns their
    // number; with `flush` batches of 64 are computed as they fill up (the diagonal must be computable then).
    __device__ __forceinline__ int scan(int d, int cls, int lo, int hi, unsigned resmask, bool flush) const {
        const int stg = d % PST;
        int n_ns = 0, total = 0;
        for (int base = lo; base <= hi; base += 64) {
            const int row = base + lane, j = d - row;
            bool ns = false;
            if (row <= hi) {
                const pg_i4 rL = PM.recL[row & (PRW - 1)], cR = PM.recR[j & (PRW - 1)];
                if (!TAB_LDS) {
                    // (row strips: a cell of row 0 / column 0 is computed like any other; site 0 has no state, its cells' match
                    //  terms meet -inf operands only: any finite score will do)
                    const int sl = rL.x & 0xffff, sr = cR.x & 0xffff;
                    PM.ssm[stg][row & (PNT - 1)] = (sl < S && sr < S) ? far_f32(table + (sl + sr * S)) : 0.0f;
                }
                ns = cls != 0 && !(rL.x & cR.x & PR_SIMPLE);
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(ns);
            const int cnt = __builtin_popcountll(mask);
            if (cnt == 0) continue;
            total += cnt;
            if (n_ns + cnt > 64) {
                if (!flush) return total + 64;                     // too many for the register batch: the caller redoes it with flush
                bool on; int r_, j_, k_; pg_i4 a_ = {0, 0, 0, 0}, b_ = {0, 0, 0, 0}; double tm_, tx_;
                fetch_batch(d, stg, n_ns, on, r_, j_, k_, a_, b_, tm_, tx_);
                compute(d, cls, resmask, on, r_, j_, k_, a_, b_, tm_, tx_);
                n_ns = 0;
            }
            if (ns) list[n_ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = row;
            n_ns += cnt;
        }
        if (flush && n_ns > 0) {
            bool on; int r_, j_, k_; pg_i4 a_ = {0, 0, 0, 0}, b_ = {0, 0, 0, 0}; double tm_, tx_;
            fetch_batch(d, stg, n_ns, on, r_, j_, k_, a_, b_, tm_, tx_);
            compute(d, cls, resmask, on, r_, j_, k_, a_, b_, tm_, tx_);
        }
        return flush ? total : n_ns;
    }
};

template <bool TAB_LDS>
__device__ __forceinline__ AssistGen<TAB_LDS> assist_gen(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, bool reduced_terminal,
                                                          bool term_on = false) {
    AssistGen<TAB_LDS> g;
    g.gE = (double)job->gE; g.Lx = job->Lx; g.Ly = job->Ly; g.term_on = term_on;
    const float f_ng = job->ng;
    g.sc_out = (gdouble_w)job->sc; g.psc = psc; g.table = (gfloat_p)job->table; g.list = PM.as_list[a];
    g.go = (double)job->go; g.ge = (double)job->ge; g.ng = (double)f_ng;
    g.tng2 = (double)(2 * f_ng); g.tng1 = (double)(0.0f + f_ng);
    g.S = job->S; g.lane = lane; g.reduced_terminal = reduced_terminal;
    return g;
}
// one diagonal, start to finish (the diagonal must be computable: the compute waves have completed d-2)
__device__ __noinline__ void assist_general_diag(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, bool reduced_terminal,
                                                 int d, int cls, int lo, int hi, unsigned resmask, bool term_on) {
    const AssistGen<true> g = assist_gen<true>(job, psc, a, lane, reduced_terminal, term_on);
    g.scan(d, cls, lo, hi, resmask, true);
}
// the cells (row, j) of the lanes with `on` (the diagonal must be computable)
__device__ __noinline__ void assist_general_cells(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, bool reduced_terminal,
                                                  int d, int cls, unsigned resmask, bool on, int row, int j, bool term_on) {
    const AssistGen<true> g = assist_gen<true>(job, psc, a, lane, reduced_terminal, term_on);
    pg_i4 rL = {0, 0, 0, 0}, cR = {0, 0, 0, 0};
    double tM = 0, tX = 0;
    int kind = 0;
    if (on) {
        rL = PM.recL[row & (PRW - 1)]; cR = PM.recR[j & (PRW - 1)];
        const int ti = (rL.x & 0xffff) + (cR.x & 0xffff) * g.S;
        tM = PM.tab2[ti & 255][0]; tX = PM.tab2[ti & 255][1];
        kind = g.classify(rL, cR);
    }
    g.compute(d, cls, resmask, on, row, j, kind, rL, cR, tM, tX);
}

// ---- assist waves, small model tables: the band's multi-edge SITES kept in the lanes, values only, two diagonals a pass ----
// A multi-edge site stays in the band for as many diagonals as the band is wide, and the band only ever takes new rows at its
// upper end (imax never falls) and new columns at d - imin (never falls either).  Instead of scanning the band of every
// diagonal and decoding its multi-edge cells from the records, an assist wave holds one site per lane slot (side, site, state,
// up to three edges) and, per pass, looks at the rows and columns that are new.  A cell with a multi-edge site on both sides
// is the left site's.  The wave's two halves hold the same 32 slots and work on two diagonals at once -- this wave's next two
// class 2 diagonals, d and d + 3, when they follow each other -- so that everything up to the ring reads (tracking, the
// other site's record, the eight operand slots per cell, the cells fetched from L2) is done once per PAIR; the values of d
// are staged when the compute waves have completed d - 2, those of d + 3 three diagonals later.
//
// Per cell EIGHT operand slots -- 0,1 the left site's edges that do not start at the previous site (X from (row-dL, j)), 2,3
// the right site's (Y from (row, j-dR)), 4..7 the (left edge, right edge) pairs (M from (row-dL, j-dR)) with their two weights
// -- each an LDS offset into the ring (the all -inf cell if absent) or, for a cell that left the ring, into a pool of
// PFAR_POOL cells per assist wave (the memory of spm[a][.]: nobody reads staged M back-pointers of a small-table job) filled
// from L2 during the preparation.  Only VALUES are staged (the back-pointers come from pg_backptr), so the order of the
// candidates does not matter: a state's value is the maximum over its edges of max(own + ge, max(other, M + ng) + go), M's
// over its pairs of (max(M + tM, max(X, Y) + tX) + lw) + rw (the folding tools/gen_hot_asm.py explains).  What this path does
// not take (more than three edges at a site, three on both sides of a cell, an edge from site 0, more sites than slots, more
// far cells than the pool holds) goes to assist_general_diag when the diagonal is due.
// (an assist wave's part of a seven-wave wide run: wide_run7, further down)
// (returns the diagonal behind the run and the loader's progress it saw: rows, columns, descriptors)
__device__ __noinline__ pg_i4 assist_wide_run(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, unsigned flags, int d0,
                                              int rows_ld, int cols_ld, int diags_ld);
template <bool STRIP>
__device__ __noinline__ void pipe_assist_lean(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, unsigned flags) {
    // (a function of its own does not know that its arguments are wave-uniform: said here, what depends on the wave's index and
    //  the flags -- the staging slot's addresses among it -- lives in SGPRs instead of being spilled and reloaded in the staging code)
    a = __builtin_amdgcn_readfirstlane(a);
    flags = (unsigned)__builtin_amdgcn_readfirstlane((int)flags);
    const bool reduced_terminal = !(flags & 2u);
    const bool term_on = STRIP && !(flags & 1u);                   // (row strips: cells of the first / last row and column come this way too)
    constexpr int CLS = STRIP ? 7 : 15;                            // (a strip's class carries PG_STRIP_TERM beside it)
    const int nd = job->nd, S = job->S, Lx = job->Lx, Ly = job->Ly;
    const gdouble_w sc_out = (gdouble_w)job->sc;
    const double go = (double)job->go, ge = (double)job->ge, ng = (double)job->ng, gE = (double)job->gE;
    const int wave = PNW + a;                                      // names this wave in an abort tag
    int rows_ld = 0, cols_ld = 0, diags_ld = 0;
    int pw0 = -1, pw1 = -1, pw2 = -1, pw3 = -1;                   // cached progress of the compute waves
#ifdef PG_PIPE_STATS
    long long st_poll_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int st_poll_n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long as_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [0] cycles outside the wait for the compute waves, [1] inside it
    int as_n = 0, as_gen = 0, as_pairs = 0, as_cells = 0;          // diagonals staged; of those by the general code, whole; passes that held two; diagonals with cells staged by the general code
    int as_why[6] = {0, 0, 0, 0, 0, 0};                            // passes sent to the general code: slots, site shape, other side, cell shape / site 0, recent operand off the ring, pool
    long long as_t0 = __builtin_readcyclecounter();
#define ASTAMP(k) do { const long long t_ = __builtin_readcyclecounter(); as_t[k] += t_ - as_t0; as_t0 = t_; } while (0)
#else
#define ASTAMP(k)
#endif
    int *list = PM.as_list[a];
    const double NI_ = neg_inf();
    // the lane's slot
    int s_site = -1;                                               // the site (-1: free)
    bool s_left = true, s_gen = false;                             // s_gen: not for this path (no edge, more than three, a saturated distance)
    int s_n = 0, s_state = 0, s_a0 = 1, s_a1 = 1, s_a2 = 1;
    float s_w0 = 0.0f, s_w1 = 0.0f, s_w2 = 0.0f;
    int seen_row = 0, seen_col = 0;                                // rows / columns below these have been looked at
    bool trk_valid = true;                                         // false: more sites than slots -- start over with the next pass
    // the prepared pass: diagonal q_d (lanes 0..31) and, if q_d2 >= 0, q_d2 (lanes 32..63)
    int q_d = -1, q_d2 = -1, q_cls[2] = {0, 0}, q_lo[2] = {0, 0}, q_hi[2] = {0, 0};
    unsigned q_mask[2] = {0, 0};
    bool q_gen[2] = {false, false};                                // the diagonal goes to the general code
    int grp = 0;                                                   // which diagonal of the pass this lane works on (two diagonals: lane >> 5)
    bool q_on = false, q_bad = false;                              // this lane holds a cell / one that the general code has to stage
    int q_j = 0;
    bool q_gap2 = false, q_pair34 = false;                         // wave-uniform: some cell has a second gap operand / more than two pairs
    bool v_msL = false, v_msR = false, v_onlyL = false, v_onlyR = false;
    int q_row = 0, v_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double q_tM = 0, q_tX = 0, v_lw[4] = {0, 0, 0, 0}, v_rw[4] = {0, 0, 0, 0};
    int scan_d = a;                                                // next diagonal whose descriptor has not been looked at
    int wide_at = -1;                                              // >= 0: prepare() met a diagonal of a seven-wave wide run (wide_run7)
    if (STRIP) { const int d0 = job->d_first; scan_d = d0 + (a + 3 - d0 % 3) % 3; }       // (the strip's first diagonal of this wave's residue)

    // the sites first..last of one side have entered the band: the multi-edge ones among them into free slots
    auto take_sites = [&](bool left, int first, int last) -> bool {
        for (int base = first; base <= last; base += 64) {
            const int s_ = base + lane;
            bool multi = false;
            if (s_ <= last) multi = !((left ? PM.recL : PM.recR)[s_ & (PRW - 1)].x & PR_SIMPLE);
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(multi);
            if (mask == 0) continue;
            const int cnt = __builtin_popcountll(mask);
            const unsigned long long fmask = __builtin_amdgcn_ballot_w64(s_site < 0);
            if (cnt > __builtin_popcountll(fmask)) return false;
            if (multi) list[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = s_;
            if (s_site < 0) {
                const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fmask, 0u));
                if (r < cnt) {
                    const int site = list[r];
                    const pg_i4 rec = (left ? PM.recL : PM.recR)[site & (PRW - 1)];
                    s_site = site; s_left = left; s_n = (rec.x >> PR_NE_SHIFT) & 127; s_state = rec.x & 0xffff;
                    s_a0 = rec.y & 0xffff; s_a1 = (int)((unsigned)rec.y >> 16);
                    s_w0 = __int_as_float(rec.z); s_w1 = __int_as_float(rec.w);
                    s_a2 = 1; s_w2 = 0.0f;
                    if (s_n == 3) {
                        const int e = (left ? PM.ebL : PM.ebR)[site & (PRW - 1)] + 2;
                        s_a2 = site - (left ? PM.esL : PM.esR)[e & (PEC - 1)];
                        s_w2 = (left ? PM.ewL : PM.ewR)[e & (PEC - 1)];
                    }
                    if (s_n < 2) { s_a1 = 1; s_w1 = 0.0f; }         // an absent edge: distance 1, as the decoding expects
                    s_gen = s_n < 1 || s_n > 3 || s_a0 >= 65535 || s_a1 >= 65535;
                }
            }
        }
        return true;
    };

    // descriptors of scan_d (lanes 0..7) and scan_d + 3 (lanes 8..15), requested a pass ahead -- as a VECTOR load: a scalar load
    // shares its counter with the LDS operations and returns out of order, so the first LDS read behind it would wait for it
    auto desc_req = [&](int dd) {
        const int t = dd + ((lane >> 3) & 1) * PNA;
        return ((PG_GLOBAL const int *)psc)[8 * (t < nd ? t : nd) + (lane & 7)];           // (the array carries one entry of padding)
    };
    int q_next = desc_req(scan_d);

    // looks for this wave's next class 2 diagonal (and the one after, if it follows directly) and prepares the pass
    auto prepare = [&]() {
        q_d = -1; q_d2 = -1;
        while (scan_d < nd) {
            if (flag_load(&PM.abort_flag) != 0) return;
            const int d = scan_d, d2 = d + PNA;
            const int ax = __builtin_amdgcn_readlane(q_next, 0), ay = __builtin_amdgcn_readlane(q_next, 1), as4 = __builtin_amdgcn_readlane(q_next, 4);
            const int bx = __builtin_amdgcn_readlane(q_next, 8), by = __builtin_amdgcn_readlane(q_next, 9), bs4 = __builtin_amdgcn_readlane(q_next, 12);
            const int hop_a = (int)((unsigned)as4 >> 20);           // the host's hop count: straight to this wave's next diagonal with work
            // a seven-wave wide run (class 4, bit 19: every diagonal of such a run is a stop of the hop counts): this wave takes its rows
            if (!STRIP && ((as4 & 15) == 4 || (as4 & 15) == 5) && (as4 & (1 << 19))) { wide_at = d; return; }
            if ((as4 & CLS) != 2) { scan_d += PNA * hop_a; q_next = desc_req(scan_d); continue; }     // (class 0 / 1: the compute waves' own)
            const bool pair_d = hop_a == 1 && d2 < nd && (bs4 & CLS) == 2;              // the descriptors allow two diagonals
            // bit 4 marks a class 2 diagonal whose operands all lie in the ring (class 2 for the shape of a site): no residency test
            q_d = d; q_cls[0] = (as4 & 16) ? 1 : 2; q_lo[0] = ax; q_hi[0] = ay; q_mask[0] = ((unsigned)as4 >> 5) & 0x7fffu;
            q_d2 = -1; q_cls[1] = (bs4 & 16) ? 1 : 2; q_lo[1] = bx; q_hi[1] = by; q_mask[1] = ((unsigned)bs4 >> 5) & 0x7fffu;
            q_gen[0] = false; q_gen[1] = false; q_on = false; q_bad = false;
            const int lo_a = ax, cmin_a = d - ay;
            const int hi_t = pair_d ? by : ay, cmax_t = pair_d ? d2 - bx : d - ax;      // the later diagonal's upper ends
            if (rows_ld <= hi_t) rows_ld = POLL(&PM.loaded[0], hi_t + 1, 1);
            if (cols_ld <= cmax_t && cmax_t < Ly) cols_ld = POLL(&PM.loaded[1], cmax_t + 1, 2);
            ASTAMP(7);
            // ---- the slots ----
            if (!trk_valid) { s_site = -1; seen_row = lo_a; seen_col = cmin_a; trk_valid = true; }
            if (s_site >= 0 && s_site < (s_left ? lo_a : cmin_a)) s_site = -1;          // the band has left this row / column behind
            if (!take_sites(true, seen_row > lo_a ? seen_row : lo_a, hi_t) || !take_sites(false, seen_col > cmin_a ? seen_col : cmin_a, cmax_t)) {
                trk_valid = false;
                q_gen[0] = true;
                scan_d = d + PNA * hop_a; q_next = desc_req(scan_d);
#ifdef PG_PIPE_STATS
                ++as_why[0];
#endif
                return;
            }
            seen_row = hi_t + 1; seen_col = cmax_t + 1;
            ASTAMP(2);
            // two diagonals if the band's sites fit one half of the wave: the halves then look at the same sites (a lane's view e_*
            // of the slot with its rank), each on its own diagonal; otherwise every lane its own slot, on d
            const unsigned long long amask = __builtin_amdgcn_ballot_w64(s_site >= 0);
            const int n_act = __builtin_popcountll(amask);
            const bool pair = pair_d && n_act <= 32;
            scan_d = pair ? d2 + PNA * (int)((unsigned)bs4 >> 20) : d + PNA * hop_a;
            q_next = desc_req(scan_d);
            q_d2 = pair ? d2 : -1;
            int e_site = s_site, e_a0 = s_a0, e_a1 = s_a1, e_a2 = s_a2, e_n = s_n, e_state = s_state;
            bool e_left = s_left, e_gen = s_gen;
            float e_w0 = s_w0, e_w1 = s_w1, e_w2 = s_w2;
            grp = 0;
            if (pair) {
                grp = lane >> 5;
                if (s_site >= 0) list[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(amask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)amask, 0u))] = lane;
                const int idx = lane & 31;
                const bool have = idx < n_act;
                const int src4 = 4 * (have ? list[idx] : lane);
                const int pk = (int)s_left | ((int)s_gen << 1) | (s_n << 2) | (s_state << 9);
                const int pk_ = __builtin_amdgcn_ds_bpermute(src4, pk);
                const int site_ = __builtin_amdgcn_ds_bpermute(src4, s_site);      // (every lane takes part: a lane that is off cannot be read from)
                e_site = have ? site_ : -1;
                e_left = pk_ & 1; e_gen = (pk_ >> 1) & 1; e_n = (pk_ >> 2) & 127; e_state = (pk_ >> 9) & 0xffff;
                e_a0 = __builtin_amdgcn_ds_bpermute(src4, s_a0); e_a1 = __builtin_amdgcn_ds_bpermute(src4, s_a1); e_a2 = __builtin_amdgcn_ds_bpermute(src4, s_a2);
                e_w0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(s_w0)));
                e_w1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(s_w1)));
                e_w2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(s_w2)));
            }
            ASTAMP(8);
            // ---- this lane's cell: (site, dg - site) or (dg - site, site) ----
            const int dg = grp ? d2 : d, lo_g = grp ? bx : ax, hi_g = grp ? by : ay;
            const bool cls1_g = (grp ? q_cls[1] : q_cls[0]) == 1;
            const unsigned mask_g = grp ? q_mask[1] : q_mask[0];
            const bool act = e_site >= 0;
            const int row = act ? (e_left ? e_site : dg - e_site) : 0, j = act ? dg - row : 0;
            const bool inb = act && row >= lo_g && row <= hi_g;
            pg_i4 o = {PR_SIMPLE | (1 << PR_NE_SHIFT), 1, 0, 0};
            if (inb) o = e_left ? PM.recR[j & (PRW - 1)] : PM.recL[row & (PRW - 1)];
            const bool oS = o.x & PR_SIMPLE;
            const int nO = (o.x >> PR_NE_SHIFT) & 127;
            bool on = inb && (e_left || oS);                       // (both sites multi-edge: the left site's lane)
            // the other side's (at most two) edges come with its record; three there: not for this path
            const int oa0 = o.y & 0xffff, oa1 = nO > 1 ? (int)((unsigned)o.y >> 16) : 1;
            const float of0 = __int_as_float(o.z), of1 = nO > 1 ? __int_as_float(o.w) : 0.0f;
            bool ok = !(on && (e_gen || nO < 1 || nO > 2 || oa0 >= 65535 || oa1 >= 65535));
#ifdef PG_PIPE_STATS
            if (__builtin_amdgcn_ballot_w64(on && e_gen) != 0) ++as_why[1];
            else if (__builtin_amdgcn_ballot_w64(!ok) != 0) ++as_why[2];
            const bool ok_before_slots = ok;
            bool ok_recent = true;
#endif
            const int nL = on ? (e_left ? e_n : nO) : 1, nR = on ? (e_left ? nO : e_n) : 1;
            const int dL0 = !on ? 1 : (e_left ? e_a0 : oa0), dL1 = !on ? 1 : (e_left ? e_a1 : oa1), dL2 = (on && e_left) ? e_a2 : 1;
            const int dR0 = !on ? 1 : (e_left ? oa0 : e_a0), dR1 = !on ? 1 : (e_left ? oa1 : e_a1), dR2 = (on && !e_left) ? e_a2 : 1;
            const float lw0 = e_left ? e_w0 : of0, lw1 = e_left ? e_w1 : of1, lw2 = e_left ? e_w2 : 0.0f;
            const float rw0 = e_left ? of0 : e_w0, rw1 = e_left ? of1 : e_w1, rw2 = e_left ? 0.0f : e_w2;
            q_row = row; q_j = j;
            {
                const int ti = ((e_left ? e_state : (o.x & 0xffff)) + (e_left ? (o.x & 0xffff) : e_state) * S) & 255;
                q_tM = PM.tab2[ti][0]; q_tX = PM.tab2[ti][1];
            }
            v_msL = e_left; v_msR = e_left ? !oS : true;
            // ---- the eight operand slots ----
            bool any_far = false, v_far[8];
            int f_age[8], f_p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { f_age[k] = 0; f_p[k] = 0; v_far[k] = false; v_off[k] = (int)offsetof(PipeSmem, null_cell); }
            {
                const int slot = grp ? d2 % PRK : d % PRK;
                const int null_off = (int)offsetof(PipeSmem, null_cell), ring_off = (int)offsetof(PipeSmem, sc);
                // one operand: in the ring (all of them where the host says so), or asked of L2 below
                auto put = [&](int k, bool present, int age, int p_) {
                    const bool resident = cls1_g || (age < PAGE && ((mask_g >> (age & 31)) & 1u));
                    int s_ = slot - age;
                    s_ += s_ < 0 ? PRK : 0;
                    const int off = ring_off + (s_ * PNT + (p_ & (PNT - 1))) * 24;
                    v_off[k] = (present && resident) ? off : null_off;
#ifdef PG_EXP_NOFAR                                                  // timing experiment (wrong results): operands that left the ring read as -inf
                    const bool far = false;
#else
                    const bool far = present && !resident;
#endif
                    ok = ok && !(far && age < PAGE);               // a recent diagonal that is not in the ring (after a wide run): not landed yet
#ifdef PG_PIPE_STATS
                    ok_recent = ok_recent && !(far && age < PAGE);
#endif
                    v_far[k] = far; any_far = any_far || far; f_age[k] = age; f_p[k] = p_;
                };
                // gap operands: the (at most two) edges of a side that do not start at the previous site
                const bool naL0 = dL0 != 1, naL1 = dL1 != 1, naL2 = dL2 != 1;
                const bool naR0 = dR0 != 1, naR1 = dR1 != 1, naR2 = dR2 != 1;
                const int cL = (int)naL0 + (int)naL1 + (int)naL2, cR_ = (int)naR0 + (int)naR1 + (int)naR2;
                if (on && (cL > 2 || cR_ > 2 || nL * nR > 4 || nL < 1 || nR < 1)) ok = false;
                const int gL0 = naL0 ? dL0 : (naL1 ? dL1 : dL2), gL1 = (naL0 && naL1) ? dL1 : dL2;
                const int gR0 = naR0 ? dR0 : (naR1 ? dR1 : dR2), gR1 = (naR0 && naR1) ? dR1 : dR2;
                put(0, on && cL >= 1, gL0, row - gL0);
                put(2, on && cR_ >= 1, gR0, row);
                q_gap2 = __builtin_amdgcn_ballot_w64(on && (cL >= 2 || cR_ >= 2)) != 0;
                if (q_gap2) {
                    put(1, on && cL >= 2, gL1, row - gL1);
                    put(3, on && cR_ >= 2, gR1, row);
                }
                v_onlyL = cL == nL; v_onlyR = cR_ == nR;            // no edge from the previous site: the staged value IS the state's
                // pairs, row-major (at most four): t -> (k1, k2)
                q_pair34 = __builtin_amdgcn_ballot_w64(on && nL * nR > 2) != 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= 2 && !q_pair34) break;
                    const int k1 = nR == 1 ? t : (nR == 2 ? t >> 1 : (t >= 3 ? 1 : 0)), k2 = t - k1 * nR;
                    const bool present = on && t < nL * nR && k1 < 3;
                    const int a1 = k1 == 0 ? dL0 : (k1 == 1 ? dL1 : dL2), a2 = k2 == 0 ? dR0 : (k2 == 1 ? dR1 : dR2);
                    v_lw[t] = (double)(k1 == 0 ? lw0 : (k1 == 1 ? lw1 : lw2));
                    v_rw[t] = (double)(k2 == 0 ? rw0 : (k2 == 1 ? rw1 : rw2));
                    put(4 + t, present, a1 + a2, row - a1);
                }
                // an edge that starts at site 0 opens a gap for free (BA.h:490-513): left to the general code
                if (on && ((naL0 && row == dL0) || (naL1 && row == dL1) || (naL2 && row == dL2) ||
                           (naR0 && j == dR0) || (naR1 && j == dR1) || (naR2 && j == dR2))) ok = false;
            }
#ifdef PG_PIPE_STATS
            if (__builtin_amdgcn_ballot_w64(on && ok_before_slots && !ok_recent) != 0) ++as_why[4];
            else if (__builtin_amdgcn_ballot_w64(on && ok_before_slots && !ok) != 0) ++as_why[3];
#endif
            // a cell this path does not take: the general code stages it when its diagonal is due
            q_bad = on && !ok;
            on = on && ok;
            ASTAMP(3);
            if (__builtin_amdgcn_ballot_w64(on && any_far) != 0) {
                // cells that left the ring (>= PAGE diagonals back) have landed once every wave has completed d - PAGE + PLAND
                // (far_ask looks up the descriptors of the diagonals d - age, age >= PAGE, in the loader's window)
                const int dl = pair ? d2 : d;
                if (diags_ld < dl - PAGE + 1) diags_ld = POLL(&PM.loaded[2], dl - PAGE + 1, 5);
                if (pw0 < dl - PAGE + PLAND) pw0 = POLL(&PM.progress[0], dl - PAGE + PLAND, 9);
                if (pw1 < dl - PAGE + PLAND) pw1 = POLL(&PM.progress[1], dl - PAGE + PLAND, 9);
                if (pw2 < dl - PAGE + PLAND) pw2 = POLL(&PM.progress[2], dl - PAGE + PLAND, 9);
                if (pw3 < dl - PAGE + PLAND) pw3 = POLL(&PM.progress[3], dl - PAGE + PLAND, 9);
                FarAsk fa[8];
                pg_d2 fxy[8];
                double fm_[8];
                int n_pool = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fa[k].need = false; fa[k].boff = 0; fxy[k].x = NI_; fxy[k].y = NI_; fm_[k] = NI_;
                    if (__builtin_amdgcn_ballot_w64(on && v_far[k]) == 0) continue;      // (wave-uniform: nobody's operand k left the ring)
                    if (on && v_far[k]) fa[k] = far_ask(psc, dg, f_age[k], f_p[k]);      // (need = false outside the band: the slot keeps -inf)
                    // its place in the pool
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(fa[k].need);
                    if (fa[k].need) {
                        const int at_ = n_pool + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                        v_off[k] = (int)((const char *)&PM.spm[a][0] - (const char *)&PM) + 24 * at_;
                    }
                    n_pool += __builtin_popcountll(mk);
                }
#ifdef PG_PIPE_STATS
                if (n_pool > PFAR_POOL) ++as_why[5];
#endif
                if (n_pool > PFAR_POOL) { q_gen[0] = true; q_gen[1] = pair; on = false; q_bad = false; }      // (more far cells than the pool holds)
                else {
                    far_fetch8(sc_out, fa, fxy, fm_);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (fa[k].need) {
                            double *c = (double *)((char *)&PM + v_off[k]);
                            c[PG_X] = fxy[k].x; c[PG_Y] = fxy[k].y; c[PG_M] = fm_[k];
                        }
                }
            }
            q_on = on;
            ASTAMP(4);
            return;
        }
    };

    // one diagonal of the prepared pass: waits until it is due, stages its cells, publishes it
    auto stage = [&](int g) {
        const int d = g ? q_d2 : q_d;
        ASTAMP(0);
        // every compute wave has completed d-2 (or sleeps through it); a poll that ran into an abort returns "done" and
        // the wave runs to the end of its list on whatever is in the ring (reads stay inside the arena)
        if (pw0 < d - 2) pw0 = POLL(&PM.progress[0], d - 2, 9);
        if (pw1 < d - 2) pw1 = POLL(&PM.progress[1], d - 2, 9);
        if (pw2 < d - 2) pw2 = POLL(&PM.progress[2], d - 2, 9);
        if (pw3 < d - 2) pw3 = POLL(&PM.progress[3], d - 2, 9);
        ASTAMP(1);
        if (q_gen[g]) {
            if (q_cls[g] == 2 && diags_ld < d) diags_ld = POLL(&PM.loaded[2], d, 5);
#ifdef PG_PIPE_STATS
            ++as_gen;
#endif
            assist_general_diag(job, psc, a, lane, reduced_terminal, d, q_cls[g], q_lo[g], q_hi[g], q_mask[g], term_on);
        } else if (q_on && grp == g) {
            const char *base = (const char *)&PM;
            double cx[8], cy[8], cm[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cx[k] = NI_; cy[k] = NI_; cm[k] = NI_;
                if (((k == 1 || k == 3) && !q_gap2) || (k >= 6 && !q_pair34)) continue;      // (wave-uniform: no cell of the pass has it)
                const double *c = (const double *)(base + v_off[k]);
                cx[k] = c[PG_X]; cy[k] = c[PG_Y]; cm[k] = c[PG_M];
            }
            auto gapv = [&](double own, double other, double m_, double ext) { return __builtin_fmax(own + ext, __builtin_fmax(other, m_ + ng) + go); };
            const double gxc = (term_on && (q_j == 0 || q_j == Ly - 1)) ? gE : ge, gyc = (term_on && (q_row == 0 || q_row == Lx - 1)) ? gE : ge;
            double ex = gapv(cx[0], cy[0], cm[0], gxc), ey = gapv(cy[2], cx[2], cm[2], gyc);
            if (q_gap2) { ex = __builtin_fmax(ex, gapv(cx[1], cy[1], cm[1], gxc)); ey = __builtin_fmax(ey, gapv(cy[3], cx[3], cm[3], gyc)); }
            double em = NI_;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= 2 && !q_pair34) break;
                const double v = (__builtin_fmax(cm[4 + t] + q_tM, __builtin_fmax(cx[4 + t], cy[4 + t]) + q_tX) + v_lw[t]) + v_rw[t];
                em = __builtin_fmax(em, v);
            }
            const int stg = a, at = q_row & (PNT - 1);              // d % PST == d % PNA == a
            if (v_msL) { PM.sx[stg][at] = ex; PM.spx[stg][at] = v_onlyL ? PS_ONLY : 0u; }
            if (v_msR) { PM.sy[stg][at] = ey; PM.spy[stg][at] = v_onlyR ? PS_ONLY : 0u; }
            PM.sM[stg][at] = em;
        }
        ASTAMP(5);
        if (!q_gen[g] && __builtin_amdgcn_ballot_w64(q_bad && grp == g) != 0) {
            if (q_cls[g] == 2 && diags_ld < d) diags_ld = POLL(&PM.loaded[2], d, 5);
#ifdef PG_PIPE_STATS
            ++as_cells;
#endif
            assist_general_cells(job, psc, a, lane, reduced_terminal, d, q_cls[g], q_mask[g], q_bad && grp == g, q_row, q_j, term_on);
        }
        ASTAMP(6);
        flag_store(&PM.assist_done[a], d);
#ifdef PG_PIPE_STATS
        ++as_n;
#endif
    };

    prepare();
    while (q_d >= 0 || wide_at >= 0) {
        if (flag_load(&PM.abort_flag) != 0) break;
        if (wide_at >= 0) {
            // the run's first diagonal: this wave stopped at the first one of its residue, at most two behind it
            int d0 = wide_at;
            const int run_cls = psc[wide_at].s4 & 15;              // (a run of class 4 may follow one of class 5 directly, or the other way)
            for (int k = 0; k < PNA - 1 && d0 > 0; ++k) { if ((psc[d0 - 1].s4 & 15) == run_cls) --d0; else break; }
            const pg_i4 back = assist_wide_run(job, psc, a, lane, flags, d0, rows_ld, cols_ld, diags_ld);
            const int e = __builtin_amdgcn_readfirstlane(back.x);
            rows_ld = __builtin_amdgcn_readfirstlane(back.y); cols_ld = __builtin_amdgcn_readfirstlane(back.z); diags_ld = __builtin_amdgcn_readfirstlane(back.w);
            scan_d = e + ((a - e) % PNA + PNA) % PNA;               // this wave's first diagonal behind the run
            q_next = desc_req(scan_d);
            trk_valid = false;                                     // (the band moved on without this wave looking)
            wide_at = -1;
            prepare();
            continue;
        }
        stage(0);
        if (q_d2 >= 0) {
#ifdef PG_PIPE_STATS
            ++as_pairs;
#endif
            stage(1);
        }
        prepare();
    }
#ifdef PG_PIPE_STATS
    if (lane == 0 && 3 * (job->Lx + job->Ly) >= 4096 && PG_STATS_MINE(job, flags)) {
        PG_GLOBAL int *o = (PG_GLOBAL int *)job->trace + 3 * (job->Lx + job->Ly) - 1000 + 16 * a;
        o[0] = as_n;
        for (int k = 0; k < 12; ++k) o[1 + k] = (int)(as_t[k] >> 8);
        o[13] = as_gen; o[14] = as_pairs; o[15] = as_cells;
#ifndef PG_AS_BUCKETS                                              // (-DPG_AS_BUCKETS: the cycle buckets 2..6 of the lean path instead)
        for (int k = 0; k < 6; ++k) o[1 + 2 + k] = as_why[k];     // (in the place of the unused cycle buckets 2..7)
#endif
    }
#endif
#undef ASTAMP
}

// ---- assist waves, large model tables: every multi-edge cell of the class 0..2 diagonals, staged with back-pointer words ----
template <bool TAB_LDS, bool STRIP>
__device__ __noinline__ void pipe_assist(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, unsigned flags) {
    static_assert(!TAB_LDS, "small tables: pipe_assist_lean");
    a = __builtin_amdgcn_readfirstlane(a);                         // (wave-uniform arguments: see pipe_assist_lean)
    flags = (unsigned)__builtin_amdgcn_readfirstlane((int)flags);
    const bool reduced_terminal = !(flags & 2u);
    const bool term_on = STRIP && !(flags & 1u);                   // (row strips: cells of the first / last row and column come this way too)
    const int nd = job->nd, S = job->S;
    const gdouble_w sc_out = (gdouble_w)job->sc;
    const gfloat_p table = (gfloat_p)job->table;
    const float f_ng = job->ng;
    const double go = (double)job->go, ge = (double)job->ge, ng = (double)f_ng;
    const double tng2 = (double)(2 * f_ng), tng1 = (double)(0.0f + f_ng);
    const int wave = PNW + a;                                      // names this wave in an abort tag
    int rows_ld = 0, cols_ld = 0, diags_ld = 0;
    int pw0 = -1, pw1 = -1, pw2 = -1, pw3 = -1;                   // cached progress of the compute waves
#ifdef PG_PIPE_STATS
    long long st_poll_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int st_poll_n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int *list = PM.as_list[a];
    const double NI_ = neg_inf();
    // Software pipeline: everything a diagonal's batch needs that does NOT come out of the ring -- its descriptor, the scan
    // of the band for multi-edge cells, their site records, edge data and model scores -- is prepared right after the
    // previous diagonal has been published, while the compute waves are still steps away.  What is left between "the
    // compute waves have completed d-2" and "d is staged" is the ring reads, the candidates and the staging stores.
    int q_d = -1, q_cls = 0, q_lo = 0, q_hi = 0, q_n = 0;          // the prepared diagonal (q_d < 0: none)
    unsigned q_mask = 0;
    bool q_big = false;                                            // more than 64 multi-edge cells: not prepared, done in batches
    int q_row = 0, q_j = 0, q_kind = 0;
    pg_i4 q_rL = {0, 0, 0, 0}, q_cR = {0, 0, 0, 0};
    double q_tM = 0, q_tX = 0;
    // the usual batch (class 1, every cell one multi-edge site with at most two edges) decoded down to LDS offsets of its four
    // operand cells, weights and back-pointer words: what remains for the critical chain is 12 LDS reads and 12 candidates
    bool q_fast = false, q_left = true;
    int q_oa0 = 0, q_oa1 = 0, q_ob0 = 0, q_ob1 = 0, q_adjs = -1;
    unsigned q_e0 = 0, q_e1 = 0;                                   // ADJ flag / slot bits of the two edges in an M back-pointer
    double q_w0 = 0, q_w1 = 0;
    // class 2 batches of the usual shape: the operand cells that left the ring, fetched from L2 while the batch is prepared
    // (off the chain "compute waves completed d-2 -> d is staged": what remains there is the same as for class 1)
    double q_fx[4] = {0, 0, 0, 0}, q_fy[4] = {0, 0, 0, 0}, q_fm[4] = {0, 0, 0, 0};
    bool q_far[4] = {false, false, false, false};
    int scan_d = a;                                                // next diagonal whose descriptor has not been looked at
    if (STRIP) { const int d0 = job->d_first; scan_d = d0 + (a + 3 - d0 % 3) % 3; }       // (the strip's first diagonal of this wave's residue)

#ifdef PG_PIPE_STATS
    long long as_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // cycles: prepare (rest), waiting for the compute waves, compute, publish; inside prepare: scan, batch, decode, far
    int as_n = 0;
    // (PG_ASSIST_RUNS_ONLY: only diagonals that directly follow this wave's previous one -- the inside of a class 2 run, where the
    // assist waves are what the compute waves wait for -- are counted; otherwise the idle time between runs dominates the averages)
    bool as_on = true;
    int as_prev = -100;
#define ASTAMP(k) do { const long long t_ = __builtin_readcyclecounter(); if (as_on) as_t[(k) == 12 ? 0 : (k)] += t_ - as_t0; as_t0 = t_; } while (0)
    long long as_t0 = __builtin_readcyclecounter();
#else
#define ASTAMP(k)
#endif
    const AssistGen<TAB_LDS> gen = assist_gen<TAB_LDS>(job, psc, a, lane, reduced_terminal, term_on);     // (large tables: the general code, inline)
    // looks for the next diagonal of this wave with work and prepares it
    // descriptor of scan_d, requested a pass ahead -- as a VECTOR load (lanes 0..7 one word each): a scalar load shares its
    // counter with the LDS operations and returns out of order, so the first LDS read behind it would wait for it (~2k cycles)
    auto desc_req = [&](int dd) { return ((PG_GLOBAL const int *)psc)[8 * (dd < nd ? dd : nd) + (lane & 7)]; };   // (the array carries one entry of padding)
    int q_next = desc_req(scan_d);
    auto prepare = [&]() {
        q_d = -1;
        while (scan_d < nd) {
            if (flag_load(&PM.abort_flag) != 0) return;
            const int d = scan_d;
            const int cur_x = __builtin_amdgcn_readlane(q_next, 0), cur_y = __builtin_amdgcn_readlane(q_next, 1);
            const int cur_s4 = __builtin_amdgcn_readlane(q_next, 4);
            scan_d += PNA * (int)((unsigned)cur_s4 >> 20);          // the host's hop count: straight to this wave's next diagonal with work
            q_next = desc_req(scan_d);
            const int cls0 = cur_s4 & 15;
            if (!(cls0 == 2 || (!TAB_LDS && cls0 <= 1))) continue;  // small tables: class 1 is the compute waves' own (hot_run)
            // small tables: bit 4 marks a class 2 diagonal whose operands all lie in the ring (class 2 for the shape of a site):
            // the ring-only code paths, as for class 1
            const int cls = (TAB_LDS && cls0 == 2 && (cur_s4 & 16)) ? 1 : cls0;
            const int lo = cur_x, hi = cur_y;
            if (rows_ld <= hi) rows_ld = POLL(&PM.loaded[0], hi + 1, 1);
            {   // (a row strip's diagonals behind its last cell hold no row: nothing to wait for beyond the last column)
                const int cmax = d - lo < job->Ly - 1 ? d - lo : job->Ly - 1;
                if (hi >= lo && cols_ld <= cmax) cols_ld = POLL(&PM.loaded[1], cmax + 1, 2);
            }
            q_d = d; q_cls = cls; q_lo = lo; q_hi = hi; q_mask = ((unsigned)cur_s4 >> 5) & 0x7fffu;
            // large tables: the scan writes the model scores into staging slot d % PST, whose previous user, diagonal
            // d - PST, the compute waves must have completed
            if (!TAB_LDS) {
                if (pw0 < d - PST) pw0 = POLL(&PM.progress[0], d - PST, 9);
                if (pw1 < d - PST) pw1 = POLL(&PM.progress[1], d - PST, 9);
                if (pw2 < d - PST) pw2 = POLL(&PM.progress[2], d - PST, 9);
                if (pw3 < d - PST) pw3 = POLL(&PM.progress[3], d - PST, 9);
            }
            ASTAMP(0);
            const int n = gen.scan(d, cls, lo, hi, q_mask, false);
            ASTAMP(4);
            q_big = n > 64;
            q_n = q_big ? 0 : n;
            q_fast = false;
            if (!q_big) {
                bool on;
                gen.fetch_batch(d, d % PST, q_n, on, q_row, q_j, q_kind, q_rL, q_cR, q_tM, q_tX);
                ASTAMP(5);
                q_fast = !TAB_LDS && cls == 1 && q_n > 0 && __builtin_amdgcn_ballot_w64(on && q_kind != 1) == 0;
                if (q_fast) {
                    bool ok = true, any_far = false;
                    int f_age[4] = {0, 0, 0, 0}, f_p[4] = {0, 0, 0, 0};
                    q_far[0] = q_far[1] = q_far[2] = q_far[3] = false;
                    if (on) {
                        // assist1_cell<., false>'s operand selection, done ahead of time
                        q_left = !(q_rL.x & PR_SIMPLE);
                        const pg_i4 m = q_left ? q_rL : q_cR;
                        const int site = q_left ? q_row : q_j;
                        const bool has1 = ((m.x >> PR_NE_SHIFT) & 127) > 1;
                        const int d0 = m.y & 0xffff, d1 = (int)((unsigned)m.y >> 16);
                        q_w0 = (double)__int_as_float(m.z); q_w1 = (double)__int_as_float(m.w);
                        const int slot = d % PRK;
                        // an operand `age` diagonals back in ring column p: its ring offset, the all -inf cell if it is absent, or
                        // -- class 2, not in the ring -- a request for L2 (k = which of the four)
                        auto cell_off = [&](int k, int age, int p, bool present) {
                            const bool resident = cls == 1 || (age < PAGE && ((q_mask >> age) & 1u));
                            if (present && !resident) {
                                q_far[k] = true; any_far = true; f_age[k] = age; f_p[k] = p;
                                if (age < PAGE) ok = false;        // a recent diagonal that is not in the ring (after a wide run): not landed yet
                            }
                            int s_ = slot - age;
                            s_ += s_ < 0 ? PRK : 0;
                            const double *c = (present && resident) ? &PM.sc[s_ < 0 ? 0 : s_][p & (PNT - 1)][0] : &PM.null_cell[0];
                            return (int)((const char *)c - (const char *)&PM);
                        };
                        q_oa0 = cell_off(0, d0, q_left ? q_row - d0 : q_row, d0 != 1);
                        q_oa1 = cell_off(1, d1, q_left ? q_row - d1 : q_row, has1 && d1 != 1);
                        q_ob0 = cell_off(2, d0 + 1, q_left ? q_row - d0 : q_row - 1, true);
                        q_ob1 = cell_off(3, d1 + 1, q_left ? q_row - d1 : q_row - 1, has1);
                        // an edge that starts at site 0 opens a gap for free (BA.h:490-513): left to the general code
                        if (cls == 2 && ((d0 != 1 && site == d0) || (has1 && d1 != 1 && site == d1))) ok = false;
                        const unsigned adj = q_left ? PG_BP_ADJL : PG_BP_ADJR, other = q_left ? PG_BP_ADJR : PG_BP_ADJL;
                        const unsigned kk = q_left ? (1u << 4) : (1u << 18);
                        q_e0 = (d0 == 1 ? adj : 0u) | other;
                        q_e1 = (d1 == 1 ? adj : 0u) | kk | other;
                        q_adjs = d0 == 1 ? 0 : ((has1 && d1 == 1) ? 1 : -1);
                    }
                    if (__builtin_amdgcn_ballot_w64(on && !ok) != 0) q_fast = false;
                    else if (__builtin_amdgcn_ballot_w64(on && any_far) != 0) {
                        // the cells that left the ring (>= PAGE diagonals back: landed, every wave keeps all but its last eight
                        // steps' stores retired), requested together and waited for here: this wave has nothing else in flight
                        if (diags_ld < d) diags_ld = POLL(&PM.loaded[2], d, 5);
                        // (this may run any number of diagonals ahead of the compute waves: a cell PAGE or more diagonals back
                        // has landed once every wave has completed d - PAGE + PLAND)
                        if (pw0 < d - PAGE + PLAND) pw0 = POLL(&PM.progress[0], d - PAGE + PLAND, 9);
                        if (pw1 < d - PAGE + PLAND) pw1 = POLL(&PM.progress[1], d - PAGE + PLAND, 9);
                        if (pw2 < d - PAGE + PLAND) pw2 = POLL(&PM.progress[2], d - PAGE + PLAND, 9);
                        if (pw3 < d - PAGE + PLAND) pw3 = POLL(&PM.progress[3], d - PAGE + PLAND, 9);
                        pg_d2 fxy[4];
                        double fm_[4];
                        FarAsk fa[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            fxy[k].x = NI_; fxy[k].y = NI_; fm_[k] = NI_;
                            fa[k].need = false; fa[k].boff = 0;
                            if (q_far[k]) fa[k] = far_ask(psc, d, f_age[k], f_p[k]);
                        }
                        far_fetch4(sc_out, fa[0], fa[1], fa[2], fa[3], fxy[0], fm_[0], fxy[1], fm_[1], fxy[2], fm_[2], fxy[3], fm_[3]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { q_fx[k] = fxy[k].x; q_fy[k] = fxy[k].y; q_fm[k] = fm_[k]; }
                    }
                }
            }
            return;
        }
    };
    prepare();
    while (q_d >= 0) {
        const int d = q_d;
        if (flag_load(&PM.abort_flag) != 0) break;
        ASTAMP(12);                                                // (the loop's top: bucket 0)
        // every compute wave has completed d-2 (or sleeps through it); a poll that ran into an abort returns "done" and
        // the wave runs to the end of its list on whatever is in the ring (reads stay inside the arena)
        if (pw0 < d - 2) pw0 = POLL(&PM.progress[0], d - 2, 9);
        if (pw1 < d - 2) pw1 = POLL(&PM.progress[1], d - 2, 9);
        if (pw2 < d - 2) pw2 = POLL(&PM.progress[2], d - 2, 9);
        if (pw3 < d - 2) pw3 = POLL(&PM.progress[3], d - 2, 9);
        if (q_cls == 2 && diags_ld < d) diags_ld = POLL(&PM.loaded[2], d, 5);
        ASTAMP(1);
        if (q_big) gen.scan(d, q_cls, q_lo, q_hi, q_mask, true);
        else if (q_fast) {
            if (lane < q_n) {
                const char *base = (const char *)&PM;
                const double *a0 = (const double *)(base + q_oa0), *a1 = (const double *)(base + q_oa1);
                const double *b0 = (const double *)(base + q_ob0), *b1 = (const double *)(base + q_ob1);
                double a0x = a0[PG_X], a0y = a0[PG_Y], a0m = a0[PG_M], a1x = a1[PG_X], a1y = a1[PG_Y], a1m = a1[PG_M];
                double b0x = b0[PG_X], b0y = b0[PG_Y], b0m = b0[PG_M], b1x = b1[PG_X], b1y = b1[PG_Y], b1m = b1[PG_M];
                if (q_cls == 2) {                                  // what came from L2 when the batch was prepared
                    a0x = q_far[0] ? q_fx[0] : a0x; a0y = q_far[0] ? q_fy[0] : a0y; a0m = q_far[0] ? q_fm[0] : a0m;
                    a1x = q_far[1] ? q_fx[1] : a1x; a1y = q_far[1] ? q_fy[1] : a1y; a1m = q_far[1] ? q_fm[1] : a1m;
                    b0x = q_far[2] ? q_fx[2] : b0x; b0y = q_far[2] ? q_fy[2] : b0y; b0m = q_far[2] ? q_fm[2] : b0m;
                    b1x = q_far[3] ? q_fx[3] : b1x; b1y = q_far[3] ? q_fy[3] : b1y; b1m = q_far[3] ? q_fm[3] : b1m;
                }
                const unsigned kk = q_left ? (1u << 4) : (1u << 18);
                const unsigned self = q_left ? PG_X : PG_Y, cross = q_left ? PG_Y : PG_X;
                double eg = NI_, em = NI_;
                unsigned pg = PG_BP_NONE, pm = PG_BP_NONE;
                const double s0 = q_left ? a0x : a0y, c0 = q_left ? a0y : a0x, s1 = q_left ? a1x : a1y, c1 = q_left ? a1y : a1x;
                // (the gap state's extension rate: the terminal one in the first / last column (X) or row (Y) -- row strips only)
                const double gs = !term_on ? ge : (q_left ? ((q_j == 0 || q_j == job->Ly - 1) ? (double)job->gE : ge)
                                                          : ((q_row == 0 || q_row == job->Lx - 1) ? (double)job->gE : ge));
                cand(s0 + gs, self, eg, pg); cand(c0 + go, cross, eg, pg); cand((a0m + ng) + go, PG_M, eg, pg);
                cand(s1 + gs, kk | self, eg, pg); cand(c1 + go, kk | cross, eg, pg); cand((a1m + ng) + go, kk | PG_M, eg, pg);
                cand((b0m + q_tM) + q_w0, q_e0 | PG_M, em, pm); cand((b0x + q_tX) + q_w0, q_e0 | PG_X, em, pm); cand((b0y + q_tX) + q_w0, q_e0 | PG_Y, em, pm);
                cand((b1m + q_tM) + q_w1, q_e1 | PG_M, em, pm); cand((b1x + q_tX) + q_w1, q_e1 | PG_X, em, pm); cand((b1y + q_tX) + q_w1, q_e1 | PG_Y, em, pm);
                const int win = (int)((pg >> (q_left ? 4 : 18)) & 127u);
                pg |= (unsigned)(q_adjs < 0 ? 0 : q_adjs) << (q_left ? 18 : 4);
                pg |= q_adjs < 0 ? PS_ONLY : ((eg > NI_ && win < q_adjs) ? PS_FIRST : 0u);
                const int stg = a, at = q_row & (PNT - 1);          // d % PST == d % PNA == a
                if (q_left) { PM.sx[stg][at] = eg; PM.spx[stg][at] = pg; } else { PM.sy[stg][at] = eg; PM.spy[stg][at] = pg; }
                PM.sM[stg][at] = em; PM.spm[stg][at] = pm;
            }
        }
        else if (q_n > 0) gen.compute(d, q_cls, q_mask, lane < q_n, q_row, q_j, q_kind, q_rL, q_cR, q_tM, q_tX);
        ASTAMP(2);
        flag_store(&PM.assist_done[a], d);
        ASTAMP(3);
#ifdef PG_PIPE_STATS
        if (as_on) ++as_n;
        as_prev = d;
#endif
        prepare();
#ifdef PG_PIPE_STATS
#ifdef PG_ASSIST_RUNS_ONLY
        as_on = q_d == as_prev + PNA;
#endif
#endif
    }
#ifdef PG_PIPE_STATS
    if (lane == 0 && 3 * (job->Lx + job->Ly) >= 4096) {
        PG_GLOBAL int *o = (PG_GLOBAL int *)job->trace + 3 * (job->Lx + job->Ly) - 1000 + 16 * a;
        o[0] = as_n;
        for (int k = 0; k < 12; ++k) o[1 + k] = (int)(as_t[k] >> 8);
    }
#endif
}


// ---- the compute waves' state that hot_run / wide_run share with the kernel's step() ----------------------------------
// Both are functions of their own (not inlined): the register allocation of their loops does not depend on what else the
// kernel contains, and the kernel's rare paths do not compete with them for SGPRs.  The state travels through this struct
// (in scratch memory) once per run, not per step.
struct WaveCtx {
    // constants of the wave
    const PgDevJob *job;
    cdesc8_p psc;
    gdouble_w sc_out;
    gu32_w bp_out;
    double go, ng, ge, tng2, tng1;
    int Lx, Ly, nd, S, sleep;
    int tid, wave, up, dn, bslot;
    unsigned flags;
    // what a run continues from and hands back
    int d, row;
    double px, py, pm, cx, cy, cm;           // this lane's cell on d-1; (row-1, j-1) on d-2
    pg_i4 ca, cb;                            // records of the columns d - row, d + 1 - row
    float smf;                               // model score of (row, d - row)
    pg_i8 dA;                                // descriptor of diagonal d
    int p_up, p_dn, ok_until, rows_ld, cols_ld, diags_ld, as0, as1, as2;
#ifdef PG_PIPE_STATS
    long long st_cls_t[5], st_poll_t[10], st_w[4];
    int st_cls_n[5], st_poll_n[10];
#endif
};
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ pg_i8 uniform_i8(const pg_i8 &v) {
    pg_i8 r;
    r.s0 = __builtin_amdgcn_readfirstlane(v.s0); r.s1 = __builtin_amdgcn_readfirstlane(v.s1); r.s2 = __builtin_amdgcn_readfirstlane(v.s2);
    r.s3 = __builtin_amdgcn_readfirstlane(v.s3); r.s4 = __builtin_amdgcn_readfirstlane(v.s4); r.s5 = __builtin_amdgcn_readfirstlane(v.s5);
    r.s6 = __builtin_amdgcn_readfirstlane(v.s6); r.s7 = __builtin_amdgcn_readfirstlane(v.s7);
    return r;
}
#define WCTX_IN(c) \
    const cdesc8_p psc = (cdesc8_p)uniform_u64((unsigned long long)(c).psc); \
    const gdouble_w sc_out = (gdouble_w)uniform_u64((unsigned long long)(c).sc_out); const gu32_w bp_out = (gu32_w)uniform_u64((unsigned long long)(c).bp_out); \
    const double go = (c).go, ng = (c).ng, ge = (c).ge, tng2 = (c).tng2, tng1 = (c).tng1; \
    const int Lx = __builtin_amdgcn_readfirstlane((c).Lx), Ly = __builtin_amdgcn_readfirstlane((c).Ly), nd = __builtin_amdgcn_readfirstlane((c).nd); \
    const int S = __builtin_amdgcn_readfirstlane((c).S), sleep = __builtin_amdgcn_readfirstlane((c).sleep); \
    const int tid = (c).tid, wave = __builtin_amdgcn_readfirstlane((c).wave), up = __builtin_amdgcn_readfirstlane((c).up), dn = __builtin_amdgcn_readfirstlane((c).dn), bslot = (c).bslot; \
    int d = __builtin_amdgcn_readfirstlane((c).d), row = (c).row; \
    int p_up = __builtin_amdgcn_readfirstlane((c).p_up), p_dn = __builtin_amdgcn_readfirstlane((c).p_dn), ok_until = __builtin_amdgcn_readfirstlane((c).ok_until); \
    int rows_ld = __builtin_amdgcn_readfirstlane((c).rows_ld), cols_ld = __builtin_amdgcn_readfirstlane((c).cols_ld), diags_ld = __builtin_amdgcn_readfirstlane((c).diags_ld); \
    int as0 = __builtin_amdgcn_readfirstlane((c).as0), as1 = __builtin_amdgcn_readfirstlane((c).as1), as2 = __builtin_amdgcn_readfirstlane((c).as2); \
    const double NI = neg_inf(); \
    (void)bp_out; (void)Lx; (void)Ly; (void)nd; (void)S; (void)up; (void)dn; (void)bslot; (void)as0; (void)as1; (void)as2; (void)diags_ld; (void)ok_until; (void)go; (void)ng; (void)ge; (void)tng2; (void)tng1; (void)NI
#define WCTX_OUT(c) \
    (c).d = d; (c).row = row; (c).p_up = p_up; (c).p_dn = p_dn; (c).ok_until = ok_until; (c).rows_ld = rows_ld; (c).cols_ld = cols_ld; \
    (c).diags_ld = diags_ld; (c).as0 = as0; (c).as1 = as1; (c).as2 = as2
#ifdef PG_PIPE_STATS
#define WCTX_STATS(c) long long (&st_cls_t)[5] = (c).st_cls_t; long long (&st_poll_t)[10] = (c).st_poll_t; int (&st_cls_n)[5] = (c).st_cls_n; int (&st_poll_n)[10] = (c).st_poll_n; long long (&st_w)[4] = (c).st_w; (void)st_w
#else
#define WCTX_STATS(c)
#endif

// every VGPR from v136 up and the SGPRs s36..s87 belong to the hand-scheduled loop (tools/gen_hot_asm.py has the plan)
#define PG_V8(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define PG_S8(a) "s" #a "0", "s" #a "1", "s" #a "2", "s" #a "3", "s" #a "4", "s" #a "5", "s" #a "6", "s" #a "7", "s" #a "8", "s" #a "9"
#define PG_HOT_CLOBBERS "v136", "v137", "v138", "v139", PG_V8(14), PG_V8(15), PG_V8(16), PG_V8(17), PG_V8(18), PG_V8(19), PG_V8(20), \
    PG_V8(21), PG_V8(22), PG_V8(23), PG_V8(24), "v250", "v251", "v252", "v253", "v254", "v255", \
    "s36", "s37", "s38", "s39", PG_S8(4), PG_S8(5), PG_S8(6), PG_S8(7), "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87"

// Far histories outside the hand-scheduled loop (round 5): the wide runs and the general steps append the cells of a line's row /
// column as the loop's writers do (tools/gen_hot_asm.py, hist_tail: the lane that holds the cell, in-band lanes only; a row's line
// is indexed by column, a column's by row), so that a history interval may cross them (dp_abi.hip, plan_far_hist).  gl / gr: the
// cell's site records (PR_SRC: bit 24, the line in bits 25-26).
__device__ __forceinline__ void hist_append(const pg_i4 &gl, const pg_i4 &gr, int r, int j, double bx, double by, double bm) {
    if (gl.x & PR_SRC) { double *h = &PM.hist[(gl.x >> 25) & 3][j & 63][0]; h[PG_X] = bx; h[PG_Y] = by; h[PG_M] = bm; }
    if (gr.x & PR_SRC) { double *h = &PM.hist[(gr.x >> 25) & 3][r & 63][0]; h[PG_X] = bx; h[PG_Y] = by; h[PG_M] = bm; }
}

// ---- hot run (model table in LDS): the diagonals from d on while their class is 0, 1 or 2 ----
// One iteration per diagonal.  Across iterations only the lane's cell of the previous diagonal (P), the shifted
// cell of the diagonal before (C) and the operand pipeline (row record, two column records, model score) live in
// registers; everything else is recomputed from the lane's row and the diagonal's descriptor, so there is no
// per-lane state to keep consistent: an out-of-band lane computes on whatever it holds and its results are
// replaced by -inf.  Class 1: the multi-edge cells are evaluated HERE, by the lanes that own them, in blocks that
// run only when an active lane of the wave needs them -- sites with one edge from the previous site and at most
// one more ("easy"; the host sends every other shape to the assist waves as class 2):
//   X: the previous-site edge's candidates from (row-1, j) as for a simple site, the other edge's from (row-kL, j)
//      in the ring; Y the same with (row, j-kR); M over the up to four (left edge, right edge) pairs, operands
//      C, (row-kL, j-1), (row-1, j-kR), (row-kL, j-kR).
// The reference walks the lists in order and replaces the incumbent on strict > only (VA:1328-1349, 1396-1433,
// basic_alignment.h:449-462): per edge / pair a first-wins maximum of its three candidates, then the groups
// combined with "equal: the one listed first" -- the same winner.  Class 2 (any other multi-edge shape, or operands
// that left the ring): what the diagonal's assist wave staged is merged as in step().
// STRIP (a row strip of a wide job, see strip_feeder): a lane keeps its row, so the y-gap state's extension rate is the
// lane's -- the first and the last row of the matrix extend at the terminal rate (VA:2116-2219: extY by row, extX by
// column) -- and a diagonal that holds a cell of the first or the last column (PG_STRIP_TERM in its class) takes the C++
// step below, which picks the x-gap state's rate per lane; the assembly loop leaves such a diagonal alone (its class is
// not 0..2 to it).
template <bool STRIP>
__device__ __noinline__ void hot_run(WaveCtx &C_) {
    WCTX_IN(C_);
    WCTX_STATS(C_);
    constexpr int CLS = STRIP ? 7 : 15;
    const bool term_on = STRIP && !(C_.flags & 1u);
    const double gE = (double)C_.job->gE;
    const double gey = in_vgpr((term_on && (row == 0 || row == Lx - 1)) ? gE : ge);
    double PX = C_.px, PY = C_.py, PMm = C_.pm, CX = C_.cx, CY = C_.cy, CM = C_.cm;
    const pg_i4 ca = C_.ca, cb = C_.cb;
    pg_i8 dA = uniform_i8(C_.dA);
    pg_i8 cur = dA;
    int sb = (d % PRK) * PROW_BYTES;                        // ring row of diagonal d, as a byte offset
    int hstg = d % PST;                                     // staging slot (and assist wave) of diagonal d
    pg_i4 rLc = PM.recL[row & (PRW - 1)];                   // (garbage while the row is beyond the staged ones: inactive)
    pg_i4 cR0 = ca, cR1 = cb;                               // records of the columns d - row and d + 1 - row
    // match terms of (row, d - row): tM = D(2*ng) + D(s), tX = D(0+ng) + D(s), ready in LDS (PM.tab2)
    int ti0 = ((rLc.x & 0xffff) + __umul24(cR0.x & 0xffff, S)) & 255;
    double tMc = PM.tab2[ti0][0], tXc = PM.tab2[ti0][1];
    const int tid24 = tid * 24, bpos24 = bslot * 24;
    const int null_off = (int)offsetof(PipeSmem, null_cell) - (int)offsetof(PipeSmem, sc);      // (ring_cell's offsets count from the ring's first byte)
    // LDS addresses the class 0 loop below works with
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_ring = (unsigned)(unsigned long long)(lds_char *)&PM.sc[0][0][0];
    const unsigned a_tid24 = lds_ring + (unsigned)tid24, a_bpos24 = lds_ring + (unsigned)bpos24;
    const unsigned a_recR = (unsigned)(unsigned long long)(lds_char *)&PM.recR[0], a_recL = (unsigned)(unsigned long long)(lds_char *)&PM.recL[0];
    const unsigned a_table = (unsigned)(unsigned long long)(lds_char *)&PM.tab2[0][0];
    const unsigned a_fdn = (unsigned)(unsigned long long)(lds_char *)&PM.progress[dn];
    const unsigned a_fup = (unsigned)(unsigned long long)(lds_char *)&PM.progress[up], a_fme = (unsigned)(unsigned long long)(lds_char *)&PM.progress[wave];
    const unsigned ni_hi = 0xfff00000u;
    const unsigned a_null = (unsigned)(unsigned long long)(lds_char *)&PM.null_cell[0];
    const unsigned a_asd = (unsigned)(unsigned long long)(lds_char *)&PM.assist_done[0];
    // far histories (tools/gen_hot_asm.py, hist_tail): the flag bytes' windows, the lines, the loader's descriptor window
    const unsigned a_hist = (unsigned)(unsigned long long)(lds_char *)&PM.hist[0][0][0], a_dring = (unsigned)(unsigned long long)(lds_char *)&PM.dring[0];
    // the third pass (tools/gen_hot_asm.py, third_pass): the sites' first edge, the edge windows
    const unsigned a_ebL = (unsigned)(unsigned long long)(lds_char *)&PM.ebL[0], a_ebR = (unsigned)(unsigned long long)(lds_char *)&PM.ebR[0];
    const unsigned a_esL = (unsigned)(unsigned long long)(lds_char *)&PM.esL[0], a_esR = (unsigned)(unsigned long long)(lds_char *)&PM.esR[0];
    const unsigned a_ewL = (unsigned)(unsigned long long)(lds_char *)&PM.ewL[0], a_ewR = (unsigned)(unsigned long long)(lds_char *)&PM.ewR[0];
    static_assert(PEC == 1024 && PRW == 512, "third_pass's address arithmetic");
    static_assert(sizeof(PM.hist[0]) == 0x600 && PDR == 128, "hist_tail's address arithmetic");
    const unsigned a_stx = (unsigned)(unsigned long long)(lds_char *)&PM.sx[0][tid], a_spx = (unsigned)(unsigned long long)(lds_char *)&PM.spx[0][tid];
    static_assert(offsetof(PipeSmem, sy) - offsetof(PipeSmem, sx) == 6144 && offsetof(PipeSmem, sM) - offsetof(PipeSmem, sx) == 12288 &&
                  offsetof(PipeSmem, spy) - offsetof(PipeSmem, spx) == 3072 && offsetof(PipeSmem, spm) - offsetof(PipeSmem, spx) == 6144 &&
                  PST == 3 && PNT == 256, "the class 2 merge of dp_pipe_hot.inc addresses the staging arrays by these strides");
    for (;;) {
#ifndef PG_NO_HOT_ASM
        if (!STRIP || !(cur.s4 & PG_STRIP_TERM)) {
            // ---- consecutive class 0 / 1 / 2 diagonals: hand-scheduled loop (tools/gen_hot_asm.py has the register plan;
            // the C++ step below states the same arithmetic) ----
            // It runs until a diagonal needs anything else -- another class, the end of the wave's
            // interval, a flag that has to be polled -- and leaves that diagonal untouched: d, the ring row, the lane's row
            // and cells come back; the operand pipeline is reloaded below.
            unsigned long long dptr = (unsigned long long)(psc + d);
            int colx = cR1.x & 0xffff, rowx = rLc.x & 0xffff;      // states of column d + 1 - row and of the row
            double tm_io = tMc, tx_io = tXc;
            const unsigned sc_lo = (unsigned)(unsigned long long)sc_out, sc_hi = (unsigned)((unsigned long long)sc_out >> 32);
            const int d_in = d;
            // the loop stops in front of the first diagonal it may not run without looking at the loader's flags or past the
            // wave's interval
            const int stop = __builtin_amdgcn_readfirstlane(sleep < ok_until + 1 ? sleep : ok_until + 1);
#ifdef PG_PIPE_STATS
            const long long st_t_in = __builtin_readcyclecounter();
#endif
            int k0 = 0, k1 = 0, k2 = 0;        // diagnostic builds (PG_HOT_EXP=k): waits for the upstream wave, looks at its flag, waits for the downstream wave
#define PG_HOT_OUTS [row] "+v"(row), [colx] "+v"(colx), [rowx] "+v"(rowx), [tm] "+v"(tm_io), [tx] "+v"(tx_io), \
                  [p0] "+v"(PX), [p1] "+v"(PY), [p2] "+v"(PMm), [c0] "+v"(CX), [c1] "+v"(CY), [c2] "+v"(CM), \
                  [d] "+s"(d), [sb] "+s"(sb), [pup] "+s"(p_up), [pdn] "+s"(p_dn), [dptr] "+s"(dptr), \
                  [k0] "+s"(k0), [k1] "+s"(k1), [k2] "+s"(k2)
#define PG_HOT_INS [ge] "v"(ge), [gey] "v"(gey), [go] "v"(go), [ng] "v"(ng), [nihi] "v"(ni_hi), [pihi] "v"(0x7ff00000u), \
                  [tid24] "v"(a_tid24), [bpos24] "v"(a_bpos24), [fup] "v"(a_fup), [fme] "v"(a_fme), \
                  [nulla] "v"(a_null), [tid] "v"(tid), [ringb] "s"(lds_ring), \
                  [asd] "s"(a_asd), [stx] "v"(a_stx), [spxa] "v"(a_spx), \
                  [stopm1] "s"(stop - 1), [S] "s"(S), [fdn] "v"(a_fdn), \
                  [bR] "s"(a_recR), [bL] "s"(a_recL), [bT] "s"(a_table), \
                  [sclo] "s"(sc_lo), [schi] "s"(sc_hi), [c24] "s"(0x3fffffffu), \
                  [histb] "v"(a_hist), [drb] "v"(a_dring), \
                  [ebl] "v"(a_ebL), [ebr] "v"(a_ebR), [esl] "v"(a_esL), [esr] "v"(a_esR), [ewl] "v"(a_ewL), [ewr] "v"(a_ewR)
            if constexpr (STRIP) {
                // (the strip's loop picks the x-gap state's rate per lane and step: tools/gen_hot_asm.py, STRIP)
                const double gE_ = term_on ? gE : ge;
                const int lym1 = __builtin_amdgcn_readfirstlane(Ly - 1);
                asm volatile(
#include "dp_pipe_hot_strip.inc"
                    : PG_HOT_OUTS
                    : PG_HOT_INS, [gel] "v"(__double2loint(ge)), [geh] "v"(__double2hiint(ge)), [gEl] "v"(__double2loint(gE_)),
                      [gEh] "v"(__double2hiint(gE_)), [lym1] "s"(lym1)
                    : "memory", "vcc", "scc",
                      PG_HOT_CLOBBERS);
            } else {
                asm volatile(
#include "dp_pipe_hot.inc"
                    : PG_HOT_OUTS
                    : PG_HOT_INS
                    : "memory", "vcc", "scc",
                      PG_HOT_CLOBBERS);
            }
#undef PG_HOT_OUTS
#undef PG_HOT_INS
            d = __builtin_amdgcn_readfirstlane(d);
            (void)d_in;
#ifdef PG_PIPE_STATS
            {   // one record per run of the loop (first diagonal, diagonal it stopped in front of, clock at entry and exit / 16,
                // waits for the upstream wave | 48-look exits << 16): upper half of the job's trace buffer, per wave
                const long long st_t_out = __builtin_readcyclecounter();
                const int n3 = 3 * (Lx + Ly), r0 = (n3 / 4 * 3 + 15) & ~15, cap = (n3 - 1200 - r0 - 16) / 16;
                if (cap > 0 && n3 >= 4096 && (tid & 63) == 0 && PG_STATS_MINE(C_.job, C_.flags)) {
                    PG_GLOBAL int *tb = (PG_GLOBAL int *)C_.job->trace;
                    const int slot = atomicAdd((int *)(tb + r0 + wave), 1);
                    if (slot < cap) {
                        PG_GLOBAL int *o = tb + r0 + 16 + 4 * (slot * 4 + wave);
                        o[0] = d_in; o[1] = __builtin_amdgcn_readfirstlane(d) | ((k0 >> 16) ? 0x40000000 : 0); o[2] = (int)(st_t_in >> 4); o[3] = (int)(st_t_out >> 4);
#ifdef PG_RUN_LOOKS                                                // (diagnostic: looks at the upstream flag / waits for the downstream wave in the place of the clocks' low bits)
                        o[2] = (int)((st_t_out - st_t_in) >> 4); o[3] = k1 | ((k2 & 0xffff) << 20);
#endif
                    }
                }
            }
            st_poll_n[9] += k0 & 0xffff; st_poll_n[6] += k1; st_poll_t[9] += (long long)(k2 & 0xffff) << 8; st_poll_n[5] += k0 >> 16; st_poll_t[5] += (long long)(k2 >> 16) << 8; st_poll_n[0] += 1; st_poll_t[0] += (long long)(d - d_in) << 8;
#else
            (void)k0; (void)k1; (void)k2;
#endif
            // the operand pipeline of diagonal d from the LDS windows, the descriptor from memory
            cur = psc[d];
            hstg = d % PST;
            rLc = PM.recL[row & (PRW - 1)];
            cR0 = PM.recR[(d - row) & (PRW - 1)];
            cR1 = PM.recR[(d + 1 - row) & (PRW - 1)];
            ti0 = ((rLc.x & 0xffff) + __umul24(cR0.x & 0xffff, S)) & 255;
            tMc = PM.tab2[ti0][0]; tXc = PM.tab2[ti0][1];
            if ((cur.s4 & CLS) > 2 || d >= sleep) { dA = cur; break; }
#ifdef PG_PIPE_STATS
            st_poll_n[7] += (__any(row <= cur.y + 1) ? 0 : 1);
#endif
            // The loop stopped at a diagonal it could run but for a flag: wait here (polls that sleep, spin limits that
            // end in an error status) for everything diagonal d needs, then go back into it -- the arithmetic below is the
            // loop's C++ rendering, compiled in with -DPG_NO_HOT_ASM only.
            {
                const int lo = cur.x, hi = cur.y;
                if (d > ok_until) {
                    int need = hi + 3 < Lx - 1 ? hi + 3 : Lx - 1;
                    // (the loader's progress read afresh: a margin worked out from a stale value shrinks by half from one look to the next)
                    rows_ld = flag_load(&PM.loaded[0]); cols_ld = flag_load(&PM.loaded[1]);
                    if (rows_ld <= need) rows_ld = POLLX(&PM.loaded[0], need + 1, 1);
                    int margin = rows_ld >= Lx ? nd : rows_ld - 1 - need;
                    need = d + 2 - lo < Ly - 1 ? d + 2 - lo : Ly - 1;
                    if (cols_ld <= need) cols_ld = POLLX(&PM.loaded[1], need + 1, 2);
                    const int mc = cols_ld >= Ly ? nd : cols_ld - 1 - need;
                    margin = mc < margin ? mc : margin;
                    ok_until = d + margin;
                }
                if (cur.s7 > p_dn) p_dn = POLLX(&PM.progress[dn], cur.s7, 3);
                if (d - 1 > p_up && __any(row <= hi + 1)) p_up = POLLX(&PM.progress[up], d - 1, 4);
                if ((cur.s4 & CLS) == 2) {
                    if (hstg == 0) { if (as0 < d) as0 = POLLX(&PM.assist_done[0], d, 8); }
                    else if (hstg == 1) { if (as1 < d) as1 = POLLX(&PM.assist_done[1], d, 8); }
                    else { if (as2 < d) as2 = POLLX(&PM.assist_done[2], d, 8); }
                }
                // after an abort (a wait somewhere ran into its spin limit) the wave leaves: the kernel's loop sees the flag
                if (flag_load(&PM.abort_flag) != 0) { dA = cur; break; }
            }
            continue;
        }
#endif
        const int lo = cur.x, hi = cur.y, cls = cur.s4 & CLS;
        // (this C++ rendering of the step knows nothing of the far histories: a job planned with them -- PAGAN_DP_HIST is not 0 --
        //  must run the hand-scheduled loop; a -DPG_NO_HOT_ASM build that meets one says so instead of computing something else)
        if (!STRIP && (cur.s4 & 32)) { if (flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, PTAG(13)); dA = cur; break; }
#ifdef PG_PIPE_STATS
        const long long st_step0 = __builtin_readcyclecounter();
        const bool st_has = __any(row <= hi && row >= lo);
#endif
        // ---- flow control: flags are read only when the cached values stop covering this step (see step()) ----
        if (d > ok_until) {
            int need = hi + 3 < Lx - 1 ? hi + 3 : Lx - 1;
            // (the loader's progress read afresh: a margin worked out from a stale value shrinks by half from one look to the next)
            rows_ld = flag_load(&PM.loaded[0]); cols_ld = flag_load(&PM.loaded[1]);
            if (rows_ld <= need) rows_ld = POLLX(&PM.loaded[0], need + 1, 1);
            int margin = rows_ld >= Lx ? nd : rows_ld - 1 - need;
            need = d + 2 - lo < Ly - 1 ? d + 2 - lo : Ly - 1;
            if (cols_ld <= need) cols_ld = POLLX(&PM.loaded[1], need + 1, 2);
            const int mc = cols_ld >= Ly ? nd : cols_ld - 1 - need;
            margin = mc < margin ? mc : margin;
            ok_until = d + margin;
        }
        if (cur.s7 > p_dn) p_dn = POLLX(&PM.progress[dn], cur.s7, 3);
        // ---- (row-1, j) on d-1: lane-1's registers; lane 0 takes the upstream wave's lane 63 from the ring (what it
        // reads is used only if lane 0's row is in the band, and then the upstream wave is waited for).  The flag is
        // read in the same LDS batch, ahead of the cell (LDS executes in order): a follower one step behind its
        // upstream wave pays no round trip of its own for it ----
        const int sb1 = sb == 0 ? PRING_BYTES - PROW_BYTES : sb - PROW_BYTES;
        double AX, AY, AM;
        const int upf = flag_peek(&PM.progress[up]);
        ring_cell(sb1 + bpos24, AX, AY, AM);
        if (d - 1 > p_up) {
            const int seen = __builtin_amdgcn_readfirstlane(upf);
            p_up = seen > p_up ? seen : p_up;
            if (d - 1 > p_up && __any(row <= hi + 1)) {
                p_up = POLLX(&PM.progress[up], d - 1, 4);
                ring_cell(sb1 + bpos24, AX, AY, AM);
            }
        }
        // ---- row hand-over: a lane whose row left the band takes the next one of its residue (far below the band) ----
        row += row < lo ? PNT : 0;
        const bool active = row <= hi;
        const double tM = tMc, tX = tXc;
        double bx, by, bm;
        unsigned px, py, pm;
        // candidates that do not need the shift first: the LDS read above is in flight
        by = fmax3_from(PY + gey, PX + go, (PMm + ng) + go, PG_Y | PG_BP_ADJR, PG_X | PG_BP_ADJR, PG_M | PG_BP_ADJR, py);
        // the x-gap state's rate: the terminal one in the first and the last column
        const int jcol = d - row;
        const double gex = (term_on && (jcol == 0 || jcol == Ly - 1)) ? gE : ge;
        // (a class 1 diagonal none of whose multi-edge cells belongs to this wave is a class 0 diagonal to it)
        const bool own_multi = cls == 1 && __any(active && !(rLc.x & cR0.x & PR_SIMPLE));
        if (!own_multi) {
            bm = fmax3_from(CM + tM, CX + tX, CY + tX, PG_M | PG_BP_ADJL | PG_BP_ADJR, PG_X | PG_BP_ADJL | PG_BP_ADJR,
                            PG_Y | PG_BP_ADJL | PG_BP_ADJR, pm);
        }
        AX = dpp_shr1(PX, AX); AY = dpp_shr1(PY, AY); AM = dpp_shr1(PMm, AM);
        // next descriptor: requested after the step's LDS wait (an s_waitcnt on LDS data also waits for scalar
        // loads in flight); the array carries one entry of padding
        pg_i8 nxt;
        {
            unsigned long long pv = (unsigned long long)(psc + d + 1);
            asm volatile("" : "+s"(pv) : "v"(AX), "v"(AY), "v"(AM));
            nxt = *(cdesc8_p)pv;
        }
        bx = fmax3_from(AX + gex, AY + go, (AM + ng) + go, PG_X | PG_BP_ADJL, PG_Y | PG_BP_ADJL, PG_M | PG_BP_ADJL, px);
        if (own_multi) {
            // ================= class 1: the lanes' own multi-edge cells =================
            const bool l2 = active && ((rLc.x >> PR_NE_SHIFT) & 127) == 2, r2 = active && ((cR0.x >> PR_NE_SHIFT) & 127) == 2;
            const bool lA2 = (rLc.y & 0xffff) != 1, rA2 = (cR0.y & 0xffff) != 1;      // (two edges:) the previous-site edge is listed second
            const double lw0 = (double)__int_as_float(rLc.z), lw1 = (double)__int_as_float(rLc.w);
            const double rw0 = (double)__int_as_float(cR0.z), rw1 = (double)__int_as_float(cR0.w);
            const bool lS = l2 && lA2, rS = r2 && rA2;                                 // the other edge comes first in the list
            const double lwA = lS ? lw1 : lw0, lwS = lS ? lw0 : lw1, rwA = rS ? rw1 : rw0, rwS = rS ? rw0 : rw1;
            const unsigned lbA = lS ? 1u << 4 : 0u, lbS = lS ? 0u : 1u << 4;           // list slots in a back-pointer
            const unsigned rbA = rS ? 1u << 18 : 0u, rbS = rS ? 0u : 1u << 18;
            px |= lbA; py |= rbA;
            bm = fmax3_from(((CM + tM) + lwA) + rwA, ((CX + tX) + lwA) + rwA, ((CY + tX) + lwA) + rwA, PG_M, PG_X, PG_Y, pm);
            pm |= PG_BP_ADJL | PG_BP_ADJR | lbA | rbA;
            const int kL = lA2 ? (rLc.y & 0xffff) : (int)((unsigned)rLc.y >> 16);
            const int kR = rA2 ? (cR0.y & 0xffff) : (int)((unsigned)cR0.y >> 16);
            const int posL = ((tid - kL) & (PNT - 1)) * 24;
            const bool anyL = __any(l2), anyR = __any(r2);
            if (anyR) {
                // the right site's other edge: Y from (row, j-kR), the pair (previous-site left edge, it) from (row-1, j-kR)
                double ux, uy, um, vx, vy, vm;
                ring_cell(r2 ? ring_back(sb, kR) + tid24 : null_off, ux, uy, um);
                ring_cell(r2 ? ring_back(sb, kR + 1) + bpos24 : null_off, vx, vy, vm);
                unsigned f;
                const double ys = fmax3_from(uy + gey, ux + go, (um + ng) + go, PG_Y, PG_X, PG_M, f);
                take_better(by, py, ys, f | rbS, rS);
                const double ms = fmax3_from(((vm + tM) + lwA) + rwS, ((vx + tX) + lwA) + rwS, ((vy + tX) + lwA) + rwS, PG_M, PG_X, PG_Y, f);
                take_better(bm, pm, ms, f | PG_BP_ADJL | lbA | rbS, rS);
            }
            if (anyL) {
                // the left site's other edge: X from (row-kL, j), the pair (it, previous-site right edge) from (row-kL, j-1)
                double ux, uy, um, vx, vy, vm;
                ring_cell(l2 ? ring_back(sb, kL) + posL : null_off, ux, uy, um);
                ring_cell(l2 ? ring_back(sb, kL + 1) + posL : null_off, vx, vy, vm);
                unsigned f, f2;
                const double xs = fmax3_from(ux + gex, uy + go, (um + ng) + go, PG_X, PG_Y, PG_M, f);
                take_better(bx, px, xs, f | lbS, lS);
                double m2 = fmax3_from(((vm + tM) + lwS) + rwA, ((vx + tX) + lwS) + rwA, ((vy + tX) + lwS) + rwA, PG_M, PG_X, PG_Y, f2);
                unsigned p2 = f2 | PG_BP_ADJR | lbS | rbA;
                if (__any(l2 && r2)) {
                    // both sites have another edge: the pair of the two, from (row-kL, j-kR)
                    double wx, wy, wm;
                    ring_cell((l2 && r2) ? ring_back(sb, kL + kR) + posL : null_off, wx, wy, wm);
                    const double m3 = fmax3_from(((wm + tM) + lwS) + rwS, ((wx + tX) + lwS) + rwS, ((wy + tX) + lwS) + rwS, PG_M, PG_X, PG_Y, f);
                    take_better(m2, p2, m3, f | lbS | rbS, rS);
                }
                take_better(bm, pm, m2, p2, lS);
            }
        }
#ifndef PG_X_NOC2
        else if (cls == 2) {
            // ================= class 2: merge what the assist wave of this diagonal staged (see step()) =================
            if (hstg == 0) { if (as0 < d) as0 = POLLX(&PM.assist_done[0], d, 8); }
            else if (hstg == 1) { if (as1 < d) as1 = POLLX(&PM.assist_done[1], d, 8); }
            else { if (as2 < d) as2 = POLLX(&PM.assist_done[2], d, 8); }
            const double ex = PM.sx[hstg][tid], ey = PM.sy[hstg][tid], em = PM.sM[hstg][tid];
            const unsigned sfx = PM.spx[hstg][tid], sfy = PM.spy[hstg][tid], sfm = PG_BP_NONE;      // (no staged M back-pointer: pg_backptr derives them all)
            const bool msL = !(rLc.x & PR_SIMPLE), msR = !(cR0.x & PR_SIMPLE);
            const bool tkx = msL && ((sfx & PS_ONLY) || ex > bx || (ex == bx && (sfx & PS_FIRST)));
            const bool tky = msR && ((sfy & PS_ONLY) || ey > by || (ey == by && (sfy & PS_FIRST)));
            px |= msL ? ((sfx >> 18) & 127u) << 4 : 0u;            // the previous-site edge's slot in the left list
            py |= msR ? ((sfy >> 4) & 127u) << 18 : 0u;            // ... in the right list
            bx = tkx ? ex : bx;  px = tkx ? (sfx & 0x3ffffu) : px;
            by = tky ? ey : by;  py = tky ? (sfy & 0x01fc000fu) : py;
            const bool tkm = msL || msR;
            bm = tkm ? em : bm;  pm = tkm ? sfm : pm;
        }
#endif
        // ---- results: -inf outside the band; a state that stayed -inf has no back-pointer ----
        bx = active ? bx : NI; by = active ? by : NI; bm = active ? bm : NI;
        px = bx > NI ? px : PG_BP_NONE; py = by > NI ? py : PG_BP_NONE; pm = bm > NI ? pm : PG_BP_NONE;
        {
            double *o = (double *)((char *)&PM.sc[0][0][0] + sb + tid24);
            o[PG_X] = bx; o[PG_Y] = by; o[PG_M] = bm;
        }
        if (active) {
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            const long long soff = ((long long)cur.w << 32) | (unsigned)cur.z;     // 24 * first cell of the diagonal
            PG_GLOBAL char *srow = (PG_GLOBAL char *)sc_out + soff;
            PG_GLOBAL char *brow = (PG_GLOBAL char *)bp_out + (soff >> 1);
            const unsigned off = (unsigned)(row - lo);
            const unsigned off12 = __umul24(off, 12u);
            store_scores<STRIP>(srow + 2u * off12, bx, by, bm);
            u3 b3; b3.x = px; b3.y = py; b3.z = pm;
            *(PG_GLOBAL u3 *)(brow + off12) = b3;
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // all but the last 8 steps' stores have retired (far reads rely on it)
#ifdef PG_X_OLDFLAG
        flag_store(&PM.progress[wave], d);
#else
        flag_store_inorder(&PM.progress[wave], d);
#endif
        // ---- operand pipeline for the next steps: one LDS wait per step, at its top ----
        const pg_i4 cR2 = PM.recR[(d + 2 - row) & (PRW - 1)];
        const int ti1 = ((rLc.x & 0xffff) + __umul24(cR1.x & 0xffff, S)) & 255;
        const double tM1 = PM.tab2[ti1][0], tX1 = PM.tab2[ti1][1];
        const pg_i4 rLn = PM.recL[row & (PRW - 1)];
        PX = bx; PY = by; PMm = bm;
        CX = AX; CY = AY; CM = AM;
#ifdef PG_PIPE_STATS
        if (st_has) {
            const long long dt = __builtin_readcyclecounter() - st_step0;
            for (int c = 0; c < 5; ++c) if (c == cls) { st_cls_t[c] += dt; ++st_cls_n[c]; }
        }
#endif
        ++d;
        sb = sb + PROW_BYTES == PRING_BYTES ? 0 : sb + PROW_BYTES;
        hstg = hstg + 1 == PST ? 0 : hstg + 1;
        cR0 = cR1; cR1 = cR2; tMc = tM1; tXc = tX1; rLc = rLn;
        if ((nxt.s4 & CLS) > 2 || d >= sleep) { dA = nxt; break; }       // the next diagonal is not class 0..2, or the wave's interval ends
        cur = nxt;
    }
    // hand the state back to step(): the column records and model score of diagonal d (the row records are reloaded there,
    // nothing of lane 0's operand is prefetched)
    C_.ca = cR0; C_.cb = cR1;
    C_.px = PX; C_.py = PY; C_.pm = PMm; C_.cx = CX; C_.cy = CY; C_.cm = CM;
    C_.dA = dA;
    WCTX_OUT(C_);
}


// ---- wide run (model table in LDS): consecutive class 4 diagonals, 242 .. PG_PIPE_WINDOW cells ----
// A lane holds the two smallest rows of its residue that are not below the band: rowA = lo + ((tid - lo) & 255) and
// rowB = rowA + 256 (a wide diagonal has at most 512 - 160 cells, so no lane ever has a third).  As in hot_run the
// hand-over stays in REGISTERS (round 5; before, every operand of every cell -- the three neighbours included -- was
// read back from LDS or L2, and a step cost 15 k cycles):
//   - per set the lane keeps its cell of the previous diagonal (P: (row, j-1)) and the shifted cell of the diagonal before
//     (C: (row-1, j-1)); (row-1, j) comes from lane T-1 by one DPP shift per set -- lane T-1's set A holds rowA(T) - 1
//     unless rowA(T) is the band's first row of the previous diagonal, in which case it holds rowB(T) - 1 and the lane's
//     set A has no neighbour in the band --, lane 0 reads the upstream wave's lane 63 out of the wide ring;
//   - a lane whose rowA fell below the band moves its set B into A (cells and all) AFTER the shift, as the narrow loop
//     does, so that the last cell of a row that leaves the band is still handed down;
//   - the operands of the other edges of "easy" multi-edge sites come from the WIDE RING -- the ring's memory as PWK rows
//     of 512 positions (row % 512), written by every lane for both its rows, -inf outside the band -- when their
//     diagonal belongs to this run and is at most PWAGE back, and from L2 otherwise (diagonals before the run: everything
//     landed at the rendezvous on entry; older diagonals of the run: every wave keeps all but its last three steps' stores
//     retired and no wave is more than two steps behind another: an L2 operand is at least PWAGE + 1 = 6 diagonals old);
//     anything else (first / last rows and columns, sites with more than two edges) goes through cell_any_t with one
//     fetch per operand;
//   - the first step of a run takes its registers from L2 (one round trip per run).
// The four waves move in lock step through flags: the wave above has completed d-1, the wave below d-2 (ring row reuse:
// with those two every wave has completed d-2, and nobody reads the row of d-7 after its step d-2).
__device__ __noinline__ void wide_run(WaveCtx &C_) {
    WCTX_IN(C_);
    WCTX_STATS(C_);
    const PgDevJob *__restrict__ job = (const PgDevJob *)uniform_u64((unsigned long long)C_.job);
    const unsigned flags_ = __builtin_amdgcn_readfirstlane(C_.flags);
    const bool no_terminal_edges = flags_ & 1u, reduced_terminal = !(flags_ & 2u);
    pg_i8 dA = uniform_i8(C_.dA);
    const View J = load_view(job);
    const int d0 = d;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(&PM.arrived[wave], d0);
    for (int w = 0; w < PNW; ++w) POLLX(&PM.arrived[w], d0, 7);
    p_up = d0 - 1 > p_up ? d0 - 1 : p_up;
    p_dn = d0 - 1 > p_dn ? d0 - 1 : p_dn;
    const double NIw = neg_inf();
    const int lane = tid & 63;
    // The run's diagonals d0 .. run_end - 1: found here with vector loads of the descriptors' class words (64 per round trip);
    // inside the loop a diagonal's rows and score offset come from the loader's LDS window (PM.dring) -- a scalar load per step
    // shares its counter with the LDS operations and returns out of order, so the step's first LDS wait behind it waited for
    // memory (~2 k cycles of every step)
    int run_end = d0 + 1;
    for (;;) {
        const int t = run_end + lane;
        const int c4 = ((PG_GLOBAL const int *)psc)[8 * (t < nd ? t : nd) + 4] & 15;      // (the array carries one entry of padding: class 0)
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(t >= nd || t >= sleep || c4 != 4);
        if (stop != 0) { run_end += __builtin_ctzll(stop); break; }
        run_end += 64;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pg_i4 cur = {dA.x, dA.y, dA.z, dA.w};                          // rows lo .. hi of the diagonal, byte offset of its first score
    // ---- the lane's two sets: rows, P = (row, j-1) on d-1, C = (row-1, j-1) on d-2, U = (row-1, j) on d-1 (first step only:
    // afterwards U is the shift's result) -- from L2, everything before the run has landed ----
    int wrow[2];
    double Px[2], Py[2], Pm[2], Cx[2], Cy[2], Cm[2], Ux[2], Uy[2], Um[2];
    wrow[0] = cur.x + ((tid - cur.x) & (PNT - 1)); wrow[1] = wrow[0] + PNT;
    if (diags_ld < d0) diags_ld = POLLX(&PM.loaded[2], d0, 5);
    {
        FarAsk fa[8];
        pg_d2 xy[8];
        double m[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            fa[3 * q + 0] = far_ask(psc, d0, 1, wrow[q]);          // P
            fa[3 * q + 1] = far_ask(psc, d0, 2, wrow[q] - 1);      // C
            fa[3 * q + 2] = far_ask(psc, d0, 1, wrow[q] - 1);      // U
        }
        fa[6].need = false; fa[6].boff = 0; fa[7] = fa[6];
#pragma unroll
        for (int k = 0; k < 8; ++k) { xy[k].x = NIw; xy[k].y = NIw; m[k] = NIw; }
        far_fetch8(sc_out, fa, xy, m);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            Px[q] = xy[3 * q].x; Py[q] = xy[3 * q].y; Pm[q] = m[3 * q];
            Cx[q] = xy[3 * q + 1].x; Cy[q] = xy[3 * q + 1].y; Cm[q] = m[3 * q + 1];
            Ux[q] = xy[3 * q + 2].x; Uy[q] = xy[3 * q + 2].y; Um[q] = m[3 * q + 2];
        }
    }
    // the run's wide-ring geometry (descriptor word 4, bit 4 of a class 4 diagonal: some diagonal of the run exceeds PG_PIPE_WINDOW_A cells)
    const bool geo_b = (dA.s4 & 16) != 0;
    const int wpos_n = geo_b ? PWPOS_B : PWPOS_A, wk = geo_b ? PWK_B : PWK_A, wage = wk - 2, wrow_bytes = wpos_n * 24;
    // position of a row in a wide-ring row: (row - wbase) % positions, wbase a multiple-of-positions step behind the band's first row
    int wbase = cur.x - 32;
    auto wpos = [&](int r_) { int q_ = r_ - wbase; q_ -= q_ >= wpos_n ? wpos_n : 0; q_ -= q_ >= wpos_n ? wpos_n : 0; return q_; };
    int n_hist1 = 0, n_hist2 = 0;                                  // stores issued in the previous step / the one before
    int lo_prev = cur.x;                                           // first row of the previous diagonal's band (the sets' rows are relative to it)
    for (;;) {
        const int lo = cur.x, hi = cur.y;
        if (lo - 2 * wage - 4 - wbase >= wpos_n) wbase += wpos_n;      // (every row a step touches lies in [wbase, wbase + 3 * 384))
#ifdef PG_PIPE_STATS
        const long long st_step0 = __builtin_readcyclecounter();
#endif
        {   // records of the diagonal's rows and columns, descriptors of every earlier diagonal
            const int nr_ = hi + 1 < Lx ? hi + 1 : Lx, nc_ = d - lo + 1 < Ly ? d - lo + 1 : Ly;
            if (rows_ld < nr_) rows_ld = POLLX(&PM.loaded[0], nr_, 1);
            if (cols_ld < nc_) cols_ld = POLLX(&PM.loaded[1], nc_, 2);
            if (diags_ld < d) diags_ld = POLLX(&PM.loaded[2], d, 5);
        }
        // (inline polls: in lock step a neighbour is, as a rule, a fraction of a step away)
        if (p_up < d - 1) p_up = __builtin_amdgcn_readfirstlane(POLL(&PM.progress[up], d - 1, 4));
        if (p_dn < d - 2) p_dn = __builtin_amdgcn_readfirstlane(POLL(&PM.progress[dn], d - 2, 3));
#ifdef PG_PIPE_STATS
        const long long st_t1 = __builtin_readcyclecounter();
#endif
        // the next diagonal's descriptor: staged by the loader (the window runs PLOOK diagonals ahead of the slowest wave)
        if (diags_ld < d + 2 && d + 1 < nd) diags_ld = POLLX(&PM.loaded[2], d + 2, 5);
        const pg_i4 nxt = PM.dring[(d + 1) & (PDR - 1)];
        const int amax = d - d0 < wage ? d - d0 : wage;               // ages 1 .. amax are in the wide ring
        const int wsb = (d % wk) * wrow_bytes;
        const long long soff = ((long long)cur.w << 32) | (unsigned)cur.z;
        PG_GLOBAL char *srow = (PG_GLOBAL char *)sc_out + soff;
#ifdef PG_EXP_WIDE_SKIP                                             // timing experiment (WRONG RESULTS): a wide step is its flags and nothing else
        if (true) {
            flag_store(&PM.progress[wave], d);
            ++d; lo_prev = lo;
            if (d >= run_end || flag_load(&PM.abort_flag) != 0) break;
            cur.x = __builtin_amdgcn_readfirstlane(nxt.x); cur.y = __builtin_amdgcn_readfirstlane(nxt.y);
            cur.z = __builtin_amdgcn_readfirstlane(nxt.z); cur.w = __builtin_amdgcn_readfirstlane(nxt.w);
            continue;
        }
#endif
        if (d > d0) {
            // ---- (row-1, j) on d-1: lane T-1's registers, lane 0 from the wide ring (the upstream wave completed d-1) ----
            int rb1 = wsb - wrow_bytes;
            rb1 += rb1 < 0 ? wk * wrow_bytes : 0;
            double ax[2], ay[2], am[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double *c = (const double *)((const char *)&PM.sc[0][0][0] + rb1 + wpos(wrow[q] - 1) * 24);
                ax[q] = dpp_shr1(Px[q], c[PG_X]); ay[q] = dpp_shr1(Py[q], c[PG_Y]); am[q] = dpp_shr1(Pm[q], c[PG_M]);
            }
            // lane T-1's set A holds rowA(T) - 1, its set B rowB(T) - 1 -- unless rowA(T) was the band's first row: then lane T-1's
            // set A holds rowB(T) - 1 (lane 0 read by position: nothing to sort out)
            // (the band's first row has no row above it in the band -- for lane 0 as well: the position it read is, with 384
            //  positions to a row, the alias of a row 384 further on, which ANOTHER wave writes, not the upstream one it waited for)
            const bool top = wrow[0] == lo_prev, first = top && lane != 0;
            Ux[0] = top ? NIw : ax[0]; Uy[0] = top ? NIw : ay[0]; Um[0] = top ? NIw : am[0];
            Ux[1] = first ? ax[0] : ax[1]; Uy[1] = first ? ay[0] : ay[1]; Um[1] = first ? am[0] : am[1];
            // ---- row hand-over: set B becomes set A, the new set B (512 rows on: far below the band) starts from -inf ----
            if (wrow[0] < lo) {
                wrow[0] = wrow[1]; wrow[1] += PNT;
                Px[0] = Px[1]; Py[0] = Py[1]; Pm[0] = Pm[1]; Cx[0] = Cx[1]; Cy[0] = Cy[1]; Cm[0] = Cm[1];
                Ux[0] = Ux[1]; Uy[0] = Uy[1]; Um[0] = Um[1];
                Px[1] = NIw; Py[1] = NIw; Pm[1] = NIw; Cx[1] = NIw; Cy[1] = NIw; Cm[1] = NIw; Ux[1] = NIw; Uy[1] = NIw; Um[1] = NIw;
            }
        }
        // one operand cell (p, d - age): read from the wide ring, or -- not there -- what to ask L2 for (the caller fetches
        // all of a step's requests in one statement); -inf outside the band
        auto wcell = [&](bool need, int age, int p_, pg_d2 &xy, double &m) -> FarAsk {
            FarAsk f = {false, 0};
            xy.x = NIw; xy.y = NIw; m = NIw;
            // (a row above the previous diagonal's band: -inf without looking -- its position in the wide ring is the alias of a row
            //  far below the band, which a wave other than the one above this one writes, and that wave need not have completed d-1)
            // (not on the run's first diagonal: lo_prev is the previous diagonal's first row from the second step on, and the first
            //  step's operands all come from L2, where the descriptor says what lies in the band)
            if (!need || (age == 1 && d > d0 && p_ < lo_prev)) return f;
            if (age <= amax) {
                int rb = wsb - age * wrow_bytes;
                rb += rb < 0 ? wk * wrow_bytes : 0;
                const double *c = (const double *)((const char *)&PM.sc[0][0][0] + rb + wpos(p_) * 24);
                xy.x = c[PG_X]; xy.y = c[PG_Y]; m = c[PG_M];
                return f;
            }
            return far_ask(psc, d, age, p_);
        };
        // Both sets in phases, so that a step pays at most ONE L2 round trip for its batched cells: read / request the operands
        // of the other edges, wait once (only if some lane asked L2: a wait also covers the wave's stores in flight), then the
        // arithmetic; cells outside the batch (general rules) fetch one operand at a time afterwards.
        int kind[2];                                            // 0 outside the band, 1 batched, 2 general rules
        pg_i4 gl[2], gr[2];
        pg_d2 o_xy[2][8];
        double o_m[2][8];
        bool l2q[2], r2q[2], lSq[2], rSq[2];
        FarAsk o_f[2][8];
        bool any_far = false;
        int n_now = 0;                                          // stores this wave issues in this step
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wrow[q], j = d - r;
            kind[q] = 0;
            l2q[q] = r2q[q] = lSq[q] = rSq[q] = false;
            int kL = 0, kR = 0;
            gl[q] = pg_i4{0, 0, 0, 0}; gr[q] = pg_i4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 8; ++t) { o_f[q][t].need = false; o_f[q][t].boff = 0; o_xy[q][t].x = NIw; o_xy[q][t].y = NIw; o_m[q][t] = NIw; }
            if (!__any(r <= hi)) continue;                      // none of the wave's lanes has a (second) row on this diagonal
            if (r <= hi) {
                gl[q] = PM.recL[r & (PRW - 1)]; gr[q] = PM.recR[j & (PRW - 1)];
                const int nl = (gl[q].x >> PR_NE_SHIFT) & 127, nr = (gr[q].x >> PR_NE_SHIFT) & 127;
                const int dl0 = gl[q].y & 0xffff, dl1 = (int)((unsigned)gl[q].y >> 16), dr0 = gr[q].y & 0xffff, dr1 = (int)((unsigned)gr[q].y >> 16);
                const bool easyL = nl == 1 ? dl0 == 1 : (nl == 2 && (dl0 == 1) != (dl1 == 1));
                const bool easyR = nr == 1 ? dr0 == 1 : (nr == 2 && (dr0 == 1) != (dr1 == 1));
                const bool l2 = nl == 2, r2 = nr == 2;
                const bool lS = l2 && dl0 != 1, rS = r2 && dr0 != 1;                 // the other edge is listed first
                kL = lS ? dl0 : dl1; kR = rS ? dr0 : dr1;
                // interior, and no edge in reach starts at site 0 (where the gap-open term differs)
                const bool inner = r >= 2 && r <= Lx - 2 && j >= 2 && j <= Ly - 2 && (!l2 || r - kL >= 1) && (!r2 || j - kR >= 1);
                kind[q] = (easyL && easyR && inner) ? 1 : 2;
                if (kind[q] == 1) { l2q[q] = l2; r2q[q] = r2; lSq[q] = lS; rSq[q] = rS; }
            }
            // the batched cells' other-edge operands.  Wave-uniform shortcuts: an operand is skipped when no lane has it, and
            // while every lane's lies in the wide ring the reads are plain LDS reads (no L2 path, no branches)
            const bool b1 = kind[q] == 1, l2 = l2q[q], r2 = r2q[q];
            if (!__any(b1 && (l2 || r2))) continue;
            auto rd = [&](bool need, int age, int p_, pg_d2 &xy, double &m) {
                int rb = wsb - age * wrow_bytes;
                rb += rb < 0 ? wk * wrow_bytes : 0;
                const int off = need ? rb + wpos(p_) * 24 : (int)offsetof(PipeSmem, null_cell) - (int)offsetof(PipeSmem, sc);
                const double *c = (const double *)((const char *)&PM.sc[0][0][0] + off);
                xy.x = c[PG_X]; xy.y = c[PG_Y]; m = c[PG_M];
            };
            auto fetch = [&](int t, bool need, int age, int p_) {
                if (!__any(need)) return;
                if (!__any(need && age > amax)) rd(need, age, p_, o_xy[q][t], o_m[q][t]);
                else { o_f[q][t] = wcell(need, age, p_, o_xy[q][t], o_m[q][t]); any_far = true; }     // (-inf for the lanes without it)
            };
            fetch(3, b1 && l2, kL, r - kL);
            fetch(4, b1 && l2, kL + 1, r - kL);
            fetch(5, b1 && r2, kR, r);
            fetch(6, b1 && r2, kR + 1, r - 1);
            fetch(7, b1 && l2 && r2, kL + kR, r - kL);
        }
#ifdef PG_PIPE_STATS
        const long long st_t2 = __builtin_readcyclecounter();
#endif
        // what the wave asked L2 for, both sets: one statement per set, requests and wait (it does nothing if no lane asked)
        if (any_far) {
            far_fetch8(sc_out, o_f[0], o_xy[0], o_m[0]);
            far_fetch8(sc_out, o_f[1], o_xy[1], o_m[1]);
        }
#ifdef PG_PIPE_STATS
        const long long st_t3 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wrow[q], j = d - r;
            const bool active = kind[q] != 0;
            if (__any(active)) n_now += 2;
            double bx = NIw, by = NIw, bm = NIw;
            unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
            if (kind[q] == 1) {
                const bool l2 = l2q[q], r2 = r2q[q], lS = lSq[q], rS = rSq[q];
                const int ti_ = ((gl[q].x & 0xffff) + __umul24(gr[q].x & 0xffff, S)) & 255;
                const double tM = PM.tab2[ti_][0], tX = PM.tab2[ti_][1];
                const double lw0 = (double)__int_as_float(gl[q].z), lw1 = (double)__int_as_float(gl[q].w);
                const double rw0 = (double)__int_as_float(gr[q].z), rw1 = (double)__int_as_float(gr[q].w);
                const double lwA = lS ? lw1 : lw0, lwS = lS ? lw0 : lw1, rwA = rS ? rw1 : rw0, rwS = rS ? rw0 : rw1;
                const pg_d2 lx_ = o_xy[q][3], lm_ = o_xy[q][4], ry_ = o_xy[q][5], rm_ = o_xy[q][6], lr_ = o_xy[q][7];
                const double lxm = o_m[q][3], lmm = o_m[q][4], rym = o_m[q][5], rmm = o_m[q][6], lrm = o_m[q][7];
                // scores only (pg_backptr derives the back-pointers): a state's value is the maximum over its edges of
                // max(own + ge, max(other, M + ng) + go), M's over its pairs of (max(M + tM, max(X, Y) + tX) + lw) + rw -- the
                // reference's candidates with the maxima regrouped (tools/gen_hot_asm.py); absent operands are -inf
                auto gapv = [&](double own, double other, double m_) { return __builtin_fmax(own + ge, __builtin_fmax(other, m_ + ng) + go); };
                auto pairv = [&](double x_, double y_, double m_, double lw, double rw) {
                    return (__builtin_fmax(m_ + tM, __builtin_fmax(x_, y_) + tX) + lw) + rw;
                };
                bx = gapv(Ux[q], Uy[q], Um[q]);
                by = gapv(Py[q], Px[q], Pm[q]);
                bm = pairv(Cx[q], Cy[q], Cm[q], lwA, rwA);
                if (__any(r2)) {
                    by = __builtin_fmax(by, gapv(ry_.y, ry_.x, rym));
                    bm = __builtin_fmax(bm, pairv(rm_.x, rm_.y, rmm, lwA, rwS));
                }
                if (__any(l2)) {
                    bx = __builtin_fmax(bx, gapv(lx_.x, lx_.y, lxm));
                    bm = __builtin_fmax(bm, pairv(lm_.x, lm_.y, lmm, lwS, rwA));
                    if (__any(l2 && r2)) bm = __builtin_fmax(bm, pairv(lr_.x, lr_.y, lrm, lwS, rwS));
                }
            } else if (kind[q] == 2) {
                // first/last rows and columns, sites without edges, more than two edges, ...: the general rules
                const int nl = (gl[q].x >> PR_NE_SHIFT) & 127, nr = (gr[q].x >> PR_NE_SHIFT) & 127;
                double tM_ = 0, tX_ = 0;
                if (r > 0 && j > 0 && nl > 0 && nr > 0) {
                    const int ti_ = ((gl[q].x & 0xffff) + __umul24(gr[q].x & 0xffff, S)) & 255;
                    tM_ = PM.tab2[ti_][0]; tX_ = PM.tab2[ti_][1];
                }
                const pg_i4 gl_ = gl[q], gr_ = gr[q];
                cell_any_t(J, r, j, r > 0 ? nl : 0, j > 0 ? nr : 0, tM_, tX_, no_terminal_edges, reduced_terminal,
                         [&](int p_, int q_, double &xs, double &ys, double &ms) {
                             pg_d2 xy; double m_;
                             const FarAsk f = wcell(p_ >= 0 && q_ >= 0, d - (p_ + q_), p_, xy, m_);
                             {
                                 const FarAsk none = {false, 0};
                                 pg_d2 e1 = {NIw, NIw}, e2 = {NIw, NIw}, e3 = {NIw, NIw}; double m1 = NIw, m2 = NIw, m3 = NIw;
                                 far_fetch4(sc_out, f, none, none, none, xy, m_, e1, m1, e2, m2, e3, m3);
                             }
                             xs = xy.x; ys = xy.y; ms = m_;
                         },
                         [&](int k, int &p_, double &lw) { int dist; edge_at<true>(gl_, k, r, dist, lw); p_ = r - dist; },
                         [&](int k, int &q_, double &rw) { int dist; edge_at<false>(gr_, k, j, dist, rw); q_ = j - dist; },
                         bx, by, bm, px, py, pm);
            }
            if (__any(active && ((gl[q].x | gr[q].x) & PR_SRC))) { if (active) hist_append(gl[q], gr[q], r, j, bx, by, bm); }
            if (r < lo + wpos_n) {                                   // (a second row 384 and more past the band's first: never in the band, and its position is another row's)
                double *o = (double *)((char *)&PM.sc[0][0][0] + wsb + wpos(r) * 24);
                o[PG_X] = bx; o[PG_Y] = by; o[PG_M] = bm;
            }
            if (active) {
                const unsigned off24 = __umul24((unsigned)(r - lo), 24u);
                pg_d2 xy; xy.x = bx; xy.y = by;
                *(PG_GLOBAL pg_d2 *)(srow + off24) = xy;
                *(PG_GLOBAL double *)(srow + off24 + 16u) = bm;
            }
            // the next step's registers: this cell is its (row, j-1), this step's (row-1, j) its (row-1, j-1)
            Px[q] = bx; Py[q] = by; Pm[q] = bm;
            Cx[q] = Ux[q]; Cy[q] = Uy[q]; Cm[q] = Um[q];
        }
        {   // all but the stores of this step and the two before it have retired (older cells of the run are read from L2): a wave
            // issues two stores per set with a row in the band, so the count is worked out, not assumed
            const int tot = n_now + n_hist1 + n_hist2;
            if (tot >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (tot >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (tot >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (tot >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (tot >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (tot >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            n_hist2 = n_hist1; n_hist1 = n_now;
        }
        flag_store(&PM.progress[wave], d);
#ifdef PG_PIPE_STATS
        {
            const long long st_t5 = __builtin_readcyclecounter();
            st_cls_t[4] += st_t5 - st_step0; ++st_cls_n[4];
            st_w[0] += st_t1 - st_step0; st_w[1] += st_t2 - st_t1; st_w[2] += st_t3 - st_t2; st_w[3] += st_t5 - st_t3;
        }
#endif
        ++d;
        lo_prev = lo;
        if (d >= run_end || flag_load(&PM.abort_flag) != 0) break;
        cur.x = __builtin_amdgcn_readfirstlane(nxt.x); cur.y = __builtin_amdgcn_readfirstlane(nxt.y);
        cur.z = __builtin_amdgcn_readfirstlane(nxt.z); cur.w = __builtin_amdgcn_readfirstlane(nxt.w);
    }
    dA = psc[d];                                                   // (the array carries one entry of padding)
    // back to step(): this lane's row for the narrow diagonals, nothing in registers or prefetched; the ring's
    // memory holds wide-ring rows now -- the host marks no diagonal up to here as ring-resident (dp_abi.hip)
    row = dA.x + ((tid - dA.x) & (PNT - 1));
    ok_until = d - 1;
    C_.dA = dA;
    WCTX_OUT(C_);
}

// ---- seven-wave wide run (round 5): a run of at least three class 4 diagonals, ONE row per lane ----
// In wide_run a wave with rows in both of its sets does twice the work of the others, and -- every wave waits for the one
// above it, around the circle -- the run moves at that wave's pace.  The three assist waves have nothing to stage while the
// compute waves are in a wide run (PipeSmem), so they join it: 7 x 64 = 448 lanes >= PG_PIPE_WINDOW = 432 cells, lane T
// (T = the thread's index in the block: compute waves 0..255, assist waves 256..447) holds the row of residue T mod 448 that
// is not below the band, and the step is wide_run's with a single set: P / C in registers, (row-1, j) by one DPP shift (lane 0
// out of the wide ring, written by the wave above), other edges' operands from the wide ring (PWK rows back) or L2, the
// general rules for whatever is not an "easy" interior cell.
// Lock step by PM.wflag: the wave above has completed d-1 and EVERY wave d-2 -- the circle of seven does not imply the
// second from the first two flags as the circle of four does --, so that, as in wide_run, a ring row is rewritten only when
// its readers are done and an L2 operand (at least PWK - 1 diagonals old) has been retired by its writer (every wave keeps
// all but its last three steps' stores retired).  Entry and exit are rendezvous of all seven (PM.warrived): the compute
// waves' stores from before the run have landed when the first step reads them, the assist waves' from inside the run when
// the compute waves' general steps behind it do.  The compute waves publish PM.progress as ever (loader, followers).
// The host marks such runs (descriptor word 4, bit 19 of a class 4 diagonal: dp_abi.hip) and makes their diagonals stops
// of the assist waves' hop counts; PAGAN_DP_WIDE7=0 leaves every wide run to wide_run (A/B switch).
#define PW7 7
#define PW7L (64 * PW7)
static_assert(PW7 == PNW + PNA && PG_PIPE_WINDOW <= PW7L - 16, "seven-wave wide runs: one row per lane");
__device__ __noinline__ void wide_run7(WaveCtx &C_) {
    WCTX_IN(C_);
    WCTX_STATS(C_);
    const PgDevJob *__restrict__ job = (const PgDevJob *)uniform_u64((unsigned long long)C_.job);
    const unsigned flags_ = __builtin_amdgcn_readfirstlane(C_.flags);
    const bool no_terminal_edges = flags_ & 1u, reduced_terminal = !(flags_ & 2u);
    pg_i8 dA = uniform_i8(C_.dA);
    const View J = load_view(job);
    const int d0 = d;
    const int lane = tid & 63, w7 = wave, up7 = w7 == 0 ? PW7 - 1 : w7 - 1;
    const double NIw = neg_inf();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(&PM.wflag[w7], d0 - 1);
    flag_store(&PM.warrived[w7], d0);
    for (int w = 0; w < PW7; ++w) POLLX(&PM.warrived[w], d0, 7);
    p_up = d0 - 1 > p_up ? d0 - 1 : p_up;
    p_dn = d0 - 1 > p_dn ? d0 - 1 : p_dn;
    // the run's diagonals d0 .. run_end - 1 (the same for every wave: the class words alone)
    int run_end = d0 + 1;
    for (;;) {
        const int t = run_end + lane;
        const int c4 = ((PG_GLOBAL const int *)psc)[8 * (t < nd ? t : nd) + 4] & 15;      // (the array carries one entry of padding: class 0)
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(t >= nd || c4 != 4);
        if (stop != 0) { run_end += __builtin_ctzll(stop); break; }
        run_end += 64;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pg_i4 cur = {dA.x, dA.y, dA.z, dA.w};                          // rows lo .. hi of the diagonal, byte offset of its first score
    int wr;                                                        // the lane's row
    { int m_ = (tid - cur.x) % PW7L; m_ += m_ < 0 ? PW7L : 0; wr = cur.x + m_; }
    if (diags_ld < d0) diags_ld = POLLX(&PM.loaded[2], d0, 5);
    // P = (row, j-1) on d-1, C = (row-1, j-1) on d-2, U = (row-1, j) on d-1 (first step only) from L2: everything before the run has landed
    double Px, Py, Pm, Cx, Cy, Cm, Ux, Uy, Um;
    {
        const FarAsk fP = far_ask(psc, d0, 1, wr), fC = far_ask(psc, d0, 2, wr - 1), fU = far_ask(psc, d0, 1, wr - 1), none = {false, 0};
        pg_d2 a = {NIw, NIw}, b = {NIw, NIw}, c = {NIw, NIw}, e = {NIw, NIw};
        double ma = NIw, mb = NIw, mc = NIw, me = NIw;
        far_fetch4(sc_out, fP, fC, fU, none, a, ma, b, mb, c, mc, e, me);
        Px = a.x; Py = a.y; Pm = ma; Cx = b.x; Cy = b.y; Cm = mb; Ux = c.x; Uy = c.y; Um = mc;
    }
    const bool geo_b = (dA.s4 & 16) != 0;
    const int wpos_n = geo_b ? PWPOS_B : PWPOS_A, wk = geo_b ? PWK_B : PWK_A, wage = wk - 2, wrow_bytes = wpos_n * 24;
    int wbase = cur.x - 32;
    auto wpos = [&](int r_) { int q_ = r_ - wbase; q_ -= q_ >= wpos_n ? wpos_n : 0; q_ -= q_ >= wpos_n ? wpos_n : 0; return q_; };
    int n_hist1 = 0, n_hist2 = 0;                                  // stores issued in the previous step / the one before
    int lo_prev = cur.x;
    for (;;) {
        const int lo = cur.x, hi = cur.y;
        if (lo - 2 * wage - 4 - wbase >= wpos_n) wbase += wpos_n;
#ifdef PG_PIPE_STATS
        const long long st_step0 = __builtin_readcyclecounter();
#endif
        {   // records of the diagonal's rows and columns, descriptors of every earlier diagonal
            const int nr_ = hi + 1 < Lx ? hi + 1 : Lx, nc_ = d - lo + 1 < Ly ? d - lo + 1 : Ly;
            if (rows_ld < nr_) rows_ld = POLLX(&PM.loaded[0], nr_, 1);
            if (cols_ld < nc_) cols_ld = POLLX(&PM.loaded[1], nc_, 2);
            if (diags_ld < d) diags_ld = POLLX(&PM.loaded[2], d, 5);
        }
        // the records and the shape of the lane's cell do not depend on the other waves: before the flags
        // (the row the lane will hold after the hand-over below)
        const int r = wr < lo ? wr + PW7L : wr, j = d - r;
        int kind = 0;                                           // 0 outside the band, 1 batched, 2 general rules
        pg_i4 gl = {0, 0, 0, 0}, gr = {0, 0, 0, 0};
        bool l2 = false, r2 = false, l3 = false, r3 = false;   // the site has an other edge / a second other edge (three edges: the third pass below)
        int kL = 1, kR = 1, kL3 = 1, kR3 = 1;
        double lwA = 0, lwS = 0, lwT = 0, rwA = 0, rwS = 0, rwT = 0;       // log weights: the previous-site edge's, the other edge's, the second other edge's
        double tM = 0, tX = 0;
        if (r <= hi) {
            gl = PM.recL[r & (PRW - 1)]; gr = PM.recR[j & (PRW - 1)];
            const int nl = (gl.x >> PR_NE_SHIFT) & 127, nr = (gr.x >> PR_NE_SHIFT) & 127;
            // one site's edges: exactly one from the previous site, beside it none, one or two others (three edges: the third from
            // the edge window, rare -- one site in a thousand at the top of cfg4's tree, but half of the root's wide diagonals hold one,
            // and the general rules it used to take are a loop of single fetches)
            auto shape = [&](auto left_tag, const pg_i4 &rec, int site, int n, bool &two, bool &three, int &k1, int &k3, double &wA, double &w1, double &w3) -> bool {
                constexpr bool LEFT = decltype(left_tag)::value;
                const int e0 = rec.y & 0xffff, e1 = (int)((unsigned)rec.y >> 16);
                const double f0 = (double)__int_as_float(rec.z), f1 = (double)__int_as_float(rec.w);
                two = false; three = false; k1 = 1; k3 = 1; wA = f0; w1 = 0.0; w3 = 0.0;
                if (n == 1) return e0 == 1;
                if (n == 2) {
                    two = true;
                    if (e0 == 1) { wA = f0; k1 = e1; w1 = f1; } else { wA = f1; k1 = e0; w1 = f0; }
                    return (e0 == 1) != (e1 == 1);
                }
                if (n == 3) {
                    int e2; double f2;
                    edge_at<LEFT>(rec, 2, site, e2, f2);
                    two = true; three = true;
                    if (e0 == 1) { wA = f0; k1 = e1; w1 = f1; k3 = e2; w3 = f2; }
                    else if (e1 == 1) { wA = f1; k1 = e0; w1 = f0; k3 = e2; w3 = f2; }
                    else { wA = f2; k1 = e0; w1 = f0; k3 = e1; w3 = f1; }
                    return ((e0 == 1) + (e1 == 1) + (e2 == 1)) == 1;
                }
                return false;
            };
            bool l2_, l3_, r2_, r3_;
            const bool easyL = shape(std::true_type(), gl, r, nl, l2_, l3_, kL, kL3, lwA, lwS, lwT);
            const bool easyR = shape(std::false_type(), gr, j, nr, r2_, r3_, kR, kR3, rwA, rwS, rwT);
            // interior, and no edge in reach starts at site 0 (where the gap-open term differs)
            const bool inner = r >= 2 && r <= Lx - 2 && j >= 2 && j <= Ly - 2 && r - kL >= 1 && j - kR >= 1 && r - kL3 >= 1 && j - kR3 >= 1;
            kind = (easyL && easyR && inner && !(l3_ && r3_)) ? 1 : 2;
            if (kind == 1) { l2 = l2_; r2 = r2_; l3 = l3_; r3 = r3_; }
            if (r > 0 && j > 0 && nl > 0 && nr > 0) {
                const int ti_ = ((gl.x & 0xffff) + __umul24(gr.x & 0xffff, S)) & 255;
                tM = PM.tab2[ti_][0]; tX = PM.tab2[ti_][1];
            }
        }
        // the next diagonal's descriptor: staged by the loader (the window runs PLOOK diagonals ahead of the slowest wave)
        if (diags_ld < d + 2 && d + 1 < nd) diags_ld = POLLX(&PM.loaded[2], d + 2, 5);
        const pg_i4 nxt = PM.dring[(d + 1) & (PDR - 1)];
        const int amax = d - d0 < wage ? d - d0 : wage;               // ages 1 .. amax are in the wide ring
        const int wsb = (d % wk) * wrow_bytes;
        const long long soff = ((long long)cur.w << 32) | (unsigned)cur.z;
        PG_GLOBAL char *srow = (PG_GLOBAL char *)sc_out + soff;
        {   // lock step: lanes 0..6 look at one wave's flag each -- the wave above has completed d-1, every wave d-2
            int spins = 0;
            for (;;) {
                const int v = lane < PW7 ? flag_peek(&PM.wflag[lane]) : 0x7fffffff;
                if (__builtin_amdgcn_ballot_w64(v < (lane == up7 ? d - 1 : d - 2)) == 0) break;
                if ((++spins & 15) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > PSPIN_LIMIT || flag_load(&PM.abort_flag) != 0) {
                        if (flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, PTAG(6));
                        break;
                    }
                }
            }
        }
#ifdef PG_PIPE_STATS
        const long long st_t1 = __builtin_readcyclecounter();
#endif
#ifdef PG_EXP_WIDE_SKIP                                             // timing experiment (WRONG RESULTS): a wide step is its flags and nothing else
        if (true) {
            flag_store(&PM.wflag[w7], d);
            if (w7 < PNW) flag_store(&PM.progress[w7], d);
            ++d; lo_prev = lo;
            if (d >= run_end || flag_load(&PM.abort_flag) != 0) break;
            cur.x = __builtin_amdgcn_readfirstlane(nxt.x); cur.y = __builtin_amdgcn_readfirstlane(nxt.y);
            cur.z = __builtin_amdgcn_readfirstlane(nxt.z); cur.w = __builtin_amdgcn_readfirstlane(nxt.w);
            continue;
        }
#endif
        if (d > d0) {
            // (row-1, j) on d-1: lane T-1's cell, lane 0 from the wide ring; the band's first row of d-1 has no row above it in
            // the band (for lane 0 as well: the position it read is another row's alias -- wide_run has the story)
            int rb1 = wsb - wrow_bytes;
            rb1 += rb1 < 0 ? wk * wrow_bytes : 0;
            const double *c = (const double *)((const char *)&PM.sc[0][0][0] + rb1 + wpos(wr - 1) * 24);
            const double ax = dpp_shr1(Px, c[PG_X]), ay = dpp_shr1(Py, c[PG_Y]), am = dpp_shr1(Pm, c[PG_M]);
            const bool top = wr == lo_prev;
            Ux = top ? NIw : ax; Uy = top ? NIw : ay; Um = top ? NIw : am;
            // row hand-over: the lane's next row (448 on: far below the band) starts from -inf
            if (wr < lo) { wr += PW7L; Px = NIw; Py = NIw; Pm = NIw; Cx = NIw; Cy = NIw; Cm = NIw; Ux = NIw; Uy = NIw; Um = NIw; }
        }
        // one operand cell (p, d - age): read from the wide ring, or -- not there -- what to ask L2 for; -inf outside the band
        auto wcell = [&](bool need, int age, int p_, pg_d2 &xy, double &m) -> FarAsk {
            FarAsk f = {false, 0};
            xy.x = NIw; xy.y = NIw; m = NIw;
            // (a row above the previous diagonal's band: -inf without looking -- its position in the wide ring is the alias of a row
            //  far below the band, which a wave other than the one above this one writes, and that wave need not have completed d-1)
            // (not on the run's first diagonal: lo_prev is the previous diagonal's first row from the second step on, and the first
            //  step's operands all come from L2, where the descriptor says what lies in the band)
            if (!need || (age == 1 && d > d0 && p_ < lo_prev)) return f;
            if (age <= amax) {
                int rb = wsb - age * wrow_bytes;
                rb += rb < 0 ? wk * wrow_bytes : 0;
                const double *c = (const double *)((const char *)&PM.sc[0][0][0] + rb + wpos(p_) * 24);
                xy.x = c[PG_X]; xy.y = c[PG_Y]; m = c[PG_M];
                return f;
            }
            return far_ask(psc, d, age, p_);
        };
        pg_d2 o_xy[8];
        double o_m[8];
        FarAsk o_f[8];
        bool any_far = false;
#pragma unroll
        for (int t = 0; t < 8; ++t) { o_f[t].need = false; o_f[t].boff = 0; o_xy[t].x = NIw; o_xy[t].y = NIw; o_m[t] = NIw; }
        const bool b1 = kind == 1;
        if (__any(b1 && (l2 || r2))) {
            // the batched cells' other-edge operands.  Wave-uniform shortcuts: an operand is skipped when no lane has it, and
            // while every lane's lies in the wide ring the reads are plain LDS reads (no L2 path, no branches)
            auto rd = [&](bool need, int age, int p_, pg_d2 &xy, double &m) {
                int rb = wsb - age * wrow_bytes;
                rb += rb < 0 ? wk * wrow_bytes : 0;
                const int off = need ? rb + wpos(p_) * 24 : (int)offsetof(PipeSmem, null_cell) - (int)offsetof(PipeSmem, sc);
                const double *c = (const double *)((const char *)&PM.sc[0][0][0] + off);
                xy.x = c[PG_X]; xy.y = c[PG_Y]; m = c[PG_M];
            };
            auto fetch = [&](int t, bool need, int age, int p_) {
                if (!__any(need)) return;
                if (!__any(need && age > amax)) rd(need, age, p_, o_xy[t], o_m[t]);
                else { o_f[t] = wcell(need, age, p_, o_xy[t], o_m[t]); any_far = true; }     // (-inf for the lanes without it)
            };
            fetch(3, b1 && l2, kL, r - kL);
            fetch(4, b1 && l2, kL + 1, r - kL);
            fetch(5, b1 && r2, kR, r);
            fetch(6, b1 && r2, kR + 1, r - 1);
            fetch(7, b1 && l2 && r2, kL + kR, r - kL);
        }
#ifdef PG_PIPE_STATS
        const long long st_t2 = __builtin_readcyclecounter();
#endif
        if (any_far) far_fetch8(sc_out, o_f, o_xy, o_m);           // (requests and wait in one statement; nothing if no lane asked)
#ifdef PG_PIPE_STATS
        const long long st_t3 = __builtin_readcyclecounter();
#endif
        const bool active = kind != 0;
        const int n_now = __any(active) ? 2 : 0;                // stores this wave issues in this step
        double bx = NIw, by = NIw, bm = NIw;
        unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;
        if (kind == 1) {
            const pg_d2 lx_ = o_xy[3], lm_ = o_xy[4], ry_ = o_xy[5], rm_ = o_xy[6], lr_ = o_xy[7];
            const double lxm = o_m[3], lmm = o_m[4], rym = o_m[5], rmm = o_m[6], lrm = o_m[7];
            // scores only (pg_backptr derives the back-pointers); the reference's candidates with the maxima regrouped
            // (tools/gen_hot_asm.py); absent operands are -inf
            auto gapv = [&](double own, double other, double m_) { return __builtin_fmax(own + ge, __builtin_fmax(other, m_ + ng) + go); };
            auto pairv = [&](double x_, double y_, double m_, double lw, double rw) {
                return (__builtin_fmax(m_ + tM, __builtin_fmax(x_, y_) + tX) + lw) + rw;
            };
            bx = gapv(Ux, Uy, Um);
            by = gapv(Py, Px, Pm);
            bm = pairv(Cx, Cy, Cm, lwA, rwA);
            if (__any(r2)) {
                by = __builtin_fmax(by, gapv(ry_.y, ry_.x, rym));
                bm = __builtin_fmax(bm, pairv(rm_.x, rm_.y, rmm, lwA, rwS));
            }
            if (__any(l2)) {
                bx = __builtin_fmax(bx, gapv(lx_.x, lx_.y, lxm));
                bm = __builtin_fmax(bm, pairv(lm_.x, lm_.y, lmm, lwS, rwA));
                if (__any(l2 && r2)) bm = __builtin_fmax(bm, pairv(lr_.x, lr_.y, lrm, lwS, rwS));
            }
            // ---- third pass: a site's SECOND other edge (three-edge sites; not on both sides of a cell).  Its gap candidate, its pair
            // with the other side's previous-site edge and -- if the other side has an other edge -- with that: three operand cells,
            // from the wide ring or, past it, from L2 in one fetch ----
            if (__any(l3 || r3)) {
                const bool t3 = l3 || r3;
                const int k3 = l3 ? kL3 : kR3;
                const bool needC = l3 ? r2 : (r3 && l2);
                pg_d2 txy[3]; double tm_[3];
                const FarAsk fA = wcell(t3, k3, l3 ? r - kL3 : r, txy[0], tm_[0]);
                const FarAsk fB = wcell(t3, k3 + 1, l3 ? r - kL3 : r - 1, txy[1], tm_[1]);
                const FarAsk fC = wcell(needC, l3 ? kL3 + kR : kL + kR3, l3 ? r - kL3 : r - kL, txy[2], tm_[2]);
                {
                    const FarAsk none = {false, 0};
                    pg_d2 e = {NIw, NIw}; double me = NIw;
                    far_fetch4(sc_out, fA, fB, fC, none, txy[0], tm_[0], txy[1], tm_[1], txy[2], tm_[2], e, me);
                }
                if (l3) {
                    bx = __builtin_fmax(bx, gapv(txy[0].x, txy[0].y, tm_[0]));
                    bm = __builtin_fmax(bm, pairv(txy[1].x, txy[1].y, tm_[1], lwT, rwA));
                    bm = __builtin_fmax(bm, pairv(txy[2].x, txy[2].y, tm_[2], lwT, rwS));      // (-inf without the other side's other edge)
                } else if (r3) {
                    by = __builtin_fmax(by, gapv(txy[0].y, txy[0].x, tm_[0]));
                    bm = __builtin_fmax(bm, pairv(txy[1].x, txy[1].y, tm_[1], lwA, rwT));
                    bm = __builtin_fmax(bm, pairv(txy[2].x, txy[2].y, tm_[2], lwS, rwT));
                }
            }
        } else if (kind == 2) {
            // first/last rows and columns, sites without edges, more than two edges, ...: the general rules
            const int nl = (gl.x >> PR_NE_SHIFT) & 127, nr = (gr.x >> PR_NE_SHIFT) & 127;
            const pg_i4 gl_ = gl, gr_ = gr;
            cell_any_t(J, r, j, r > 0 ? nl : 0, j > 0 ? nr : 0, tM, tX, no_terminal_edges, reduced_terminal,
                     [&](int p_, int q_, double &xs, double &ys, double &ms) {
                         pg_d2 xy; double m_;
                         const FarAsk f = wcell(p_ >= 0 && q_ >= 0, d - (p_ + q_), p_, xy, m_);
                         {
                             const FarAsk none = {false, 0};
                             pg_d2 e1 = {NIw, NIw}, e2 = {NIw, NIw}, e3 = {NIw, NIw}; double m1 = NIw, m2 = NIw, m3 = NIw;
                             far_fetch4(sc_out, f, none, none, none, xy, m_, e1, m1, e2, m2, e3, m3);
                         }
                         xs = xy.x; ys = xy.y; ms = m_;
                     },
                     [&](int k, int &p_, double &lw) { int dist; edge_at<true>(gl_, k, r, dist, lw); p_ = r - dist; },
                     [&](int k, int &q_, double &rw) { int dist; edge_at<false>(gr_, k, j, dist, rw); q_ = j - dist; },
                     bx, by, bm, px, py, pm);
        }
        if (wr < lo + wpos_n) {                                  // (a row that many past the band's first: never in the band, and its position is another row's)
            double *o = (double *)((char *)&PM.sc[0][0][0] + wsb + wpos(wr) * 24);
            o[PG_X] = bx; o[PG_Y] = by; o[PG_M] = bm;
        }
        if (__any(kind != 0 && ((gl.x | gr.x) & PR_SRC))) { if (kind != 0) hist_append(gl, gr, r, j, bx, by, bm); }
        if (wpos_n > PW7L && wr < lo + wpos_n - PW7L) {           // (512 positions, 448 lanes: the positions of the rows lo + 448 .. lo + 511 -- -inf,
            double *o = (double *)((char *)&PM.sc[0][0][0] + wsb + wpos(wr + PW7L) * 24);      //  what a read above an older diagonal's band lands on)
            o[PG_X] = NIw; o[PG_Y] = NIw; o[PG_M] = NIw;
        }
        {   // all but the stores of the last two steps have retired (this step's are issued behind the flag)
            const int tot = n_hist1 + n_hist2;
            if (tot >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (tot >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            n_hist2 = n_hist1; n_hist1 = n_now;
        }
        flag_store(&PM.wflag[w7], d);
        if (w7 < PNW) flag_store(&PM.progress[w7], d);
        if (active) {
            const unsigned off24 = __umul24((unsigned)(wr - lo), 24u);
            pg_d2 xy; xy.x = bx; xy.y = by;
            *(PG_GLOBAL pg_d2 *)(srow + off24) = xy;
            *(PG_GLOBAL double *)(srow + off24 + 16u) = bm;
        }
        // the next step's registers: this cell is its (row, j-1), this step's (row-1, j) its (row-1, j-1)
        Px = bx; Py = by; Pm = bm;
        Cx = Ux; Cy = Uy; Cm = Um;
#ifdef PG_PIPE_STATS
        if (w7 < PNW) {
            const long long st_t5 = __builtin_readcyclecounter();
            st_cls_t[4] += st_t5 - st_step0; ++st_cls_n[4];
            st_w[0] += st_t1 - st_step0; st_w[1] += st_t2 - st_t1; st_w[2] += st_t3 - st_t2; st_w[3] += st_t5 - st_t3;
        }
#endif
        ++d;
        lo_prev = lo;
        if (d >= run_end || flag_load(&PM.abort_flag) != 0) break;
        cur.x = __builtin_amdgcn_readfirstlane(nxt.x); cur.y = __builtin_amdgcn_readfirstlane(nxt.y);
        cur.z = __builtin_amdgcn_readfirstlane(nxt.z); cur.w = __builtin_amdgcn_readfirstlane(nxt.w);
    }
    // every wave's stores of the run have landed before anyone goes on (the compute waves' general steps read them from L2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(&PM.warrived[w7], d);
    for (int w = 0; w < PW7; ++w) POLLX(&PM.warrived[w], d, 7);
    dA = psc[d];                                                   // (the array carries one entry of padding)
    // back to step() (compute waves): this lane's row for the narrow diagonals, nothing in registers or prefetched
    row = dA.x + ((tid - dA.x) & (PNT - 1));
    ok_until = d - 1;
    C_.dA = dA;
    WCTX_OUT(C_);
}

// ---- seven-wave run of class 5 diagonals (wider than the record windows: every operand, graph arrays included, from L2) ----
// widest_step's cells over 448 lanes instead of 256: a diagonal of 433 .. 922 cells is two passes instead of three or four, and a
// pass is four dependent trips to L2.  Same entry / exit rendezvous and flags as wide_run7; a step starts when EVERY wave has
// completed the one before with its stores retired (each waits for its own before its flag), as the kernel's general steps do
// among the four compute waves.  The host marks runs of at least three such diagonals (bit 19, as for class 4).
__device__ __noinline__ void widest_run7(WaveCtx &C_) {
    WCTX_IN(C_);
    WCTX_STATS(C_);
    const PgDevJob *__restrict__ job = (const PgDevJob *)uniform_u64((unsigned long long)C_.job);
    const unsigned flags_ = __builtin_amdgcn_readfirstlane(C_.flags);
    const bool no_terminal_edges = flags_ & 1u, reduced_terminal = !(flags_ & 2u);
    pg_i8 dA = uniform_i8(C_.dA);
    const int d0 = d;
    const int lane = tid & 63, w7 = wave;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(&PM.wflag[w7], d0 - 1);
    flag_store(&PM.warrived[w7], d0);
    for (int w = 0; w < PW7; ++w) POLLX(&PM.warrived[w], d0, 7);
    p_up = d0 - 1 > p_up ? d0 - 1 : p_up;
    p_dn = d0 - 1 > p_dn ? d0 - 1 : p_dn;
    int run_end = d0 + 1;
    for (;;) {
        const int t = run_end + lane;
        const int c5 = ((PG_GLOBAL const int *)psc)[8 * (t < nd ? t : nd) + 4] & 15;      // (the array carries one entry of padding: class 0)
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(t >= nd || c5 != 5);
        if (stop != 0) { run_end += __builtin_ctzll(stop); break; }
        run_end += 64;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (;;) {
        {   // every wave has completed d - 1 (and retired its stores before saying so)
            int spins = 0;
            for (;;) {
                const int v = lane < PW7 ? flag_peek(&PM.wflag[lane]) : 0x7fffffff;
                if (__builtin_amdgcn_ballot_w64(v < d - 1) == 0) break;
                if ((++spins & 15) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > PSPIN_LIMIT || flag_load(&PM.abort_flag) != 0) {
                        if (flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, PTAG(6));
                        break;
                    }
                }
            }
        }
        const pg_i8 cur = psc[d];
#ifdef PG_PIPE_STATS
        const long long st_step0 = __builtin_readcyclecounter();
#endif
#ifndef PG_EXP_WIDEST_SKIP                                          // (timing experiment, WRONG RESULTS: a class 5 step is its flags and nothing else)
        widest_step(job, psc, d, cur.x, cur.y, tid, no_terminal_edges, reduced_terminal, PW7L);
#endif
        flag_store(&PM.wflag[w7], d);
        if (w7 < PNW) flag_store(&PM.progress[w7], d);
#ifdef PG_PIPE_STATS
        if (w7 < PNW) { st_cls_t[4] += __builtin_readcyclecounter() - st_step0; ++st_cls_n[4]; }
#endif
        ++d;
        if (d >= run_end || flag_load(&PM.abort_flag) != 0) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(&PM.warrived[w7], d);
    for (int w = 0; w < PW7; ++w) POLLX(&PM.warrived[w], d, 7);
    dA = psc[d];                                                   // (the array carries one entry of padding)
    row = dA.x + ((tid - dA.x) & (PNT - 1));
    ok_until = d - 1;
    C_.dA = dA;
    WCTX_OUT(C_);
}

__device__ __noinline__ pg_i4 assist_wide_run(const PgDevJob *__restrict__ job, cdesc8_p psc, int a, int lane, unsigned flags, int d0,
                                              int rows_ld, int cols_ld, int diags_ld) {
    WaveCtx C_;
    C_.job = job; C_.psc = psc; C_.sc_out = (gdouble_w)job->sc; C_.bp_out = (gu32_w)job->bp;
    C_.go = (double)job->go; C_.ng = (double)job->ng; C_.ge = (double)job->ge; C_.tng2 = 0.0; C_.tng1 = 0.0;
    C_.Lx = job->Lx; C_.Ly = job->Ly; C_.nd = job->nd; C_.S = job->S; C_.sleep = job->nd;
    C_.tid = PNT + 64 * a + lane; C_.wave = PNW + a; C_.up = 0; C_.dn = 0; C_.bslot = 0; C_.flags = flags;
    C_.d = d0; C_.row = 0;
    C_.px = 0.0; C_.py = 0.0; C_.pm = 0.0; C_.cx = 0.0; C_.cy = 0.0; C_.cm = 0.0;
    C_.ca = pg_i4{0, 0, 0, 0}; C_.cb = pg_i4{0, 0, 0, 0}; C_.smf = 0.0f;
    C_.dA = psc[d0];
    C_.p_up = -1; C_.p_dn = -1; C_.ok_until = -1; C_.rows_ld = rows_ld; C_.cols_ld = cols_ld; C_.diags_ld = diags_ld;
    C_.as0 = 0; C_.as1 = 0; C_.as2 = 0;
#ifdef PG_PIPE_STATS
    for (int k = 0; k < 5; ++k) { C_.st_cls_t[k] = 0; C_.st_cls_n[k] = 0; }
    for (int k = 0; k < 10; ++k) { C_.st_poll_t[k] = 0; C_.st_poll_n[k] = 0; }
    for (int k = 0; k < 4; ++k) C_.st_w[k] = 0;
#endif
    if ((C_.dA.s4 & 15) == 5) widest_run7(C_); else wide_run7(C_);
    return pg_i4{C_.d, C_.rows_ld, C_.cols_ld, C_.diags_ld};
}

#ifdef PG_PIPE_STATS
#ifndef PG_STAT_CLASS
#define PG_STAT_CLASS 0      // the class whose steps the phase stamps cover (-DPG_STAT_CLASS=1 for multi-edge steps)
#endif
#define PSTAMP(k) do { const long long t_ = __builtin_readcyclecounter(); if (st_on) st_acc[k] += t_ - st_t; st_t = t_; } while (0)
#else
#define PSTAMP(k)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Follower workgroups (blockIdx.x >= n_fill of the same dispatch): back-pointers behind the fill.
//
// The fill stores scores only; the back-pointers are a function of the stored scores (pg_backptr, dp_kernels.hip) that needs
// no wave of the dependency chain.  A banded fill occupies 1 - 31 compute units for 45 - 170 ms, so the pass runs WHILE the
// fill goes on, on units it leaves idle: every wave of a follower workgroup claims chunks of PG_FOLLOW_CHUNK diagonals of a
// job (an atomic counter per job), waits until the scores of the chunk's predecessors have landed (follow[0], published by
// the job's loader wave) and writes the chunk's back-pointers.  The kernel ends a few microseconds after the last fill
// workgroup instead of being followed by a pass over every cell.
//
// Scores are read from L2 (sc1 loads: a line of the frontier may sit in this unit's L1 from an earlier diagonal).  L2 is per
// XCD: a follower serves only fill workgroups of its own XCD.  Workgroups of a dispatch go round the XCDs by index, so
// follower g looks at the jobs w with w % 8 == g % 8 and checks the XCC id the fill workgroup published; a job whose id
// differs, or never shows, is left alone.  Every wait is bounded; what no follower wrote (bp_done[chunk] == 0) pg_backptr
// writes after the kernel, so a follower that gives up costs time, not results.
__device__ __forceinline__ int peek_l2(PG_GLOBAL const int *p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ unsigned my_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

__device__ __noinline__ void follow_chunk(const PgDevJob *__restrict__ job, int chunk, int lane, unsigned flags) {
    const View J = load_view(job);
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const int first = chunk * PG_FOLLOW_CHUNK, end = first + PG_FOLLOW_CHUNK < J.nd ? first + PG_FOLLOW_CHUNK : J.nd;
    for (int d = first; d < end; ++d) {
        const pg_i4 cur = J.dsc[d];
        const int lo = cur.x, hi = cur.y;
        if (hi < lo) continue;
        const long long base = ((long long)cur.w << 32) | (unsigned)cur.z;
        Diag d1 = {0, -1, 0}, d2 = {0, -1, 0};
        if (d > 0) { const pg_i4 p = J.dsc[d - 1]; d1 = {p.x, p.y, ((long long)p.w << 32) | (unsigned)p.z}; }
        if (d > 1) { const pg_i4 p = J.dsc[d - 2]; d2 = {p.x, p.y, ((long long)p.w << 32) | (unsigned)p.z}; }
        for (int i = lo + lane; i <= hi; i += 64) {
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
                         if (ix >= 0) far_cell((PG_GLOBAL const double *)(J.sc + 3 * ix), xs, ys, ms);
                     },
                     [&](int k, int &p, double &lw) { p = J.srcL[l0 + k]; lw = (double)J.lwL[l0 + k]; },
                     [&](int k, int &q, double &rw) { q = J.srcR[r0 + k]; rw = (double)J.lwR[r0 + k]; },
                     bx, by, bm, px, py, pm);
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            u3 b; b.x = px; b.y = py; b.z = pm;
            *(PG_GLOBAL u3 *)(J.bp + 3 * at) = b;
            if (flags & PG_FLAG_SCORE_CHECK) {
                // the recurrence holds at this cell, bit for bit (dp_kernels.hip, pg_backptr has the reasoning): the cell's own
                // scores have landed with its diagonal (the chunk is claimed behind the landed counter)
                double sx, sy, sm_;
                far_cell((PG_GLOBAL const double *)(J.sc + 3 * at), sx, sy, sm_);
                const bool same = __double_as_longlong(bx) == __double_as_longlong(sx) && __double_as_longlong(by) == __double_as_longlong(sy) &&
                                  __double_as_longlong(bm) == __double_as_longlong(sm_);
                if (!same) report_fill_status(job, PG_FILL_SCORE_MISMATCH);
            }
        }
    }
    if (lane == 0) ((PG_GLOBAL unsigned char *)job->bp_done)[chunk] = 1;
}

__device__ __noinline__ void pipe_follower(const PgDevJob *__restrict__ jobs, const int *__restrict__ which, int n_fill, unsigned flags) {
    const int lane = threadIdx.x & 63;
    const unsigned xcc = my_xcc_id();
    // the jobs of this dispatch whose fill workgroup runs on this XCD (at most four: 31 jobs a dispatch)
    int mine[4], n_mine = 0;
    for (int w = (int)(blockIdx.x & 7u); w < n_fill && n_mine < 4; w += 8) {
        const PgDevJob *__restrict__ job = jobs + which[w];
        if (!job->follow) continue;
        int id = 0;
        for (int spin = 0; spin < 20000 && (id = peek_l2((PG_GLOBAL const int *)job->follow + 1)) == 0; ++spin) __builtin_amdgcn_s_sleep(32);
        if (id == (int)xcc + 1) mine[n_mine++] = which[w];
    }
    unsigned left = (1u << n_mine) - 1u;
    int idle = 0;
    const int start = (int)(threadIdx.x >> 6);                    // the waves of a workgroup start at different jobs
    while (left != 0 && idle < 200000) {
        bool worked = false;
        for (int t = 0; t < n_mine; ++t) {
            const int k = (t + start) % n_mine;
            if (!(left & (1u << k))) continue;
            const PgDevJob *__restrict__ job = jobs + mine[k];
            PG_GLOBAL int *fw = (PG_GLOBAL int *)job->follow;
            const int nd = job->nd, n_chunks = (nd + PG_FOLLOW_CHUNK - 1) / PG_FOLLOW_CHUNK;
            const int next = peek_l2(fw + 2);
            if (next >= n_chunks) { left &= ~(1u << k); continue; }
            // a chunk's cells read the diagonals below its last one: claim it once those have landed
            // (with the score check the chunk's own last diagonal is read too: one more)
            const int chk = (flags & PG_FLAG_SCORE_CHECK) ? 1 : 0;
            const int last = ((next + 1) * PG_FOLLOW_CHUNK - 1 < nd - 1 ? (next + 1) * PG_FOLLOW_CHUNK - 1 : nd - 1) + chk;
            if (peek_l2(fw) < last) continue;                      // (follow[0] = landed + 1 >= last: diagonals <= last - 1 are there)
            int c = 0;
            if (lane == 0) c = __hip_atomic_fetch_add(fw + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = __builtin_amdgcn_readfirstlane(c);
            if (c >= n_chunks) { left &= ~(1u << k); continue; }
            // (another wave may have taken `next` in between: c is a later chunk, wait for it)
            const int lastc = ((c + 1) * PG_FOLLOW_CHUNK - 1 < nd - 1 ? (c + 1) * PG_FOLLOW_CHUNK - 1 : nd - 1) + chk;
            int spin = 0;
            while (peek_l2(fw) < lastc && spin < 200000) { __builtin_amdgcn_s_sleep(32); ++spin; }
            if (spin >= 200000) return;                            // the fill stopped publishing: pg_backptr writes this chunk
            follow_chunk(job, c, lane, flags);
            worked = true;
        }
        if (worked) idle = 0; else { __builtin_amdgcn_s_sleep(64); ++idle; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row strips: a wide job (a full matrix, a band wider than the lanes) as a chain of banded jobs.
//
// A strip is PG_STRIP_ROWS = 192 rows of the matrix over all their columns: to this kernel a band whose first and last
// row stand still.  Rows map to lanes as ever (row % 256), so a strip's rows fill three compute waves; the fourth -- the
// wave "upstream" of the strip's first row -- is its FEEDER: it owns the 64 rows above the strip, computes nothing, and
// writes the cells of the last PHALO = 16 of them (a ring operand lies at most PAGE - 1 rows up; the other columns stay
// -inf and are never read), read back from the scores the strip above stored, into its ring columns diagonal by diagonal,
// publishing its progress like any compute wave.  To the strip's waves the rows above are in the ring exactly as if a
// wave had computed them: lane 0 of the first wave takes row-1 from the feeder's lane 63 (the hot loop checks the feeder's
// flag as it checks any upstream wave's), multi-edge cells find their ring operands above the strip in the ring, and ring
// rows are reused under the same per-diagonal rule.  A strip starts when the one above has landed the diagonals its first
// cells read, and then follows it a few dozen diagonals behind: the job's anti-diagonal sweeps all its strips at once.
// Strip -> strip visibility (round 5): a strip's scores and its "landed" counter are written THROUGH to memory (sc1 stores:
// store_scores, the strips' assembly loop, publish_landed) and read with sc1 loads (far_fetch*, peek_l2), so the strips of a
// job run on whatever XCD the dispatcher gives them -- HIP promises no placement.  Round 4's form is behind
// PAGAN_DP_STRIP_SPREAD=0: a job's strips at workgroup indices of one residue mod 8 (one XCD as dispatches are observed to
// go), each strip publishing its XCC id and the feeder below checking it (tag 11; the host's re-runs: dp_abi.hip).
#define PHALO 16                  // rows above the strip the feeder keeps in the ring: a ring operand lies at most PAGE - 1 <= 16 rows up
#define PFEED 32                  // diagonals the feeder requests from L2 at a time: 4 per load (lane = diagonal % 4, row), 8 loads, one round trip
static_assert(PHALO >= PAGE - 1 && 64 / PHALO * 8 == PFEED, "feeder geometry");
__device__ __noinline__ void strip_feeder(const PgDevJob *__restrict__ job, cdesc8_p psc, int tid, int wave, unsigned flags) {
    const int lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(wave);
    const int nd = job->nd, d0 = job->d_first;
    const int hq = lane / PHALO, hr = lane % PHALO;                // this lane's diagonal (of four) and halo row in a request
    const int row = job->strip_row0 - PHALO + hr;
    const int col = row & (PNT - 1);                               // its ring column (the last PHALO lanes' of this wave)
    const int dn = (wave + 1) % PNW;
    PG_GLOBAL const int *prev = (PG_GLOBAL const int *)job->prev_follow;
    const int prev_last = job->prev_nd - 1;
    PG_GLOBAL const pg_i4 *pdsc = (PG_GLOBAL const pg_i4 *)job->pdsc;
    const gdouble_w sc = (gdouble_w)job->sc;
    const double NI = neg_inf();
    int d = d0;                                                    // (names the diagonal in an abort tag)
    flag_store(&PM.arrived[wave], nd);                             // no stores to drain: never what a rendezvous waits for
    if (!(flags & PG_FLAG_STRIPS_SPREAD)) {
        // PAGAN_DP_STRIP_SPREAD=0 (round 4's placement): the strip above runs on this XCD
        int id = 0, spin = 0;
        while ((id = peek_l2(prev + 1)) == 0 && spin < (1 << 22) && flag_load(&PM.abort_flag) == 0) { __builtin_amdgcn_s_sleep(16); ++spin; }
        // (debug flag 0x800, tests: as if it did not -- the host clears the bit when it launches the strips alone)
        if ((id != (int)my_xcc_id() + 1 || (flags & 0x800u)) && flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, PTAG(11));
    }
    int landed1 = 0, p_dn = d0 - 1, slot = d0 % PRK;
    for (int t0 = d0; t0 < nd && flag_load(&PM.abort_flag) == 0; t0 += PFEED) {
        const int t1 = t0 + PFEED < nd ? t0 + PFEED : nd;
        d = t0;
        // what the strip above stored of the diagonals t0 .. t1-1 has landed (its counter ends at its own last diagonal)
        const int need1 = (t1 - 1 < prev_last ? t1 - 1 : prev_last) + 1;
        if (landed1 < need1) {
            int spin = 0;
            while ((landed1 = peek_l2(prev)) < need1) {
                __builtin_amdgcn_s_sleep(4);
                if (++spin > PSPIN_LIMIT / 8 || flag_load(&PM.abort_flag) != 0) {
                    if (flag_load(&PM.abort_flag) == 0) flag_store(&PM.abort_flag, PTAG(12));
                    break;
                }
            }
            if (landed1 < need1) break;
        }
        FarAsk fa[8];
        pg_d2 xy[8];
        double m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                              // request k: the diagonals t0 + 4k .. t0 + 4k + 3, this lane's t0 + 4k + hq
            const int t = t0 + 4 * k + hq;
            const pg_i4 ds = pdsc[t < nd ? t : nd - 1];            // the whole band's rows on t, the offset (in cells) of its first
            fa[k].need = t < t1 && row >= ds.x && row <= ds.y;
            fa[k].boff = 24ll * ((((long long)ds.w << 32) | (unsigned)ds.z) + (row - ds.x));
            xy[k].x = NI; xy[k].y = NI; m[k] = NI;
        }
        // the batch's ring-reuse rule (descriptor word 7: what the downstream wave must have completed, as for any wave) in one load
        const int s7v = ((PG_GLOBAL const int *)psc)[8 * (t0 + (lane & 31) < nd ? t0 + (lane & 31) : nd - 1) + 7];
        far_fetch8(sc, fa, xy, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int q = 0; q < 64 / PHALO; ++q) {
                const int t = t0 + 4 * k + q;
                if (t >= t1) break;
                d = t;
                const int s7 = __builtin_amdgcn_readlane(s7v, 4 * k + q);
                if (s7 > p_dn) p_dn = poll_ge(&PM.progress[dn], s7, PTAG(3));
                if (hq == q) { PM.sc[slot][col][PG_X] = xy[k].x; PM.sc[slot][col][PG_Y] = xy[k].y; PM.sc[slot][col][PG_M] = m[k]; }
                flag_store_inorder(&PM.progress[wave], t);          // (behind the cell: a wave's LDS operations execute in order)
                slot = slot + 1 == PRK ? 0 : slot + 1;
            }
        }
    }
    flag_store(&PM.progress[wave], nd);
}

template <bool TAB_LDS, bool STRIP>
__global__ __launch_bounds__(PBLOCK) void pg_fill_pipe(const PgDevJob *__restrict__ jobs, const int *__restrict__ which,
                                                         unsigned flags, int n_fill) {
    if (!STRIP && (int)blockIdx.x >= n_fill) { pipe_follower(jobs, which, n_fill, flags); return; }
    if (STRIP && which[blockIdx.x] < 0) return;                    // (padding: the strips of a job sit at workgroup indices of one residue mod 8)
    const PgDevJob *__restrict__ job = jobs + which[blockIdx.x];
    if (threadIdx.x == 0 && job->follow) {
        const int id = (int)my_xcc_id() + 1;
        if (STRIP) asm volatile("global_store_dword %0, %1, off sc1" :: "v"((PG_GLOBAL int *)job->follow + 1), "v"(id) : "memory");
        else asm volatile("global_store_dword %0, %1, off" :: "v"((PG_GLOBAL int *)job->follow + 1), "v"(id) : "memory");
    }
    const cdesc8_p psc = (cdesc8_p)job->psc;
    const bool no_terminal_edges = flags & 1u;
    const bool reduced_terminal = !(flags & 2u);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int S = job->S;
    if (TAB_LDS) {
        const float f_ng0 = job->ng;
        const double t2 = (double)(2 * f_ng0), t1 = (double)(0.0f + f_ng0);       // VA:1364-1367: float operations, promoted
        for (int k = tid; k < S * S; k += PBLOCK) { const float t = job->table[k]; PM.tab2[k][0] = t2 + (double)t; PM.tab2[k][1] = t1 + (double)t; }
        // (row strips: site 0's state indexes the table like any other -- a finite term beside the -inf of its cells' operands)
        if (STRIP) for (int k = S * S + tid; k < 256; k += PBLOCK) { PM.tab2[k][0] = 0.0; PM.tab2[k][1] = 0.0; }
    }
    for (int k = tid; k < PRK * PNT * 3; k += PBLOCK) (&PM.sc[0][0][0])[k] = neg_inf();
    if (tid < 4) PM.null_cell[tid] = neg_inf();
    if (tid < PNW) { PM.progress[tid] = STRIP ? job->d_first - 1 : -1; PM.arrived[tid] = STRIP ? job->d_first - 1 : -1; }
    if (tid == 0) { PM.loaded[0] = 0; PM.loaded[1] = 0; PM.loaded[2] = 0; PM.abort_flag = 0; PM.pdsc = STRIP ? job->pdsc : nullptr; }
    if (tid < 8) { PM.wflag[tid] = -1; PM.warrived[tid] = -1; }
#ifdef PG_PIPE_STATS
    if (tid == 0) PM.far_limit = 24ll * job->cells;
#endif
    if (tid < PNA) PM.assist_done[tid] = -1;
    __syncthreads();

#ifdef PG_PIPE_STATS
    if (tid < 4 && 3 * (job->Lx + job->Ly) >= 4096 && PG_STATS_MINE(job, flags))           // slot counters of the per-run records (hot_run)
        ((PG_GLOBAL int *)job->trace)[((3 * (job->Lx + job->Ly) / 4 * 3 + 15) & ~15) + tid] = 0;
    if (lane == 0 && 3 * (job->Lx + job->Ly) >= 4096 && PG_STATS_MINE(job, flags)) {       // which SIMD / CU every wave of the workgroup landed on
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        ((PG_GLOBAL int *)job->trace)[3 * (job->Lx + job->Ly) - 1100 + (tid >> 6)] = (int)hwid;
    }
#endif
    if (tid >= PNT + 64 * PNA) {
        const View J = load_view(job);
        pipe_loader(J, psc, lane, (PG_GLOBAL int *)job->follow, STRIP ? job : nullptr, TAB_LDS,
                    (TAB_LDS && !STRIP) ? (PG_GLOBAL const unsigned char *)job->hfL : nullptr, (TAB_LDS && !STRIP) ? (PG_GLOBAL const unsigned char *)job->hfR : nullptr);
        return;
    }
    if (tid >= PNT) {
#ifdef PG_ASSIST_IDLE                                          // timing experiment (wrong results): nothing staged, everything "done"
        if (lane == 0) flag_store(&PM.assist_done[(tid - PNT) >> 6], 0x7ffffff0);
        return;
#endif
        if constexpr (TAB_LDS) pipe_assist_lean<STRIP>(job, psc, __builtin_amdgcn_readfirstlane((tid - PNT) >> 6), lane, flags);
        else pipe_assist<false, STRIP>(job, psc, __builtin_amdgcn_readfirstlane((tid - PNT) >> 6), lane, flags);
        return;
    }

    // Only what the steady state touches stays in SGPRs across the loop; the wide path reloads the full
    // job view (graph arrays, band index) when it runs -- the descriptor is ~40 pointers, and keeping
    // them live spilled the per-diagonal scalars.
    const int Lx = job->Lx, Ly = job->Ly, nd = job->nd;
    const gdouble_w sc_out = (gdouble_w)job->sc;
    const gu32_w bp_out = (gu32_w)job->bp;
    const gfloat_p table = (gfloat_p)job->table;
    const float f_go = job->go, f_ge = job->ge, f_gE = job->gE, f_ng = job->ng;

    // ================= compute waves =================
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform: keeps the schedule and the step counter in SGPRs
    const int up = (wave + PNW - 1) % PNW, dn = (wave + 1) % PNW;
    if (STRIP && wave == job->feed_wave) {
        strip_feeder(job, psc, tid, wave, flags);
        if (lane == 0) {
            const int aborted = flag_load(&PM.abort_flag);
            if (aborted != 0) report_fill_status(job, aborted);
        }
        return;
    }
    const cint_p sched = (cint_p)job->sched + ((cint_p)job->sched)[wave];     // awake intervals [a,b) of this wave
    const double NI = neg_inf();
    // the recurrence's constants live in VGPRs: a VALU instruction reads one SGPR operand at most, and the
    // loop is short of SGPRs, not of VGPRs
    const double go = in_vgpr((double)f_go), ng = in_vgpr((double)f_ng), ge = in_vgpr((double)f_ge);
    const double tng2 = in_vgpr((double)(2 * f_ng)), tng1 = in_vgpr((double)(0.0f + f_ng));
    const int bslot = (tid + PNT - 1) & (PNT - 1);                 // ring column of row-1 (lane 0: the upstream wave's lane 63)
    int rows_ld = 0, cols_ld = 0, diags_ld = 0;
    int as0 = -1, as1 = -1, as2 = -1;                              // cached progress of the assist waves (wave a: diagonals d % 3 == a)
#ifdef PG_PIPE_STATS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_cls_t[5] = {0, 0, 0, 0, 0};
    int st_cls_n[5] = {0, 0, 0, 0, 0};
    long long st_w[4] = {0, 0, 0, 0};
    long long st_poll_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int st_poll_n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int st_n = 0;
    bool st_on = false;
#endif

    for (int iv = 0;; iv += 2) {
        const int wake = sched[iv], sleep = sched[iv + 1];
        // asleep until `wake`: this wave's ring columns are -inf at every depth, its stores have retired
        // and it reads nothing
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (wake >= nd: the wave's last interval is behind it and so are its stores -- `nd`, one more than any diagonal's
        //  flag, is what tells the loader that the last diagonals have landed too: pipe_loader)
        flag_store(&PM.arrived[wave], wake >= nd ? nd : wake - 1);
        flag_store(&PM.progress[wave], wake >= nd ? nd : wake - 1);
        if (wake >= nd || flag_load(&PM.abort_flag) != 0) break;

        cdesc8_p pp = psc + wake;                                  // descriptor last requested
        pg_i8 dA = *pp, dB = dA;                                   // descriptors of this step / the next one (roles alternate)
        int row = dA.x + ((tid - dA.x) & (PNT - 1));               // smallest row of this lane's residue inside or below the band
        bool have = false;                                         // rL holds row's record
        pg_i4 rL = {0, 0, 0, 0};
        pg_i4 ca = {0, 0, 0, 0}, cb = {0, 0, 0, 0};                // column records of this step / the next one (roles alternate)
        float smf = 0.0f;                                          // model score of this lane's cell on d (prefetched)
        // four register sets whose roles rotate with period 2 (no copies in the steady state):
        //   P  this lane's cell on d-1, (row, j-1)          C  (row-1, j-1) on d-2: last step's shift
        //   A  in: lane 0's operand if prefetched, out: (row-1, j) on d-1          R  out: this step's cell
        double r1x = NI, r1y = NI, r1m = NI, r2x = NI, r2y = NI, r2m = NI;
        double r3x = NI, r3y = NI, r3m = NI, r4x = NI, r4y = NI, r4m = NI;
        bool nb_valid = false;
        int p_up = -1, p_dn = -1;                                  // cached progress of the neighbours
        int ok_until = wake - 1;                                   // flow control holds through this diagonal without reading a flag
        int slot = wake % PRK, slot1 = (wake + PRK - 1) % PRK;     // d % PRK, (d-1) % PRK
        int stg = wake % PST;                                       // d % PST: the staging slot (and assist wave) of this diagonal
        int lo_prev = -1, hi_prev = -1;

        // One cell by the general rules (first/last rows and columns, any edge
        // list): records and edges from the LDS windows, earlier cells from the ring or from L2/HBM.
        auto gen_cell = [&](const int d, const int slot_, const unsigned resmask, const int r, const int j, double &bx,
                            double &by, double &bm, unsigned &px, unsigned &py, unsigned &pm) {
            const pg_i4 gl = PM.recL[r & (PRW - 1)], gr = PM.recR[j & (PRW - 1)];
            const int wi = gl.x, wj = gr.x;
            int l0 = 0, nL = 0, r0 = 0, nR = 0, n_items = 0;
            double tM = 0, tX = 0;
            if (r > 0) { l0 = PM.ebL[r & (PRW - 1)]; nL = (wi >> PR_NE_SHIFT) & 127; }
            if (j > 0) { r0 = PM.ebR[j & (PRW - 1)]; nR = (wj >> PR_NE_SHIFT) & 127; }
            if (r == 0 && j == 0) bm = 0.0;                                    // initialise_array_corner, VA:725-736
            else n_items = (nL > 0 ? nL : 1) * (nR > 0 ? nR : 1);
            if (nL > 0 && nR > 0) {
                const int ti = (wi & 0xffff) + (wj & 0xffff) * S;
                if (TAB_LDS) { tM = PM.tab2[ti][0]; tX = PM.tab2[ti][1]; }
                else { const float sm = far_f32(table + ti); tM = tng2 + (double)sm; tX = tng1 + (double)sm; }
            }
            const double extX = (double)(((j == 0 || j == Ly - 1) && !no_terminal_edges) ? f_gE : f_ge);
            const double extY = (double)(((r == 0 || r == Lx - 1) && !no_terminal_edges) ? f_gE : f_ge);
            const int nRp = nR > 0 ? nR : 1;
            int k1 = 0, k2 = 0;
            for (int t = 0; t < n_items; ++t) {
                int p = 0, q = 0;
                double lw = 0, rw = 0, c;
                if (nL > 0) { p = PM.esL[(l0 + k1) & (PEC - 1)]; lw = (double)PM.ewL[(l0 + k1) & (PEC - 1)]; }
                if (nR > 0) { q = PM.esR[(r0 + k2) & (PEC - 1)]; rw = (double)PM.ewR[(r0 + k2) & (PEC - 1)]; }
                double v[3][3];
                const CellAsk ax = {nL > 0 && k2 == 0, r - p, p}, ay = {nR > 0 && k1 == 0, j - q, r};
                const CellAsk am = {nL > 0 && nR > 0, (r - p) + (j - q), p};
                old_cells3<true>(sc_out, psc, d, slot_, resmask, ax, ay, am, v);
                if (nL > 0 && k2 == 0) {                                       // X candidates of left edge k1
                    const double open = (reduced_terminal && p == 0) ? 0.0 : go;
                    c = v[0][0] + extX;          if (c > bx) { bx = c; px = pack_bp(PG_X, k1, 0, p == r - 1, false); }
                    c = (v[0][1] + 0.0) + go;    if (c > bx) { bx = c; px = pack_bp(PG_Y, k1, 0, p == r - 1, false); }
                    c = (v[0][2] + ng) + open;   if (c > bx) { bx = c; px = pack_bp(PG_M, k1, 0, p == r - 1, false); }
                }
                if (nR > 0 && k1 == 0) {                                       // Y candidates of right edge k2
                    const double open = (reduced_terminal && q == 0) ? 0.0 : go;
                    c = v[1][1] + extY;          if (c > by) { by = c; py = pack_bp(PG_Y, 0, k2, false, q == j - 1); }
                    c = (v[1][0] + 0.0) + go;    if (c > by) { by = c; py = pack_bp(PG_X, 0, k2, false, q == j - 1); }
                    c = (v[1][2] + ng) + open;   if (c > by) { by = c; py = pack_bp(PG_M, 0, k2, false, q == j - 1); }
                }
                if (nL > 0 && nR > 0) {                                        // M candidates of the pair
                    c = ((v[2][2] + tM) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_M, k1, k2, p == r - 1, q == j - 1); }
                    c = ((v[2][0] + tX) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_X, k1, k2, p == r - 1, q == j - 1); }
                    c = ((v[2][1] + tX) + lw) + rw;  if (c > bm) { bm = c; pm = pack_bp(PG_Y, k1, k2, p == r - 1, q == j - 1); }
                }
                if (++k2 == nRp) { k2 = 0; ++k1; }
            }
        };

        // One diagonal.  HOT (a tag type): the class is 0 or 1 and the step runs as one half of a pair.
        auto step = [&](auto hot_tag, const int d, const pg_i8 &cur, pg_i8 &oth, double &PX, double &PY, double &PM_,
                        double &CX, double &CY, double &CM, double &AX, double &AY, double &AM, double &RX, double &RY,
                        double &RM, pg_i4 &cra, const pg_i4 &crb) {
            constexpr bool HOT = decltype(hot_tag)::value;
            const int lo = cur.x, hi = cur.y, cls = cur.s4 & 15;
            PSTAMP(6);
#ifdef PG_PIPE_STATS
            const long long st_step0 = __builtin_readcyclecounter();
            const bool st_has = __any(row <= hi && row >= lo);
#endif
            // ---- flow control: flags are read only when the cached values stop covering this step ----
            if (d > ok_until) {
                int need = hi + 3 < Lx - 1 ? hi + 3 : Lx - 1;
                rows_ld = flag_load(&PM.loaded[0]); cols_ld = flag_load(&PM.loaded[1]);      // (afresh: see hot_run)
                if (rows_ld <= need) rows_ld = POLL(&PM.loaded[0], need + 1, 1);
                int margin = rows_ld >= Lx ? nd : rows_ld - 1 - need;
                need = d + 2 - lo < Ly - 1 ? d + 2 - lo : Ly - 1;
                if (cols_ld <= need) cols_ld = POLL(&PM.loaded[1], need + 1, 2);
                const int mc = cols_ld >= Ly ? nd : cols_ld - 1 - need;
                margin = mc < margin ? mc : margin;
                ok_until = d + margin;
            }
            // downstream neighbour: this step overwrites the ring row of diagonal d - PRK; the host worked out the
            // last diagonal whose cells still read that one (cur.s7; 2 diagonals back in simple stretches, up to
            // PAGE-1 where long edges are about)
            if (cur.s7 > p_dn) p_dn = POLL(&PM.progress[dn], cur.s7, 3);
            // upstream neighbour: only a wave with a row about to use (row-1, .) has to wait for it
            if (d - 1 > p_up && __any(row <= hi + 1)) p_up = POLL(&PM.progress[up], d - 1, 4);
            if (!HOT && cls >= 3) {
                if (diags_ld < d) diags_ld = POLL(&PM.loaded[2], d, 5);      // descriptor window covers every earlier diagonal
                {
                    // rendezvous: every awake wave has completed d-1 and its stores have landed
#ifdef PG_PIPE_STATS
                    const long long w0 = __builtin_readcyclecounter();
#endif
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PG_PIPE_STATS
                    const long long w1 = __builtin_readcyclecounter();
#endif
                    flag_store(&PM.arrived[wave], d);
                    for (int w = 0; w < PNW; ++w) POLL(&PM.arrived[w], d, 7);
#ifdef PG_PIPE_STATS
                    if (cls == 4) { st_w[0] += w1 - w0; st_w[1] += __builtin_readcyclecounter() - w1; }
#endif
                    p_up = d - 1 > p_up ? d - 1 : p_up;
                    p_dn = d - 1 > p_dn ? d - 1 : p_dn;
                }
            }
            // ---- large model table: the assist wave of this diagonal gathered the cells' model scores ----
            if (!TAB_LDS && (HOT || cls <= 2)) {
                if (stg == 0) { if (as0 < d) as0 = POLL(&PM.assist_done[0], d, 8); }
                else if (stg == 1) { if (as1 < d) as1 = POLL(&PM.assist_done[1], d, 8); }
                else { if (as2 < d) as2 = POLL(&PM.assist_done[2], d, 8); }
                smf = PM.ssm[stg][tid];
            }
            PSTAMP(0);

            // ---- (row-1, j) on d-1: lane-1's registers, lane 0 from the ring (prefetched into A) ----
            if (!nb_valid) {
                if (p_up >= d - 1 && d > 0) {
                    AX = PM.sc[slot1][bslot][PG_X]; AY = PM.sc[slot1][bslot][PG_Y]; AM = PM.sc[slot1][bslot][PG_M];
                } else {
                    AX = NI; AY = NI; AM = NI;                     // no lane of this wave can use it (see header)
                }
            }
            AX = dpp_shr1(PX, AX); AY = dpp_shr1(PY, AY); AM = dpp_shr1(PM_, AM);
            PSTAMP(1);
            // next descriptor: requested here, after the step's LDS wait, so that the scalar load has the
            // whole step to land (an s_waitcnt on LDS data also waits for scalar loads in flight); the
            // array carries one entry of padding
            {   // (the empty asm ties the address to the shift's result: without it the compiler hoists the load
                //  above the LDS wait)
                unsigned long long pv = (unsigned long long)(pp + 1);
                asm volatile("" : "+s"(pv) : "v"(AX), "v"(AY), "v"(AM));
                pp = (cdesc8_p)pv;
            }
            oth = *pp;

            // ---- row hand-over: a lane whose row left the band takes the next one of its residue ----
            if (lo != lo_prev) {
                lo_prev = lo;
                if (row < lo) { row += PNT; PX = NI; PY = NI; PM_ = NI; have = false; }
                hi_prev = -2;
            }
            if (hi != hi_prev) {
                hi_prev = hi;
                if (!have && row <= hi + 3) { rL = PM.recL[row & (PRW - 1)]; have = true; }
            }
            PSTAMP(2);
            const int j = d - row;
            const bool active = row <= hi;
            double bx = NI, by = NI, bm = NI;
            unsigned px = PG_BP_NONE, py = PG_BP_NONE, pm = PG_BP_NONE;

            if (HOT || cls <= 2) {
                // ================= interior diagonal: the previous diagonal in registers, the rest staged =================
                if (active) {
                    // straight-line code for the edges from the previous sites: `+ 0.0` (log_gap_close, unit edge
                    // weights) is omitted -- exact, no score is ever -0.0
                    // (row strips: the terminal extension rate in the first / last column (x-gap) and row (y-gap); a banded job's
                    //  diagonals with such cells are general steps)
                    const bool term_on = STRIP && !no_terminal_edges;
                    const double gex = (term_on && (j == 0 || j == Ly - 1)) ? (double)f_gE : ge;
                    const double gey = (term_on && (row == 0 || row == Lx - 1)) ? (double)f_gE : ge;
                    bx = first_max3(AX + gex, AY + go, (AM + ng) + go, PG_X | PG_BP_ADJL, PG_Y | PG_BP_ADJL, PG_M | PG_BP_ADJL, px);
                    by = first_max3(PY + gey, PX + go, (PM_ + ng) + go, PG_Y | PG_BP_ADJR, PG_X | PG_BP_ADJR, PG_M | PG_BP_ADJR, py);
                    const double tM = tng2 + (double)smf, tX = tng1 + (double)smf;
                    bm = first_max3(CM + tM, CX + tX, CY + tX, PG_M | PG_BP_ADJL | PG_BP_ADJR, PG_X | PG_BP_ADJL | PG_BP_ADJR,
                                    PG_Y | PG_BP_ADJL | PG_BP_ADJR, pm);
                }
                if (cls != 0) {
                    // ---- what the assist wave of this diagonal staged: the multi-edge candidates that do not need
                    // diagonal d-1.  As late in the step as possible: the assist wave could only start on d when d-2
                    // was complete, and everything above ran beside it.  The flag is read only when the cached value
                    // stops covering d ----
                    if (stg == 0) { if (as0 < d) as0 = POLL(&PM.assist_done[0], d, 8); }
                    else if (stg == 1) { if (as1 < d) as1 = POLL(&PM.assist_done[1], d, 8); }
                    else { if (as2 < d) as2 = POLL(&PM.assist_done[2], d, 8); }
                    if (active) {
                        const double ex = PM.sx[stg][tid], ey = PM.sy[stg][tid], em = PM.sM[stg][tid];
                        const unsigned sfx = PM.spx[stg][tid], sfy = PM.spy[stg][tid], sfm = PM.spm[stg][tid];
                        // merge (branch-free; simple sites keep what they have).  The previous-site edge's candidates
                        // sit at that edge's list position: a staged winner takes a tie only if its edge comes first
                        // (first_is_bigger is strict, basic_alignment.h:449-462).
                        const bool msL = !(rL.x & PR_SIMPLE), msR = !(cra.x & PR_SIMPLE);
                        const bool tkx = msL && ((sfx & PS_ONLY) || ex > bx || (ex == bx && (sfx & PS_FIRST)));
                        const bool tky = msR && ((sfy & PS_ONLY) || ey > by || (ey == by && (sfy & PS_FIRST)));
                        px |= msL ? ((sfx >> 18) & 127u) << 4 : 0u;            // the previous-site edge's slot in the left list
                        py |= msR ? ((sfy >> 4) & 127u) << 18 : 0u;            // ... in the right list
                        px = bx > NI ? px : PG_BP_NONE;
                        py = by > NI ? py : PG_BP_NONE;
                        bx = tkx ? ex : bx;  px = tkx ? (sfx & 0x3ffffu) : px;
                        by = tky ? ey : by;  py = tky ? (sfy & 0x01fc000fu) : py;
                        const bool tkm = msL || msR;
                        bm = tkm ? em : bm;  pm = tkm ? sfm : pm;
                    }
                }
                PSTAMP(3);
                commit_cell<STRIP>(sc_out, bp_out, cur, slot, tid, row - lo, active, bx, by, bm, px, py, pm);
                PSTAMP(4);
            } else if (cls <= 3) {
                const unsigned resmask = ((unsigned)cur.s4 >> 5) & 0x7fffu;
#ifndef PG_EXP_GENERAL_SKIP                                         // (timing experiment, WRONG RESULTS: a general step computes nothing)
                if (active) gen_cell(d, slot, resmask, row, j, bx, by, bm, px, py, pm);
#endif
                commit_cell<STRIP>(sc_out, bp_out, cur, slot, tid, row - lo, active, bx, by, bm, px, py, pm);
                if (TAB_LDS && !STRIP && (cur.s4 & 32) && active)      // (a history interval crosses this general step: hist_append)
                    hist_append(PM.recL[row & (PRW - 1)], PM.recR[j & (PRW - 1)], row, j, bx, by, bm);
            } else if (cls == 4) {
                // ---- wider than the lanes, but inside the record windows: every lane takes its rows row, row+256,
                // ...; cells come from L2 (all waves are here and drained), simple interior cells with their
                // three predecessors requested together, the others by the general rules; nothing enters the ring ----
                const unsigned resmask = ((unsigned)cur.s4 >> 5) & 0x7fffu;
#ifdef PG_PIPE_STATS
                const long long w2 = __builtin_readcyclecounter();
#endif
                const pg_i8 p1 = psc[d - 1], p2 = psc[d - 2];              // a wide diagonal has d > 240
                const long long off1 = ((long long)p1.w << 32) | (unsigned)p1.z, off2 = ((long long)p2.w << 32) | (unsigned)p2.z;
                const long long soff = ((long long)cur.w << 32) | (unsigned)cur.z;
                PG_GLOBAL char *s1 = (PG_GLOBAL char *)sc_out + off1, *s2 = (PG_GLOBAL char *)sc_out + off2;
                PG_GLOBAL char *srow = (PG_GLOBAL char *)sc_out + soff, *brow = (PG_GLOBAL char *)bp_out + (soff >> 1);
                for (int r = row; r <= hi; r += PNT) {
                    const int jj = d - r;
                    double wx = NI, wy = NI, wm = NI;
                    unsigned qx = PG_BP_NONE, qy = PG_BP_NONE, qm = PG_BP_NONE;
                    const pg_i4 gl = PM.recL[r & (PRW - 1)], gr = PM.recR[jj & (PRW - 1)];
                    if ((gl.x & gr.x & PR_SIMPLE) && r >= 2 && r <= Lx - 2 && jj >= 2 && jj <= Ly - 2) {
                        // (r-1, jj) and (r, jj-1) on d-1, (r-1, jj-1) on d-2; outside the band: -inf
                        const int pa = r - 1;
                        const bool inA = pa >= p1.x && pa <= p1.y, inB = r <= p1.y, inC = pa >= p2.x && pa <= p2.y;
                        const int ca_ = pa < p1.x ? p1.x : (pa > p1.y ? p1.y : pa), cb_ = r > p1.y ? p1.y : r;
                        const int cc_ = pa < p2.x ? p2.x : (pa > p2.y ? p2.y : pa);
                        pg_d2 axy = {NI, NI}, bxy = {NI, NI}, cxy = {NI, NI}, dxy = {NI, NI};
                        double am = NI, bmm = NI, cm = NI, dmm = NI;
                        const FarAsk fa_ = {true, off1 + 24ll * (ca_ - p1.x)}, fb_ = {true, off1 + 24ll * (cb_ - p1.x)};
                        const FarAsk fc_ = {true, off2 + 24ll * (cc_ - p2.x)}, fn_ = {false, 0};
                        const int ti = (gl.x & 0xffff) + (gr.x & 0xffff) * S;
                        double tM = 0, tX = 0;
                        if (TAB_LDS) { tM = PM.tab2[ti & 255][0]; tX = PM.tab2[ti & 255][1]; }
                        far_fetch4(sc_out, fa_, fb_, fc_, fn_, axy, am, bxy, bmm, cxy, cm, dxy, dmm);
                        if (!TAB_LDS) { const float smv = far_f32(table + ti); tM = tng2 + (double)smv; tX = tng1 + (double)smv; }
                        const double ax = inA ? axy.x : NI, ay = inA ? axy.y : NI, amv = inA ? am : NI;
                        const double bxv = inB ? bxy.x : NI, byv = inB ? bxy.y : NI, bmv = inB ? bmm : NI;
                        const double cxv = inC ? cxy.x : NI, cyv = inC ? cxy.y : NI, cmv = inC ? cm : NI;
                        wx = first_max3(ax + ge, ay + go, (amv + ng) + go, PG_X | PG_BP_ADJL, PG_Y | PG_BP_ADJL, PG_M | PG_BP_ADJL, qx);
                        wy = first_max3(byv + ge, bxv + go, (bmv + ng) + go, PG_Y | PG_BP_ADJR, PG_X | PG_BP_ADJR, PG_M | PG_BP_ADJR, qy);
                        wm = first_max3(cmv + tM, cxv + tX, cyv + tX, PG_M | PG_BP_ADJL | PG_BP_ADJR, PG_X | PG_BP_ADJL | PG_BP_ADJR,
                                        PG_Y | PG_BP_ADJL | PG_BP_ADJR, qm);
                    } else {
                        const int nl = (gl.x >> PR_NE_SHIFT) & 127, nr = (gr.x >> PR_NE_SHIFT) & 127;
                        if ((unsigned)(nl - 1) < 2u && (unsigned)(nr - 1) < 2u && r >= 2 && r <= Lx - 2 && jj >= 2 && jj <= Ly - 2) {
                            // interior cell, at most two bwd edges per site: its eight operands in one round trip
                            const int ti = (gl.x & 0xffff) + (gr.x & 0xffff) * S;
                            double tM, tX;
                            if (TAB_LDS) { tM = PM.tab2[ti & 255][0]; tX = PM.tab2[ti & 255][1]; }
                            else { const float smv = far_f32(table + ti); tM = tng2 + (double)smv; tX = tng1 + (double)smv; }
                            multi2_cell<true>(sc_out, psc, d, resmask, slot, gl, gr, r, jj, reduced_terminal, go, ge, ge, ng,
                                              tM, tX, wx, wy, wm, qx, qy, qm);
                        } else {
                            gen_cell(d, slot, resmask, r, jj, wx, wy, wm, qx, qy, qm);
                        }
                    }
                    typedef unsigned u3 __attribute__((ext_vector_type(3)));
                    const unsigned off = (unsigned)(r - lo);
                    pg_d2 xy; xy.x = wx; xy.y = wy;
                    *(PG_GLOBAL pg_d2 *)(srow + 24u * off) = xy;
                    *(PG_GLOBAL double *)(srow + 24u * off + 16u) = wm;
                    u3 b3; b3.x = qx; b3.y = qy; b3.z = qm;
                    *(PG_GLOBAL u3 *)(brow + 12u * off) = b3;
                }
#ifdef PG_PIPE_STATS
                const long long w3 = __builtin_readcyclecounter();
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PG_PIPE_STATS
                st_w[2] += w3 - w2; st_w[3] += __builtin_readcyclecounter() - w3;
#endif
                have = false;
                hi_prev = -2;
            } else {
                // ---- wider than the record windows: every cell from HBM/L2 operands, graph arrays included ----
#ifndef PG_EXP_WIDEST_SKIP                                          // (timing experiment, WRONG RESULTS: a class 5 step is its rendezvous and nothing else)
                widest_step(job, psc, d, lo, hi, tid, no_terminal_edges, reduced_terminal);
#endif
                have = false;                                      // the record window may have lapped this lane's row
                hi_prev = -2;
            }

            // ---- hand the step over: R is the lane's cell, A the shift; both are read by the next step ----
            RX = bx; RY = by; RM = bm;
            flag_store(&PM.progress[wave], d);                     // after the ring writes: LDS ops of a wave execute in order
            // Operand pipeline, so that a step has ONE LDS wait, at its top: the column record two steps
            // ahead (column d+2-row) replaces this step's, the model score one step ahead (a lane's row
            // record is in its registers at least three steps before the row enters the band).
            cra = PM.recR[(d + 2 - row) & (PRW - 1)];
            // (small tables: classes 0..2 run in hot_run, which takes the match terms from PM.tab2 itself)
            // lane 0's operand of the next step, if the upstream wave has already produced it: into the set
            // that was C here and is A there
            nb_valid = p_up >= d;
            if (nb_valid) { CX = PM.sc[slot][bslot][PG_X]; CY = PM.sc[slot][bslot][PG_Y]; CM = PM.sc[slot][bslot][PG_M]; }
            slot1 = slot;
            slot = slot + 1 == PRK ? 0 : slot + 1;
            stg = stg + 1 == PST ? 0 : stg + 1;
            PSTAMP(5);
#ifdef PG_PIPE_STATS
            if (st_has) {
                const long long dt = __builtin_readcyclecounter() - st_step0;
                for (int c = 0; c < 5; ++c) if (c == cls) { st_cls_t[c] += dt; ++st_cls_n[c]; }
            }
#endif
        };


        int d = wake;
        while (d < sleep) {
            // a wait somewhere ran into its spin limit: every wave leaves as fast as it can (what it would compute from here on is
            // built on operands nobody waited for); the status is reported below
            if (flag_load(&PM.abort_flag) != 0) { d = nd; break; }
#ifdef PG_PIPE_STATS
            st_on = (dA.s4 & 15) == PG_STAT_CLASS && __any(row <= dA.y && row >= dA.x);
            st_t = __builtin_readcyclecounter();
            if (st_on) st_n += 2;
#endif
            if (TAB_LDS && (STRIP ? (dA.s4 & 7) <= 2 : ((dA.s4 & 15) <= 2 || (dA.s4 & 15) == 4 || ((dA.s4 & 15) == 5 && (dA.s4 & (1 << 19)))))) {      // model table in LDS: classes 0..2, 4 and seven-wave runs of 5 as functions of their own
                WaveCtx C_;
                C_.job = job; C_.psc = psc; C_.sc_out = sc_out; C_.bp_out = bp_out;
                C_.go = go; C_.ng = ng; C_.ge = ge; C_.tng2 = tng2; C_.tng1 = tng1;
                C_.Lx = Lx; C_.Ly = Ly; C_.nd = nd; C_.S = S; C_.sleep = sleep;
                C_.tid = tid; C_.wave = wave; C_.up = up; C_.dn = dn; C_.bslot = bslot; C_.flags = flags;
                C_.d = d; C_.row = row;
                C_.px = r1x; C_.py = r1y; C_.pm = r1m; C_.cx = r2x; C_.cy = r2y; C_.cm = r2m;
                C_.ca = ca; C_.cb = cb; C_.smf = smf; C_.dA = dA;
                C_.p_up = p_up; C_.p_dn = p_dn; C_.ok_until = ok_until; C_.rows_ld = rows_ld; C_.cols_ld = cols_ld; C_.diags_ld = diags_ld;
                C_.as0 = as0; C_.as1 = as1; C_.as2 = as2;
#ifdef PG_PIPE_STATS
                for (int k = 0; k < 5; ++k) { C_.st_cls_t[k] = st_cls_t[k]; C_.st_cls_n[k] = st_cls_n[k]; }
                for (int k = 0; k < 10; ++k) { C_.st_poll_t[k] = st_poll_t[k]; C_.st_poll_n[k] = st_poll_n[k]; }
                for (int k = 0; k < 4; ++k) C_.st_w[k] = st_w[k];
#endif
                if (!STRIP && (dA.s4 & 15) == 5) widest_run7(C_);
                else if (!STRIP && (dA.s4 & 15) == 4) { if (dA.s4 & (1 << 19)) wide_run7(C_); else wide_run(C_); } else hot_run<STRIP>(C_);
#ifdef PG_PIPE_STATS
                for (int k = 0; k < 5; ++k) { st_cls_t[k] = C_.st_cls_t[k]; st_cls_n[k] = C_.st_cls_n[k]; }
                for (int k = 0; k < 10; ++k) { st_poll_t[k] = C_.st_poll_t[k]; st_poll_n[k] = C_.st_poll_n[k]; }
                for (int k = 0; k < 4; ++k) st_w[k] = C_.st_w[k];
#endif
                d = __builtin_amdgcn_readfirstlane(C_.d); row = C_.row;
                r1x = C_.px; r1y = C_.py; r1m = C_.pm; r2x = C_.cx; r2y = C_.cy; r2m = C_.cm;
                ca = C_.ca; cb = C_.cb; smf = C_.smf; dA = uniform_i8(C_.dA);
                p_up = __builtin_amdgcn_readfirstlane(C_.p_up); p_dn = __builtin_amdgcn_readfirstlane(C_.p_dn);
                ok_until = __builtin_amdgcn_readfirstlane(C_.ok_until); rows_ld = __builtin_amdgcn_readfirstlane(C_.rows_ld);
                cols_ld = __builtin_amdgcn_readfirstlane(C_.cols_ld); diags_ld = __builtin_amdgcn_readfirstlane(C_.diags_ld);
                as0 = __builtin_amdgcn_readfirstlane(C_.as0); as1 = __builtin_amdgcn_readfirstlane(C_.as1); as2 = __builtin_amdgcn_readfirstlane(C_.as2);
                // step() reloads the row records; nothing of lane 0's operand is prefetched
                have = false; lo_prev = -1; hi_prev = -2; nb_valid = false;
                slot = d % PRK; slot1 = (d + PRK - 1) % PRK; stg = d % PST;
                pp = psc + d;
            } else if (!TAB_LDS && (dA.s4 & 0x1c) == 0x10 && (dA.s4 & 3) != 3 && d + 1 < sleep) {      // class 0..2 and the next one too
                // two hot steps: the register sets swap roles and are back in place afterwards
                step(std::true_type(), d, dA, dB, r1x, r1y, r1m, r2x, r2y, r2m, r3x, r3y, r3m, r4x, r4y, r4m, ca, cb);
                step(std::true_type(), d + 1, dB, dA, r4x, r4y, r4m, r3x, r3y, r3m, r2x, r2y, r2m, r1x, r1y, r1m, cb, ca);
                d += 2;
            } else {
                step(std::false_type(), d, dA, dB, r1x, r1y, r1m, r2x, r2y, r2m, r3x, r3y, r3m, r4x, r4y, r4m, ca, cb);
                // a single step leaves the sets rotated by one: put them back
                { const pg_i8 t = dA; dA = dB; dB = t; }
                { const pg_i4 t = ca; ca = cb; cb = t; }
                double t;
                t = r1x; r1x = r4x; r4x = t;  t = r1y; r1y = r4y; r4y = t;  t = r1m; r1m = r4m; r4m = t;
                t = r2x; r2x = r3x; r3x = t;  t = r2y; r2y = r3y; r3y = t;  t = r2m; r2m = r3m; r3m = t;
                d += 1;
            }
        }
    }
#ifdef PG_PIPE_STATS
    if (lane == 0 && 3 * (Lx + Ly) >= 4096 && PG_STATS_MINE(job, flags)) {     // the counters borrow the tail of the trace buffer: long jobs only
        PG_GLOBAL int *o = (PG_GLOBAL int *)job->trace + 3 * (Lx + Ly) - 200 + 12 * wave;
        o[0] = st_n;
        for (int k = 0; k < 7; ++k) o[1 + k] = (int)(st_acc[k] >> 4);
        PG_GLOBAL int *q = (PG_GLOBAL int *)job->trace + 3 * (Lx + Ly) - 400 + 12 * wave;
        for (int c = 0; c < 5; ++c) { q[c] = st_cls_n[c]; q[5 + c] = (int)(st_cls_t[c] >> 8); }
        PG_GLOBAL int *u = (PG_GLOBAL int *)job->trace + 3 * (Lx + Ly) - 600 + 4 * wave;
        for (int c = 0; c < 4; ++c) u[c] = (int)(st_w[c] >> 8);
        PG_GLOBAL int *v = (PG_GLOBAL int *)job->trace + 3 * (Lx + Ly) - 800 + 20 * wave;
        for (int c = 0; c < 10; ++c) { v[c] = st_poll_n[c]; v[10 + c] = (int)(st_poll_t[c] >> 8); }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // every wave reports an abort it sees when it leaves (the host staged 0): wave 0 may be done long before a
    // later wave's wait runs into its spin limit, and after an abort every wait passes, so a wave that continued
    // on unsynchronised operands still ends here with the flag set
    if (lane == 0) {
        const int aborted = flag_load(&PM.abort_flag);
        if (aborted != 0) report_fill_status(job, aborted);
    }
}

template __global__ void pg_fill_pipe<true, false>(const PgDevJob *, const int *, unsigned, int);
template __global__ void pg_fill_pipe<false, false>(const PgDevJob *, const int *, unsigned, int);
template __global__ void pg_fill_pipe<true, true>(const PgDevJob *, const int *, unsigned, int);
template __global__ void pg_fill_pipe<false, true>(const PgDevJob *, const int *, unsigned, int);
