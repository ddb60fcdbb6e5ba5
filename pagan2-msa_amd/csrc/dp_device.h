// dp_device.h -- device-side job descriptor shared by the kernels and the C-ABI layer.
//
// HBM layout of one alignment (all arrays 256-B aligned inside one arena per batch):
//
//   inputs   left/right graph CSR (state, bwd_off, bwd_src, bwd_logw) -- 4 B per site + 8 B per edge
//            model table S*S f32
//            diagonal index: imin[d], imax[d] (i32), doff[d] (i64) for d = 0 .. Lx+Ly-2
//   outputs  score[cells][3] f64  (X, Y, M)   } DIAGONAL-MAJOR: cell (i,j), d=i+j, lives at
//            bp[cells][3]    u32  (X, Y, M)   } doff[d] + (i - imin[d]) -- one anti-diagonal is
//                                                contiguous, so a wavefront's loads/stores coalesce;
//                                                the three states of a cell sit together (24 B + 12 B):
//                                                one address computation and three store
//                                                instructions per cell
//            trace[Lx+Ly] {i,j,mat|slots}      visited cells of the Viterbi path, end to start
//            endcell                           the end-corner result (max_end)
//
// A back-pointer packs what Matrix_pointer (basic_alignment.h:33-50) records for the Viterbi
// path: the predecessor matrix and WHICH bwd edges were taken, as slots into the two sites'
// bwd lists (x_ind/y_ind/x_edge_ind/y_edge_ind follow from the slot via the CSR arrays):
//   bits 0-1   from: 0 = X, 1 = Y, 2 = M, 3 = none (cell kept score -inf / matrix -1)
//   bit  2     the left edge taken starts at site i-1   } lets the traceback step without
//   bit  3     the right edge taken starts at site j-1  } touching the graph arrays
//   bits 4-17  slot in the left site's bwd list   (M and X cells)
//   bits 18-31 slot in the right site's bwd list  (M and Y cells)
//
// Traceback (dp_kernels.hip, pg_trace_*): the path is cut at "boundary" diagonal pairs
// {k*PG_SEG, k*PG_SEG-1}; ttab[tb[k] + 3*c + s] describes where the chase that starts in
// state s of the c-th cell of boundary k arrives at boundary k-1 and after how many cells.
#pragma once
#include <stdint.h>

#define PG_BP_NONE 3u
#define PG_BP_ADJL 4u
#define PG_BP_ADJR 8u
#define PG_MAX_SLOT 16383
#define PG_SEG 256           // diagonals per traceback segment
#define PG_BP_DIAGS 64        // diagonals per workgroup of the back-pointer pass (pg_backptr)
#define PG_BP_CELLS 1024      // ... and cells of each of them (grid.z covers the rest of a wide diagonal)
#define PG_STATUS_PATH_CHECK 3 // endcell[0]: a visited cell's stored score / back-pointer differs from its re-evaluation (pg_trace_check)
#define PG_FLAG_SCORE_CHECK 0x200u   // kernel flag: pg_backptr / the follower workgroups compare every cell's re-evaluated scores with the stored ones
#define PG_FLAG_STRIPS_SPREAD 0x10000u   // kernel flag (pg_fill_pipe<., true>): a job's strips run on any XCD (strip_feeder does not ask where the strip above runs)
#define PG_FILL_SCORE_MISMATCH 0x7c  // fill_status: ... and found a difference (pagan_batch_fetch runs the batch once more without followers, then reports)
#define PG_FILL_OTHER_XCD 0x20000000  // fill_status, sticky: a row strip found the strip above on another XCD (the abort tags stay below bit 28)
#define PG_FOLLOW_CHUNK 16    // diagonals a follower wave of pg_fill_pipe claims at a time (dp_pipe.hip, pipe_follower)

// Geometry of the banded fill kernel (dp_pipe.hip) that the host-side planner (dp_abi.hip:
// classify_diagonals, schedule_waves) has to agree with.
#define PG_PIPE_WIDTH 242        // widest diagonal computed in the lanes' registers (256 lanes - PG_PIPE_REACH)
#define PG_PIPE_REACH 14         // a cell may read PG_PIPE_REACH-1 diagonals back in the LDS ring: the whole ring but the
                                 // row being written.  Older operands come from L2.  Since the multi-edge candidates are
                                 // evaluated by assist waves ahead of the compute waves (dp_pipe.hip), an L2 read is off the
                                 // critical path, and five diagonals of the ring (20 in round 1) became the staging slots
                                 // through which the assist waves hand their results over.
#define PG_PIPE_RING 14          // ring depth in diagonals: a wave stays awake this long after its last cell
#define PG_HIST_SLOTS 3           // far histories (dp_abi.hip, plan_far_hist; dp_pipe.hip, PipeSmem::hist): lines of 64 cells
#define PG_HIST_MAX_SPAN 44      // longest edge a history line serves: an entry lives 64 steps, the reader comes k + 1 steps after the
                                 // writer, and the writer's wave may be up to a ring's depth ahead of the reader's
#define PG_PIPE_ASSIST 3         // assist waves: wave a takes the multi-edge cells of the diagonals d with d % 3 == a
#define PG_PIPE_STAGE 3          // staging slots (d % 3, one per assist wave): a diagonal is staged at most two ahead of the compute waves
#define PG_PIPE_WAKE 6           // ... and wakes this many diagonals before its first one (operand prefetch pipeline)
#define PG_PIPE_WINDOW 432       // widest diagonal the kernel's site-record windows (512 sites) still cover: the loader keeps the records from 8
                                 // sites before the slowest wave's first row (column) and stages at most 440 sites past it (round 5: 352 before,
                                 // when it also staged a fixed 64 diagonals ahead whatever the width; cfg4's third level alone has ~2,000
                                 // diagonals of 353 .. 922 cells, each 28 k cycles as a class 5 step against ~6 k as a wide step)
#define PG_PIPE_WINDOW_A 352     // ... a wide run whose diagonals all stay at or below this uses the deeper of the two wide-ring geometries
#define PG_PIPE_EDGE_CAP 1024    // bwd edges of any PG_RING_SITE_SPAN consecutive sites must fit the LDS edge window
#define PG_PIPE_SITE_EDGES 126   // bwd edges per site (7-bit count in the site record)

#define PG_TILE 64               // dp_tiles.hip: side of the square tiles a wide matrix is cut into (one wave each)
#define PG_TILE_EDGES 512        // ... and how many bwd edges the sites of one tile row / tile column may have together

struct PgDevJob {
    int Lx, Ly;              // matrix dimensions (sites minus the stop site)
    int nd;                  // number of anti-diagonals = Lx + Ly - 1
    int S;                   // model states
    float go, ge, gE, ng;    // log_gap_open, log_gap_ext, log_gap_end_ext, log_non_gap
    // left / right graph (device pointers)
    const int *stL, *offL, *srcL; const float *lwL;
    const int *stR, *offR, *srcR; const float *lwR;
    const float *table;      // [a + b*S]
    // band in diagonal form
    const int *imin, *imax;
    const long long *doff;
    const int *dsc;          // [nd][4] = imin, imax, doff low, doff high: one 16-byte scalar load per diagonal
    const int *psc;          // [nd+1][8] (dp_pipe.hip; null for jobs of the other kernels): imin, imax, byte offset of the
                             //   diagonal's first score (= 24 * doff; low, high), class | next-is-class-0/1 << 4 | resident-mask << 5, doff low,
                             //   doff high, diagonal the downstream wave must have completed before this one
                             //   overwrites its ring row.  class: 0 simple, 1 multi-edge, 2 multi-edge with far edges, 3 general,
                             //   4 wide; bit a of the mask: diagonal d-a was computed by the lanes (is in the LDS ring)
    const int *sched;        // dp_pipe.hip: [4] offsets, then per compute wave its awake intervals a0,b0,a1,b1,...,nd,nd
    const unsigned char *hfL, *hfR;   // dp_pipe.hip, far histories: a flag byte per left / right site (dp_abi.hip, plan_far_hist); null: none
    int *fill_status;        // [1] 0 = filled; nonzero = the fill kernel abandoned a wait (internal error)
    long long cells;
    // outputs
    double *sc;              // [cells][3], state index = PAGAN_X_MAT / Y_MAT / M_MAT
    unsigned *bp;            // [cells][3]
    int *trace;              // [3 * (Lx+Ly)]
    int *endcell;            // [8]: status, matrix, x_ind, y_ind, slot_l, slot_r, n_trace, n_segments ; score in endscore
    double *endscore;        // [1]
    // segmented traceback
    int n_bound;             // K: boundaries k = 1..K at diagonals k*PG_SEG
    const int *tb;           // [K+2] first table entry of boundary k (tb[K+1] = total)
    int *ttab;               // [total][8]: exit i, exit j, exit matrix | kind<<2, cells visited (-1: dead entry),
                             //             the exit cell's own entry (absolute index, -1: none), 3 x pad
    int *segs;               // [2K+8][6]: start i, j, matrix, cells, output offset, pad
    // back-pointers behind the banded fill (dp_pipe.hip, pipe_follower; null for jobs of the other kernels): zeroed before
    // every launch
    int *follow;             // [4]: diagonals whose scores have landed in L2 + 1, the fill workgroup's XCC id + 1, next
                             //      chunk of PG_FOLLOW_CHUNK diagonals to claim, pad
    unsigned char *bp_done;  // [ceil(nd / PG_FOLLOW_CHUNK)] 1: the chunk's back-pointers are written (pg_backptr skips it)
    // Row strips of a wide job on the banded kernel (dp_pipe.hip, pg_fill_pipe<true, true>).  A strip is a
    // job of its own -- PG_STRIP_ROWS rows of the parent's matrix, its own diagonal descriptors (psc, indexed from d_first: the
    // pointer is moved back so that psc[d] works), schedule and follow words -- that shares the parent's graphs and output arrays.
    int is_strip;            // 0: not a strip (every field below is then 0 / null)
    int strip_row0;          // first row of the strip
    int d_first;             // first diagonal the strip's waves look at (descriptors exist from here to nd)
    int feed_wave;           // the compute wave that has no rows of the strip and feeds the 64 rows above it into the ring; -1: none (first strip)
    int col_first;           // first right-graph site whose record the loader stages (a multiple of 64)
    int prev_nd;             // the previous strip's nd (its last diagonal + 1): what its landed counter ends at
    const int *prev_follow;  // the previous strip's follow words (null: first strip)
    const int *pdsc;         // the PARENT's dsc array: the whole band's rows per diagonal, for operands in other strips
};
#ifndef PG_STRIP_ROWS
#define PG_STRIP_ROWS 192        // rows of a strip: three of dp_pipe.hip's four compute waves (the fourth feeds the rows above)
#endif
#define PG_STRIP_TERM 8          // psc class bit of a strip diagonal that holds a cell of the first / last column (gap extension differs)
