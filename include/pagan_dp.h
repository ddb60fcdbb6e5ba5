/*
 * pagan_dp.h -- C ABI of the MI355X pairwise graph-vs-graph Viterbi aligner.
 *
 * This is the drop-in seam between a guide-tree walk (the reference's
 * Node::align_sequences_this_node, src/main/node.cpp:52-192) and the pairwise
 * aligner it calls (Viterbi_alignment::align, src/main/viterbi_alignment.cpp:187-465;
 * interface in src/main/viterbi_alignment.h:214-223).  The reference has no
 * FFI layer; these entry points are what a binding at that seam would call.
 *
 * Conventions:
 *   - a graph has sites 0..n_sites-1; site 0 is the start site, site
 *     n_sites-1 the stop site (src/main/sequence.cpp:155-158,292-301);
 *   - the DP matrices are Lx x Ly with Lx = left.n_sites-1, Ly = right.n_sites-1
 *     (viterbi_alignment.cpp:229-247); the stop sites only enter the end corner;
 *   - bwd edge lists are in the order Site::get_first_bwd_edge/get_next_bwd_edge
 *     iterate them (src/main/sequence.h:395-417) -- the order decides ties;
 *   - matrix labels follow enum Matrix_pt {x_mat=0,y_mat=1,m_mat=2}
 *     (src/main/basic_alignment.h:107);
 *   - path states follow Site::Path_state (src/main/sequence.h:229):
 *     matched=2, xgapped=3, ygapped=4, xskipped=5, yskipped=6.
 *
 * All pointers are host pointers borrowed for the duration of the call unless
 * stated otherwise.  No function throws or exits; errors are negative codes.
 */
#ifndef PAGAN_DP_H
#define PAGAN_DP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error / status codes ------------------------------------------------ */
#define PAGAN_OK              0
#define PAGAN_E_ARG          -1   /* null pointer / inconsistent sizes                 */
#define PAGAN_E_GRAPH        -2   /* malformed CSR graph (offsets, edge direction)     */
#define PAGAN_E_BAND         -3   /* band arrays wrong length or not monotone          */
#define PAGAN_E_MODEL        -4   /* state outside the model table                     */
#define PAGAN_E_NODEVICE     -5   /* no HIP device / HIP runtime error                 */
#define PAGAN_E_NOMEM        -6   /* device or host allocation failed                  */
#define PAGAN_E_INTERNAL     -7
/* result.status: alignment-level outcome (the call itself still returns PAGAN_OK)    */
#define PAGAN_DP_REACHED      0
#define PAGAN_DP_UNREACHABLE  1   /* end corner score is -inf ("anchored alignment
                                     failed", viterbi_alignment.cpp:298-323); caller
                                     decides whether to retry without a band          */

/* ---- option bits (the Settings flags that change the DP) --- */
#define PAGAN_OPT_NO_TERMINAL_EDGES        1u  /* --no-terminal-edges (VA:866,877)      */
#define PAGAN_OPT_NO_REDUCED_TERMINAL_PEN  2u  /* --no-reduced-terminal-penalties
                                                  (basic_alignment.h:627-628)          */

enum { PAGAN_X_MAT = 0, PAGAN_Y_MAT = 1, PAGAN_M_MAT = 2 };
enum { PAGAN_MATCHED = 2, PAGAN_XGAPPED = 3, PAGAN_YGAPPED = 4,
       PAGAN_XSKIPPED = 5, PAGAN_YSKIPPED = 6 };

/* One child Sequence flattened to CSR: replaces the Site/Edge read accessors
 * the DP uses (src/main/sequence.h:80-96,262,395-417,831).                     */
typedef struct pagan_graph {
    int32_t        n_sites;   /* including start and stop site                        */
    int32_t        n_edges;   /* size of the edge-id space (Sequence::edges_length()) */
    const int32_t *state;     /* [n_sites] Site::character_state (-1 at the ends)     */
    const int32_t *bwd_off;   /* [n_sites+1] offsets into bwd_*                       */
    const int32_t *bwd_src;   /* [bwd_off[n_sites]] Edge::start_site_index            */
    const float   *bwd_logw;  /* [..] Edge::log_posterior_weight (float, logf)        */
    const int32_t *bwd_eid;   /* [..] Edge::index                                     */
} pagan_graph;

/* Per-alignment Evol_model view (src/utils/evol_model.h:59-63,78-88).          */
typedef struct pagan_model {
    int32_t      n_states;        /* S: 15 DNA, 211 protein, ...                      */
    const float *log_score;       /* [S*S], log_score(a,b) = log_score[a + b*S]
                                     (Db_matrix::g, src/utils/db_matrix.h:76-83)      */
    float        log_gap_open;    /* Evol_model::log_id_prob                          */
    float        log_gap_ext;     /* Evol_model::log_ext_prob                         */
    float        log_gap_end_ext; /* Evol_model::log_end_ext_prob                     */
    float        log_non_gap;     /* Evol_model::log_match_prob                       */
} pagan_model;

/* The "tunnel" (Viterbi_alignment::upper_bound/lower_bound, filled by
 * define_tunnel, viterbi_alignment.cpp:148-164).  Row i of the left graph may
 * use columns max(0,upper[i]) .. min(lower[i],Ly-1)
 * (src/utils/tunnel_matrix.h:194).  n must be >= Lx; entries beyond Lx-1 are
 * ignored (find_anchors.cpp:373 emits length1+1 of them).                       */
typedef struct pagan_band {
    int32_t        n;
    const int32_t *upper;
    const int32_t *lower;
} pagan_band;

typedef struct pagan_opts {
    uint32_t flags;    /* PAGAN_OPT_*                                                 */
    int32_t  device;   /* HIP device ordinal, -1 = current                            */
} pagan_opts;

/* One alignment column = one Site of the parent sequence
 * (Basic_alignment::create_ancestral_sequence, basic_alignment.cpp:61-179).     */
typedef struct pagan_col {
    int32_t left;        /* left child site index, -1 if none                         */
    int32_t right;       /* right child site index, -1 if none                        */
    int32_t path_state;  /* PAGAN_MATCHED ... PAGAN_YSKIPPED                          */
} pagan_col;

/* What Viterbi_alignment::align leaves behind for build_ancestral_sequence:
 * the path (viterbi_alignment.cpp:1038-1189), the end-corner cell `max_end`
 * (viterbi_alignment.cpp:289-296) and the child edges marked used
 * (Edge::is_used(true) at viterbi_alignment.cpp:1054-1057,1079-1101,1128,1155). */
typedef struct pagan_result {
    int32_t    status;        /* PAGAN_DP_REACHED / PAGAN_DP_UNREACHABLE              */
    double     score;         /* max_end.score: the node's Viterbi log score          */
    int32_t    end_matrix;    /* max_end.matrix                                       */
    int32_t    end_x, end_y;  /* max_end.x_ind / y_ind                                */
    int32_t    end_x_edge, end_y_edge;
    int32_t    n_cols;
    pagan_col *cols;          /* [n_cols] forward order, skip columns included        */
    int32_t    n_left_used;
    int32_t   *left_used;     /* edge ids of the left graph marked used (ascending)   */
    int32_t    n_right_used;
    int32_t   *right_used;
    int64_t    cells;         /* in-band DP cells filled                              */
    double     fill_ms;       /* device time of the fill kernel (HIP events)          */
    double     trace_ms;      /* device time of end corner + traceback                */
} pagan_result;

typedef struct pagan_job {
    const pagan_graph *left;
    const pagan_graph *right;
    const pagan_model *model;
    const pagan_band  *band;   /* NULL = full matrix (--no-anchors)                   */
} pagan_job;

/* ---- entry points --------------------------------------------------------- */

/* Replaces Viterbi_alignment::align up to (not including) build_ancestral_sequence
 * (viterbi_alignment.cpp:187-384).  Fill, end corner and traceback run on the GPU.
 * Fails with PAGAN_E_NODEVICE when no HIP device is usable: there is no CPU path. */
int pagan_dp_align(const pagan_graph *left, const pagan_graph *right,
                   const pagan_model *model, const pagan_band *band,
                   const pagan_opts *opts, pagan_result *out);

/* Same, for n independent alignments in one launch (the ready nodes of one
 * guide-tree level, node.cpp:227-285).  out[k] is filled for every job.         */
int pagan_dp_align_batch(int32_t n, const pagan_job *jobs,
                         const pagan_opts *opts, pagan_result *out);

void pagan_result_free(pagan_result *r);

/* Viterbi_alignment::get_predicted_memory_consumption (viterbi_alignment.cpp:555-568)
 * restated for the device layout: bytes of HBM one alignment needs.             */
int64_t pagan_dp_predict_bytes(int32_t left_sites, int32_t right_sites,
                               const pagan_band *band);

/* Number of in-band cells, SURVEY.md s.8(d): sum_i (min(lower,Ly-1)-max(upper,0)+1). */
int64_t pagan_dp_count_cells(int32_t left_sites, int32_t right_sites,
                             const pagan_band *band);

int pagan_dp_device_count(void);
int pagan_dp_select_device(int32_t device);

/* ---- resident-batch interface (used by bench.py and the tree driver) ------
 * Uploads the jobs once, keeps them in HBM, and lets the caller re-run the hot
 * path on the resident inputs.  pagan_batch_run leaves results on the device;
 * pagan_batch_fetch copies them out.                                            */
typedef struct pagan_batch pagan_batch;

int  pagan_batch_create(int32_t n, const pagan_job *jobs, const pagan_opts *opts,
                        pagan_batch **out);
int  pagan_batch_run(pagan_batch *b);                 /* fill + traceback, async  */
int  pagan_batch_sync(pagan_batch *b);
int  pagan_batch_fetch(pagan_batch *b, pagan_result *out /* [n] */);
/* device milliseconds of the last run: [0]=fill kernel, [1]=end corner+traceback  */
int  pagan_batch_last_ms(pagan_batch *b, double ms[2]);
/* the same per kernel: ms[0] banded fill (pg_fill_pipe), ms[1] pg_backptr, ms[2] tiled fill (pg_fill_tiles_flow; it runs
 * beside the banded kernels of a mixed batch), ms[3] HBM wavefront fill, ms[4] end corner + traceback, ms[5] the whole
 * fill; -1 where the batch launched no such kernel */
int  pagan_batch_last_ms_detail(pagan_batch *b, double ms[6]);
int64_t pagan_batch_cells(const pagan_batch *b);
void pagan_batch_destroy(pagan_batch *b);
/* diagnostic builds only: raw bytes of job k's device trace buffer                    */
int  pagan_batch_debug_trace(pagan_batch *b, int32_t k, void *dst, int64_t bytes);
/* diagnostic, host only (no device needed): the plan the banded fill kernel would get for a job --
 * cls[Lx+Ly-1] = class of every anti-diagonal (0 simple, 1 multi-edge, 2 multi-edge with far edges,
 * 3 general, 4 wide) and the four compute waves' awake intervals (layout: dp_device.h, sched)      */
int  pagan_dp_debug_plan(const pagan_graph *left, const pagan_graph *right, const pagan_band *band,
                         uint8_t *cls, int32_t n_cls, int32_t *sched, int32_t sched_cap, int32_t *sched_len,
                         int32_t *lead_req /* [n_cls] or NULL: diagonal the downstream wave must have completed
                                              before diagonal d may overwrite its row of the LDS ring, -1 none */);
/* diagnostic, host only: the far histories of a banded job (dp_abi.hip, plan_far_hist) with the classes that go with them --
 * hfL[Lx], hfR[Ly]: flag byte of every left / right site (bit 7 reads a history line, bits 0-1 which; bit 6 writes one, bits
 * 4-5 which); hbit[Lx+Ly-1]: 1 where a reader or writer has a cell; cls as pagan_dp_debug_plan's (which plans WITHOUT
 * histories: its classes are those of PAGAN_DP_HIST=0).  Returns the number of far sites served (>= 0) or an error (< 0).  */
int  pagan_dp_debug_far(const pagan_graph *left, const pagan_graph *right, const pagan_band *band,
                        uint8_t *hfL, uint8_t *hfR, uint8_t *hbit, uint8_t *cls);
/* diagnostic, host only: the row strips a wide job would be filled as (dp_pipe.hip, strip_feeder), whether
 * or not the library would route it that way (max_sites <= 0: no bound on the multi-edge sites of a diagonal).  Returns the
 * number of strips (0: the bound refused the job), < 0 on error.  strips[6 * k] = first row, last row, first diagonal, last
 * diagonal + 1, feeder wave (-1: none), first column staged; for strip k and diagonal d in [first, last + 1) the entry
 * desc[4 * (desc_off[k] + d - first)] = first row of the strip on d, last row (first - 1: none), the cell index of the first
 * row's score in the JOB's arrays, class (0 simple .. 3 general).  desc_off[k] = entries before strip k's (n + 1 values). */
int  pagan_dp_debug_strips(const pagan_graph *left, const pagan_graph *right, const pagan_band *band, int32_t max_sites,
                           int32_t *strips /* [6 * cap] */, int32_t cap, int64_t *desc_off /* [cap + 1] */,
                           int64_t *desc /* [4 * desc_cap] */, int64_t desc_cap);
/* diagnostic, host only: the tiles the wide-matrix kernel (dp_tiles.hip) would be launched over for this
 * job, as (tile row, tile column) pairs of side *tile_side; returns the number of tiles (pairs written:
 * min(count, cap)), 0 when the job cannot be tiled (a tile's sites have too many bwd edges), < 0 on error */
int  pagan_dp_debug_tiles(const pagan_graph *left, const pagan_graph *right, const pagan_band *band,
                          int32_t *tiles /* [2 * cap] */, int32_t cap, int32_t *tile_side);
/* diagnostic, host only: 1 if the n tiles (tile row, tile column pairs) form a staircase -- every tile row a contiguous
 * run of columns, first and last column never falling, no empty row between two rows, consecutive rows touching -- which
 * is when the tiled kernel's dataflow launch orders a tile behind its three neighbours only; 0 otherwise; < 0 on error */
int  pagan_dp_debug_tiles_staircase(const int32_t *tiles, int32_t n);
/* diagnostic, host only: what the library does about sites without a live predecessor (from 5 % of a job's sites
 * on it aligns the compacted graphs and maps the path back; PAGAN_DP_COMPACT=0 switches that off).  keep_*[t] = the
 * caller's site of compacted site t ([n_sites] at most); slot_*[e] = for the compacted graph's bwd edges in order, the
 * position of the edge in the caller's list of its site ([bwd_off[n_sites]] at most); upper / lower = the band over the
 * compacted matrices ([n_sites of left] at most, NULL: not wanted); n_out[4] = kept left sites, kept right sites, kept
 * left edges, kept right edges.  Any output pointer may be NULL. */
int  pagan_dp_debug_compact(const pagan_graph *left, const pagan_graph *right, const pagan_band *band, int32_t *keep_left,
                            int32_t *keep_right, int32_t *slot_left, int32_t *slot_right, int32_t *upper, int32_t *lower,
                            int32_t *n_out);
/* diagnostic (host only): the fill kernel pagan_batch_create would give this job -- 0 pg_fill_pipe (model table in LDS),
 * 1 pg_fill_pipe (large table), 2 pg_fill_tiles_flow, 3 pg_fill_wavefront; n_out[0] = 1 when its dead sites are taken
 * out first, n_out[1] = cells of its widest anti-diagonal (n_out may be NULL); negative: the job's validation error  */
int  pagan_dp_debug_route(const pagan_graph *left, const pagan_graph *right, const pagan_model *model, const pagan_band *band,
                          int32_t *n_out);
/* diagnostic: job k's scores, [cells][3] doubles (X, Y, M), diagonal-major                */
int  pagan_batch_debug_scores(pagan_batch *b, int32_t k, double *dst, int64_t count);
/* diagnostic: job k's back-pointers, [cells][3] packed words (X, Y, M), diagonal-major   */
int  pagan_batch_debug_backptrs(pagan_batch *b, int32_t k, uint32_t *dst, int64_t count);
/* diagnostic: overwrite all device outputs with 0xFF (NaN scores) before a run           */
int  pagan_batch_debug_poison(pagan_batch *b);
/* Diagnostic: counts[0] = chunks of 16 diagonals of job k whose back-pointers the follower workgroups of the banded fill
 * wrote while the fill was running, counts[1] = all chunks of the job (0 of 0 for a job of another kernel).          */
int  pagan_batch_debug_followed(pagan_batch *b, int32_t k, int32_t *counts);
/* Test hook for the path check (every cell the traceback visited is re-evaluated from the stored scores and compared with
 * the stored score and back-pointer, always; a difference makes pagan_batch_fetch run the batch once more with every
 * back-pointer written after the fill, and report PAGAN_E_INTERNAL if it stays; PAGAN_DP_RERUN=0: report at once):
 * state `vit` (0 X, 1 Y, 2 M) of cell (i, j) of job k gets `word` as its back-pointer after the NEXT run's fill, once.  */
int  pagan_batch_debug_poke_bp(pagan_batch *b, int32_t k, int32_t i, int32_t j, int32_t vit, uint32_t word);
/* how often pagan_batch_fetch has run this batch again after a failed path check           */
int  pagan_batch_debug_reruns(pagan_batch *b);

/* The library keeps up to two idle device arenas per device and a few host staging buffers for the next
 * batch (a level of a tree walk is followed by the next; freeing and re-allocating GBs costs tens of ms).
 * pagan_dp_release_cache frees them; pagan_dp_cached_device_bytes says how much of a device's memory they hold
 * (memory a caller sizing batches from hipMemGetInfo may count as free). */
void    pagan_dp_release_cache(void);
int64_t pagan_dp_cached_device_bytes(int32_t device);

/* ---- forward/backward full probability, posteriors, path sampling -----------------------------
 * The reference's compute_full_score pass (--full-probability, --sample-path; basic_alignment.h:621-625,
 * viterbi_alignment.cpp:329-371, 740-854, 975-1034, 1571-1662, 2259-2305; sampling :1193-1322), in log
 * space on the GPU: the reference multiplies raw probabilities and under/overflows on long inputs.      */
typedef struct pagan_model_prob {   /* Evol_model's probability-space accessors (evol_model.h:70-88)       */
    int32_t      n_states;
    const float *score;             /* [S*S] Evol_model::score(a,b) = score[a + b*S] (charPr as float)     */
    float        gap_open;          /* Evol_model::gap_open()  = id_prob                                   */
    float        gap_ext;           /* Evol_model::gap_ext()   = ext_prob                                  */
    float        non_gap;           /* Evol_model::non_gap()   = match_prob;  gap_close() is 1             */
} pagan_model_prob;

typedef struct pagan_fb pagan_fb;   /* forward and backward matrices of one alignment, resident in HBM     */

/* Runs both passes.  left/right must stay valid until pagan_fb_destroy (sample_path reads them).          */
int  pagan_fb_run(const pagan_graph *left, const pagan_graph *right, const pagan_model_prob *model,
                  const pagan_band *band, const pagan_opts *opts, pagan_fb **out);
/* The same pass for n alignments at once (the reference runs compute_full_score node by node, viterbi_alignment.cpp:329-371; a
 * caller that holds several independent node pairs -- a level of the guide tree -- hands them over together): the forward sweeps
 * of all wide pairs in ONE launch and the backward sweeps in another, so that how many run side by side is what the device holds,
 * not what the runtime's hardware queues allow.  band may be null (no pair has a band) or hold null entries.  out[k] as from
 * pagan_fb_run (pagan_fb_kernel_ms: the launches' times at the batch's first wide pair, 0 at the others); on an error nothing
 * is handed back.                                                                                                        */
int  pagan_fb_run_batch(int32_t n, const pagan_graph *const *left, const pagan_graph *const *right,
                        const pagan_model_prob *const *model, const pagan_band *const *band, const pagan_opts *opts,
                        pagan_fb **out);
/* log of max_end.fwd_score ("full probability", VA:1562-1563) and of match[0][0].bwd_score (VA:345-349);
 * the reference checks their ratio (VA:351-355).                                                          */
int  pagan_fb_totals(const pagan_fb *fb, double *log_fwd, double *log_bwd, int64_t *cells);
/* device time of the two sweeps of pagan_fb_run, milliseconds: ms[0] pg_fb_forward, ms[1] pg_fb_backward */
int  pagan_fb_kernel_ms(const pagan_fb *fb, double ms[2]);
/* workgroups of the pair's forward sweep: 1 = the one-workgroup kernel (a barrier per cell diagonal), > 1 = 64 x 64 blocks,
 * a wave each (wide matrices, and -- round 5 -- tunnels of 4,096 cell diagonals or more), 0 = the LDS-ring sweeps (two
 * plain sequences, widest diagonal <= 1,024 cells: one workgroup, lane = row mod B); diagnostic, for the tests                          */
int  pagan_fb_groups(const pagan_fb *fb);
/* which: 0 log forward, 1 log backward, 2 posterior (compute_posterior_score, VA:1029-1034);
 * dst [Lx][Ly][3] row-major, states X, Y, M; outside the tunnel -inf / 0.                                  */
int  pagan_fb_dump(pagan_fb *fb, int32_t which, double *dst);
/* posterior of n cells given as (state, i, j) triples                                                     */
int  pagan_fb_posterior_cells(pagan_fb *fb, int32_t n, const int32_t *cells, double *post);
/* sample_new_path (VA:1193-1322): u[k] in [0,1) replaces rand()/(RAND_MAX+1), one per step, the end corner
 * first (at most Lx+Ly+1 are consumed).  `out` has the shape of a Viterbi result (free with
 * pagan_result_free); visited (optional, 3*(Lx+Ly) ints): the path's cells end -> start as (i, j, state).  */
int  pagan_fb_sample_path(pagan_fb *fb, const double *u, int32_t n_u, pagan_result *out,
                          int32_t *visited, int32_t *n_visited);
void pagan_fb_destroy(pagan_fb *fb);

const char *pagan_dp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PAGAN_DP_H */
