// oracle_dp.cpp -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's
// pairwise graph-vs-graph Viterbi alignment, used as the checker for the HIP path.
//
// PARITY UNPINNED: the reference (ariloytynoja/pagan2-msa @ 2024_08_07) ships no
// tests, fixtures or golden vectors for this path, and cannot be built in this
// image (every translation unit on the path includes Boost program_options via
// utils/settings.h; Boost is absent and stand-in headers are not allowed).  This
// file therefore restates the algorithm from the reference's source text, one
// function per reference function, each citing the file:line it follows.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
// into this file.  The shipped library (libpagan_dp.so) never links it.
//
// Restated functions ("VA" = src/main/viterbi_alignment.cpp,
// "BA.h" = src/main/basic_alignment.h, "TM" = src/utils/tunnel_matrix.h):
//   Cell                    <- struct Matrix_pointer            BA.h:33-50
//   Band / at()             <- Tunnel_matrix / Tunnel_slice::at TM:85-98,185-233
//   bigger()                <- first_is_bigger                  BA.h:449-462
//   open_pen()              <- get_log_gap_open_penalty         BA.h:490-513
//   fill_cell()             <- compute_fwd_scores               VA:856-971
//   gap_edges()             <- iterate_bwd_edges_for_gap        VA:1328-1349
//                              score_gap_ext/double/open        VA:2116-2219
//   match_edges()           <- iterate_bwd_edges_for_match      VA:1353-1436
//                              score_m/x/y_match                VA:2029-2112
//   end_corner()            <- iterate_bwd_edges_for_end_corner VA:1440-1552
//                              score_gap_close                  VA:2221-2255
//   backtrack()             <- backtrack_new_path               VA:1038-1189
//                              insert_preexisting_gap etc.      viterbi_alignment.h:127-200
//   columns()               <- create_ancestral_sequence's l_pos/r_pos walk
//                              src/main/basic_alignment.cpp:73-171
#include "../include/pagan_dp.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>

namespace {

const double NEG_INF = -HUGE_VAL;

// BA.h:33-50 (only the fields the Viterbi path reads)
struct Cell {
    double score = NEG_INF;
    int x_ind = -1, y_ind = -1, x_edge_ind = -1, y_edge_ind = -1, matrix = -1;
};

// TM:185-233: row x owns columns [begin[x], end[x]]; anything else reads `empty`.
struct BandMatrix {
    int Lx = 0, Ly = 0;
    std::vector<int> begin, end;
    std::vector<int64_t> off;
    std::vector<Cell> cells;
    Cell empty;

    void init(int lx, int ly, const pagan_band *band) {
        Lx = lx; Ly = ly;
        begin.assign(Lx, 0); end.assign(Lx, Ly - 1);
        if (band) {
            for (int x = 0; x < Lx; x++) {           // TM:194
                begin[x] = std::max(0, band->upper[x]);
                end[x] = std::min(band->lower[x], Ly - 1);
            }
        }
        off.assign(Lx + 1, 0);
        for (int x = 0; x < Lx; x++)
            off[x + 1] = off[x] + std::max(0, end[x] - begin[x] + 1);
        cells.assign((size_t)off[Lx], Cell());
    }
    bool inside(int x, int y) const { return y >= begin[x] && y <= end[x]; }
    // TM:85-98: reads outside the tunnel return the shared empty entry (score -inf)
    const Cell &at(int x, int y) const {
        if (!inside(x, y)) return empty;
        return cells[(size_t)(off[x] + (y - begin[x]))];
    }
    Cell &ref(int x, int y) { return cells[(size_t)(off[x] + (y - begin[x]))]; }
};

struct Aligner {
    const pagan_graph *L, *R;
    const pagan_model *mod;
    bool no_terminal_edges, reduced_terminal;
    int Lx, Ly;
    BandMatrix M, X, Y;

    // BA.h:449-462
    static bool bigger(double a, double b) {
        if (a == NEG_INF && b == NEG_INF) return false;
        return a > b;
    }
    // BA.h:490-513 (pair_end_reads is never set: BA.h:565, so that branch is dead)
    float open_pen(int prev_site) const {
        if (reduced_terminal && prev_site == 0) return 0;
        return mod->log_gap_open;
    }
    // evol_model.h:80 : log_gap_close() == 0; BA.h:515-542 returns 0 either way
    static float close_pen() { return 0; }

    // VA:1328-1349 with VA:2116-2219.  `zs/ws/ms` are the slices the reference
    // passes: for X (is_x) the column j of X,Y,M indexed by left site; for Y the
    // row i of Y,X,M indexed by right site.
    void gap_edges(const pagan_graph *g, int site, bool is_x, int fixed, bool end_gap, Cell *max) {
        const BandMatrix &Z = is_x ? X : Y;   // extension
        const BandMatrix &W = is_x ? Y : X;   // double gap
        for (int k = g->bwd_off[site]; k < g->bwd_off[site + 1]; k++) {
            int prev = g->bwd_src[k];
            int eid = g->bwd_eid[k];
            auto rd = [&](const BandMatrix &B) -> double {
                return is_x ? B.at(prev, fixed).score : B.at(fixed, prev).score;
            };
            // score_gap_ext VA:2116-2156 (edge weight deliberately not added, VA:2118,2121)
            {
                double s = rd(Z) + (end_gap ? mod->log_gap_end_ext : mod->log_gap_ext);
                if (bigger(s, max->score)) {
                    max->score = s;
                    if (is_x) { max->matrix = PAGAN_X_MAT; max->x_ind = prev; max->x_edge_ind = eid; }
                    else      { max->matrix = PAGAN_Y_MAT; max->y_ind = prev; max->y_edge_ind = eid; }
                }
            }
            // score_gap_double VA:2158-2188
            {
                double s = rd(W) + close_pen() + mod->log_gap_open;
                if (bigger(s, max->score)) {
                    max->score = s;
                    if (is_x) { max->matrix = PAGAN_Y_MAT; max->x_ind = prev; max->x_edge_ind = eid; }
                    else      { max->matrix = PAGAN_X_MAT; max->y_ind = prev; max->y_edge_ind = eid; }
                }
            }
            // score_gap_open VA:2190-2219
            {
                double s = rd(M) + mod->log_non_gap + open_pen(prev);
                if (bigger(s, max->score)) {
                    max->score = s;
                    max->matrix = PAGAN_M_MAT;
                    if (is_x) { max->x_ind = prev; max->x_edge_ind = eid; }
                    else      { max->y_ind = prev; max->y_edge_ind = eid; }
                }
            }
        }
    }

    // score_m/x/y_match VA:2029-2112
    void score_match(const BandMatrix &B, int label, int k1, int k2, double log_match, Cell *max) {
        double lw = L->bwd_logw[k1];
        int lp = L->bwd_src[k1];
        double rw = R->bwd_logw[k2];
        int rp = R->bwd_src[k2];
        double s = B.at(lp, rp).score + log_match + lw + rw;
        if (bigger(s, max->score)) {
            max->score = s;
            max->x_ind = lp; max->y_ind = rp;
            max->x_edge_ind = L->bwd_eid[k1]; max->y_edge_ind = R->bwd_eid[k2];
            max->matrix = label;
        }
    }

    // VA:1353-1436
    void match_edges(int i, int j, Cell *max) {
        int l0 = L->bwd_off[i], l1 = L->bwd_off[i + 1];
        int r0 = R->bwd_off[j], r1 = R->bwd_off[j + 1];
        if (l0 == l1 || r0 == r1) return;
        float ng = mod->log_non_gap;
        double lms = mod->log_score[(size_t)L->state[i] + (size_t)R->state[j] * mod->n_states]; // VA:1363
        double m_log = 2 * ng + lms;                 // VA:1364 (float product, then double sum)
        double x_log = close_pen() + ng + lms;       // VA:1366 (float sum, then double sum)
        double y_log = close_pen() + ng + lms;       // VA:1367
        // order (l0,r0), (l0,r1..), (l1,r0), (l1,r1..) ... = row-major, VA:1396-1433
        for (int k1 = l0; k1 < l1; k1++)
            for (int k2 = r0; k2 < r1; k2++) {
                score_match(M, PAGAN_M_MAT, k1, k2, m_log, max);
                score_match(X, PAGAN_X_MAT, k1, k2, x_log, max);
                score_match(Y, PAGAN_Y_MAT, k1, k2, y_log, max);
            }
    }

    // VA:856-971
    void fill_cell(int i, int j) {
        if (i == 0 && j == 0) return;
        bool j_end = (j == 0 || j == Ly - 1) && !no_terminal_edges;   // VA:864-868
        bool i_end = (i == 0 || i == Lx - 1) && !no_terminal_edges;   // VA:875-879
        Cell *mx = &X.ref(i, j), *my = &Y.ref(i, j), *mm = &M.ref(i, j);
        if (i > 0) { gap_edges(L, i, true, j, j_end, mx); mx->y_ind = j; }   // VA:898-915
        if (j > 0) { gap_edges(R, j, false, i, i_end, my); my->x_ind = i; }  // VA:927-944
        if (i > 0 && j > 0) match_edges(i, j, mm);                           // VA:956-963
    }

    // score_gap_close VA:2221-2255
    void gap_close(const pagan_graph *g, int k, bool is_x, Cell *max) {
        int prev = g->bwd_src[k];
        double s = (is_x ? X.at(prev, Ly - 1).score : Y.at(Lx - 1, prev).score) + close_pen();
        if (bigger(s, max->score)) {
            max->score = s;
            if (is_x) { max->matrix = PAGAN_X_MAT; max->x_ind = prev; max->x_edge_ind = g->bwd_eid[k]; max->y_edge_ind = -1; }
            else      { max->matrix = PAGAN_Y_MAT; max->y_ind = prev; max->y_edge_ind = g->bwd_eid[k]; max->x_edge_ind = -1; }
        }
    }

    // VA:1440-1552
    void end_corner(Cell *max) {
        int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1];
        int r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        if (l0 == l1 || r0 == r1) return;
        double m_log = mod->log_non_gap;             // VA:1451
        auto m_cand = [&](int k1, int k2) { score_match(M, PAGAN_M_MAT, k1, k2, m_log, max); };
        double best;
        m_cand(l0, r0);
        best = max->score;
        gap_close(L, l0, true, max);
        if (bigger(max->score, best)) { best = max->score; max->y_ind = Ly - 1; }
        gap_close(R, r0, false, max);
        if (bigger(max->score, best)) { best = max->score; max->x_ind = Lx - 1; }
        for (int k2 = r0 + 1; k2 < r1; k2++) {        // VA:1479-1500
            m_cand(l0, k2);
            if (bigger(max->score, best)) best = max->score;
            gap_close(R, k2, false, max);
            if (bigger(max->score, best)) { best = max->score; max->x_ind = Lx - 1; }
        }
        for (int k1 = l0 + 1; k1 < l1; k1++) {        // VA:1504-1550
            m_cand(k1, r0);
            if (bigger(max->score, best)) best = max->score;
            gap_close(L, k1, true, max);
            if (bigger(max->score, best)) { best = max->score; max->y_ind = Ly - 1; }
            for (int k2 = r0 + 1; k2 < r1; k2++) {
                m_cand(k1, k2);
                if (bigger(max->score, best)) best = max->score;
                gap_close(R, k2, false, max);
                if (bigger(max->score, best)) { best = max->score; max->x_ind = Lx - 1; }
            }
        }
    }

    // One entry of the reference's `path` vector, reduced to what
    // create_ancestral_sequence reads: the matrix label and real/skip flag.
    struct Step { int matrix; bool real; };

    // first bwd edge of `site` whose start is `start` (Sequence::get_fwd_edge_index_at_site
    // looked up from the other end, sequence.h:782-798; start/end pairs are unique per site
    // by construction, basic_alignment.cpp:579)
    static int find_edge(const pagan_graph *g, int start, int site) {
        for (int k = g->bwd_off[site]; k < g->bwd_off[site + 1]; k++)
            if (g->bwd_src[k] == start) return g->bwd_eid[k];
        return -1;
    }

    // VA:1038-1189.  Returns 0, or -1 on an "incorrect backward pointer" (VA:1167-1171).
    int backtrack(const Cell &fp, std::vector<Step> *path, std::vector<char> *lused, std::vector<char> *rused) {
        std::vector<Step> stack;
        int vit = fp.matrix, x_ind = fp.x_ind, y_ind = fp.y_ind;
        bool first_x = true, first_y = true;
        if (fp.x_edge_ind >= 0) (*lused)[fp.x_edge_ind] = 1;     // VA:1054-1057
        if (fp.y_edge_ind >= 0) (*rused)[fp.y_edge_ind] = 1;
        int j = Ly - 1, i = Lx - 1;
        int max_j = j + 1, max_i = i + 1;
        // insert_preexisting_gap viterbi_alignment.h:146-193 (mark_used is false there)
        auto skips = [&](int xi, int yi) {
            while (xi < i) { stack.push_back({PAGAN_X_MAT, false}); --i; }
            while (yi < j) { stack.push_back({PAGAN_Y_MAT, false}); --j; }
        };
        // insert_new_path_pointer viterbi_alignment.h:196-200
        auto push = [&](int matrix) { if (i > 0 || j > 0) stack.push_back({matrix, true}); };
        skips(x_ind, y_ind);
        push(fp.matrix);
        while (j >= 0) {
            while (i >= 0) {
                if (vit == PAGAN_M_MAT) {
                    if (first_x) { int e = find_edge(L, x_ind, max_i); if (e >= 0) (*lused)[e] = 1; first_x = false; }
                    if (first_y) { int e = find_edge(R, y_ind, max_j); if (e >= 0) (*rused)[e] = 1; first_y = false; }
                    const Cell &c = M.at(i, j);
                    vit = c.matrix; x_ind = c.x_ind; y_ind = c.y_ind;
                    if (c.x_edge_ind < 0 || c.y_edge_ind < 0) return -1;  // vector::at would throw
                    (*lused)[c.x_edge_ind] = 1; (*rused)[c.y_edge_ind] = 1;
                    int label = c.matrix;
                    i--; j--;
                    skips(x_ind, y_ind);
                    push(label);
                } else if (vit == PAGAN_X_MAT) {
                    if (first_x) { int e = find_edge(L, x_ind, max_i); if (e >= 0) (*lused)[e] = 1; first_x = false; }
                    const Cell &c = X.at(i, j);
                    vit = c.matrix; x_ind = c.x_ind; y_ind = c.y_ind;
                    if (c.x_edge_ind < 0) return -1;
                    (*lused)[c.x_edge_ind] = 1;
                    int label = c.matrix;
                    i--;
                    skips(x_ind, y_ind);
                    push(label);
                } else if (vit == PAGAN_Y_MAT) {
                    if (first_y) { int e = find_edge(R, y_ind, max_j); if (e >= 0) (*rused)[e] = 1; first_y = false; }
                    const Cell &c = Y.at(i, j);
                    vit = c.matrix; x_ind = c.x_ind; y_ind = c.y_ind;
                    if (c.y_edge_ind < 0) return -1;
                    (*rused)[c.y_edge_ind] = 1;
                    int label = c.matrix;
                    j--;
                    skips(x_ind, y_ind);
                    push(label);
                } else {
                    return -1;                                        // VA:1167-1171
                }
                if (i < 1 && j < 1) break;
            }
            if (i < 1 && j < 1) break;
        }
        path->assign(stack.rbegin(), stack.rend());                   // VA:1183-1187
        return 0;
    }
};

int check_graph(const pagan_graph *g) {
    if (!g || g->n_sites < 2 || !g->state || !g->bwd_off || g->bwd_off[0] != 0) return PAGAN_E_GRAPH;
    for (int s = 0; s < g->n_sites; s++) {
        if (g->bwd_off[s + 1] < g->bwd_off[s]) return PAGAN_E_GRAPH;
        for (int k = g->bwd_off[s]; k < g->bwd_off[s + 1]; k++) {
            if (g->bwd_src[k] < 0 || g->bwd_src[k] >= s) return PAGAN_E_GRAPH;
            if (g->bwd_eid[k] < 0 || g->bwd_eid[k] >= g->n_edges) return PAGAN_E_GRAPH;
        }
    }
    return PAGAN_OK;
}

} // namespace

extern "C" {

// Same contract as pagan_dp_align (include/pagan_dp.h), computed on the CPU.
// out->fill_ms / trace_ms hold CPU wall milliseconds.
int oracle_dp_align(const pagan_graph *left, const pagan_graph *right, const pagan_model *model,
                    const pagan_band *band, const pagan_opts *opts, pagan_result *out) {
    if (!left || !right || !model || !out) return PAGAN_E_ARG;
    int rc;
    if ((rc = check_graph(left)) != PAGAN_OK) return rc;
    if ((rc = check_graph(right)) != PAGAN_OK) return rc;
    std::memset(out, 0, sizeof(*out));
    Aligner a;
    a.L = left; a.R = right; a.mod = model;
    uint32_t flags = opts ? opts->flags : 0;
    a.no_terminal_edges = flags & PAGAN_OPT_NO_TERMINAL_EDGES;
    a.reduced_terminal = !(flags & PAGAN_OPT_NO_REDUCED_TERMINAL_PEN);   // BA.h:627-628
    a.Lx = left->n_sites - 1; a.Ly = right->n_sites - 1;
    if (band && (band->n < a.Lx || !band->upper || !band->lower)) return PAGAN_E_BAND;
    for (int s = 1; s < a.Lx; s++) if (left->state[s] < 0 || left->state[s] >= model->n_states) return PAGAN_E_MODEL;
    for (int s = 1; s < a.Ly; s++) if (right->state[s] < 0 || right->state[s] >= model->n_states) return PAGAN_E_MODEL;

    auto t0 = std::chrono::steady_clock::now();
    a.M.init(a.Lx, a.Ly, band); a.X.init(a.Lx, a.Ly, band); a.Y.init(a.Lx, a.Ly, band);
    // initialise_array_corner VA:725-736 (X,Y corner stay -inf by construction).  A band
    // that excludes (0,0) would make the reference write 0 into the shared out-of-tunnel
    // entry (TM:85-98); define_tunnel never produces one (find_anchors.cpp:392-406), so
    // it is rejected here and in the shipped library alike.
    if (!a.M.inside(0, 0)) return PAGAN_E_BAND;
    a.M.ref(0, 0).score = 0.0;
    int64_t cells = 0;
    if (band) {                                      // VA:260-272
        for (int i = 0; i < a.Lx; i++)
            for (int j = a.M.begin[i]; j <= a.M.end[i]; j++) { a.fill_cell(i, j); cells++; }
    } else {                                         // VA:273-282 loops j outer, i inner; every
        for (int i = 0; i < a.Lx; i++)               // predecessor has p<i or q<j, so row-major gives
            for (int j = 0; j < a.Ly; j++) { a.fill_cell(i, j); cells++; }   // the same cells (and is cache-friendly)
    }
    Cell max_end;
    a.end_corner(&max_end);                          // VA:289-296
    auto t1 = std::chrono::steady_clock::now();

    out->cells = cells;
    out->score = max_end.score;
    out->end_matrix = max_end.matrix; out->end_x = max_end.x_ind; out->end_y = max_end.y_ind;
    out->end_x_edge = max_end.x_edge_ind; out->end_y_edge = max_end.y_edge_ind;
    out->fill_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (max_end.score == NEG_INF) { out->status = PAGAN_DP_UNREACHABLE; return PAGAN_OK; }

    std::vector<Aligner::Step> path;
    std::vector<char> lused(left->n_edges, 0), rused(right->n_edges, 0);
    if (a.backtrack(max_end, &path, &lused, &rused) != 0) return PAGAN_E_INTERNAL;
    auto t2 = std::chrono::steady_clock::now();
    out->trace_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();

    // basic_alignment.cpp:73-171: columns consume child sites sequentially
    out->n_cols = (int32_t)path.size();
    out->cols = (pagan_col *)std::malloc(sizeof(pagan_col) * (path.size() + 1));
    int l_pos = 1, r_pos = 1;
    for (size_t k = 0; k < path.size(); k++) {
        pagan_col c;
        if (path[k].matrix == PAGAN_X_MAT) { c.left = l_pos++; c.right = -1; c.path_state = path[k].real ? PAGAN_XGAPPED : PAGAN_XSKIPPED; }
        else if (path[k].matrix == PAGAN_Y_MAT) { c.left = -1; c.right = r_pos++; c.path_state = path[k].real ? PAGAN_YGAPPED : PAGAN_YSKIPPED; }
        else { c.left = l_pos++; c.right = r_pos++; c.path_state = PAGAN_MATCHED; }
        out->cols[k] = c;
    }
    auto collect = [](const std::vector<char> &u, int32_t *n, int32_t **arr) {
        int cnt = 0; for (char c : u) cnt += c;
        *arr = (int32_t *)std::malloc(sizeof(int32_t) * (cnt + 1));
        int k = 0; for (size_t e = 0; e < u.size(); e++) if (u[e]) (*arr)[k++] = (int32_t)e;
        *n = cnt;
    };
    collect(lused, &out->n_left_used, &out->left_used);
    collect(rused, &out->n_right_used, &out->right_used);
    return PAGAN_OK;
}

void oracle_result_free(pagan_result *r) {
    if (!r) return;
    std::free(r->cols); std::free(r->left_used); std::free(r->right_used);
    r->cols = nullptr; r->left_used = nullptr; r->right_used = nullptr;
}

} // extern "C"
