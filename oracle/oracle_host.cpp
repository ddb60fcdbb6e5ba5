// oracle_host.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_dp.cpp header; PARITY UNPINNED).
//
// CPU restatement of the host-side steps either side of the DP, kept as close to the
// reference's data structures as possible (per-site linked edge lists with the shared
// iteration/tail cursor, src/main/sequence.h:343-417) so that list ORDER -- which
// decides DP ties -- is reproduced, not re-derived:
//
//   Graph::leaf()            <- Sequence::create_default_sequence   src/main/sequence.cpp:152-303
//   Graph::parent()          <- Basic_alignment::build_ancestral_sequence
//                               create_ancestral_sequence           src/main/basic_alignment.cpp:61-179
//                               create_ancestral_edges              basic_alignment.cpp:181-368
//                               transfer_child_edge (both)          basic_alignment.cpp:510-653
//                               check_skipped_boundaries            basic_alignment.cpp:370-489
//                               delete_edge_range                   basic_alignment.cpp:491-508
//   Graph::sequence_string() <- Sequence::get_sequence_string       sequence.cpp:704-740
//   prefix_hits()            <- Find_anchors::find_long_substrings  src/utils/find_anchors.cpp:35-127
//   order_conflicts()        <- Find_anchors::check_hits_order_conflict find_anchors.cpp:225-317
//   tunnel()                 <- Find_anchors::define_tunnel         find_anchors.cpp:320-447
//   dna_parsimony()          <- Model_factory::define_dna_alphabet  src/utils/model_factory.cpp:120-227
#include "../include/pagan_dp.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum SiteType { start_site, real_site, stop_site, break_start_site, break_stop_site, non_real }; // sequence.h:226
enum PathState { ends_site, terminal, matched, xgapped, ygapped, xskipped, yskipped };            // sequence.h:229

struct Edge {                       // sequence.h:34-59
    int index = -1, start, end;
    float w = 1.0f, logw = 0.0f;
    int next_fwd = -1, next_bwd = -1;
    bool used = false;
    int count_since_used = 0;       // branch_count_since_last_used
    float dist_since_used = 0;      // branch_distance_since_last_used
    int count_as_skipped = 0;       // branch_count_as_skipped_edge
    Edge(int s, int e) : start(s), end(e) {}
    Edge(int s, int e, float wt) : start(s), end(e), w(wt), logw(std::log(wt)) {}   // sequence.h:61-62 (logf)
    void set_weight(float x) { w = x; logw = std::log(x); }                          // sequence.h:98
    void multiply_weight(float x) { w *= x; logw = std::log(w); }                    // sequence.h:99
    bool same(const Edge &b) const { return start == b.start && end == b.end; }      // sequence.h:101-104
};

struct Site {                       // sequence.h:216-251
    int state = -1, type = real_site, path_state = terminal;
    int left = -1, right = -1;
    int first_fwd = -1, cur_fwd = -1, first_bwd = -1, cur_bwd = -1;
    int count_since_used = 0;
    float dist_since_used = 0;
    bool ambiguous = false;
    char symbol = '0';
};

struct Graph {
    std::vector<Site> sites;
    std::vector<Edge> edges;
    bool terminal_sequence = false;

    // ---- linked-list primitives, sequence.h:343-417 ---------------------------------
    void add_fwd(int s, int e) {
        Site &t = sites[s];
        if (t.first_fwd < 0) { t.first_fwd = t.cur_fwd = e; return; }
        int prev = t.cur_fwd; t.cur_fwd = e; edges[prev].next_fwd = e;
    }
    void add_bwd(int s, int e) {
        Site &t = sites[s];
        if (t.first_bwd < 0) { t.first_bwd = t.cur_bwd = e; return; }
        int prev = t.cur_bwd; t.cur_bwd = e; edges[prev].next_bwd = e;
    }
    bool has_bwd(int s) const { return sites[s].first_bwd >= 0; }
    bool has_fwd(int s) const { return sites[s].first_fwd >= 0; }
    int first_bwd(int s) { sites[s].cur_bwd = sites[s].first_bwd; return sites[s].cur_bwd; }
    bool has_next_bwd(int s) const { return edges[sites[s].cur_bwd].next_bwd >= 0; }
    int next_bwd(int s) { sites[s].cur_bwd = edges[sites[s].cur_bwd].next_bwd; return sites[s].cur_bwd; }
    int first_fwd(int s) { sites[s].cur_fwd = sites[s].first_fwd; return sites[s].cur_fwd; }
    bool has_next_fwd(int s) const { return edges[sites[s].cur_fwd].next_fwd >= 0; }
    int next_fwd(int s) { sites[s].cur_fwd = edges[sites[s].cur_fwd].next_fwd; return sites[s].cur_fwd; }

    int push_edge(Edge e) { e.index = (int)edges.size(); edges.push_back(e); return e.index; }   // sequence.h:727-733

    // Site::contains_bwd_edge (non-thorough), sequence.h:419-450
    bool contains_bwd(int s, const Edge &c) {
        if (!has_bwd(s)) return false;
        int e = first_bwd(s);
        if (edges[e].same(c)) return true;
        while (has_next_bwd(s)) { e = next_bwd(s); if (edges[e].same(c)) return true; }
        return false;
    }
    // Site::update_bwd_edge_details (non-thorough), sequence.h:452-502
    void update_bwd(int s, const Edge &c) {
        if (!has_bwd(s)) return;
        auto upd = [&](Edge &e) {
            if (!e.same(c)) return;
            e.count_as_skipped = c.count_as_skipped; e.count_since_used = c.count_since_used;
            e.dist_since_used = c.dist_since_used; e.set_weight((float)(double)c.w);
        };
        int e = first_bwd(s); upd(edges[e]);
        while (has_next_bwd(s)) { e = next_bwd(s); upd(edges[e]); }
    }
    // Sequence::get_bwd_edge_index_at_site, sequence.h:756-772
    int bwd_index_at(int s, const Edge &c) {
        if (!has_bwd(s)) return -1;
        int e = first_bwd(s);
        if (edges[e].same(c)) return e;
        while (has_next_bwd(s)) { e = next_bwd(s); if (edges[e].same(c)) return e; }
        return -1;
    }
    // Site::delete_bwd_edge, sequence.h:537-581
    void delete_bwd_edge(int s, int edge_ind) {
        if (!has_bwd(s)) return;
        Site &t = sites[s];
        int e = first_bwd(s);
        if (e == edge_ind) {
            if (has_next_bwd(s)) { int e2 = next_bwd(s); t.cur_bwd = t.first_bwd = e2; }
            else t.cur_bwd = t.first_bwd = -1;
            return;
        }
        while (has_next_bwd(s)) {
            int prev = e;
            e = next_bwd(s);
            if (e == edge_ind) {
                if (has_next_bwd(s)) { e = next_bwd(s); edges[prev].next_bwd = e; }
                else edges[prev].next_bwd = -1;
            }
        }
    }
    // Site::delete_fwd_edge, sequence.h:583-625
    void delete_fwd_edge(int s, int edge_ind) {
        if (!has_fwd(s)) return;
        Site &t = sites[s];
        int e = first_fwd(s);
        if (e == edge_ind) {
            if (has_next_fwd(s)) { e = next_fwd(s); t.cur_fwd = t.first_fwd = e; }
            else t.cur_fwd = t.first_fwd = -1;
            return;
        }
        while (has_next_fwd(s)) {
            int prev = e;
            e = next_fwd(s);
            if (e == edge_ind) {
                if (has_next_fwd(s)) { e = next_fwd(s); edges[prev].next_fwd = e; }
                else edges[prev].next_fwd = -1;
            }
        }
    }
    // Sequence::delete_all_bwd_edges_at_site / fwd, sequence.h:836-870
    void delete_all_bwd(int s) {
        if (has_bwd(s)) {
            int e = first_bwd(s);
            delete_fwd_edge(edges[e].start, e);
            while (has_next_bwd(s)) { e = next_bwd(s); delete_fwd_edge(edges[e].start, e); }
        }
        sites[s].first_bwd = sites[s].cur_bwd = -1;
    }
    void delete_all_fwd(int s) {
        if (has_fwd(s)) {
            int e = first_fwd(s);
            delete_bwd_edge(edges[e].end, e);
            while (has_next_fwd(s)) { e = next_fwd(s); delete_bwd_edge(edges[e].end, e); }
        }
        sites[s].first_fwd = sites[s].cur_fwd = -1;
    }

    // ---- leaf: sequence.cpp:152-303 --------------------------------------------------
    // flags: 1 = --454, 2 = --homopolymer
    static Graph *leaf(const char *seq, const char *alphabet, int flags) {
        Graph *g = new Graph();
        g->terminal_sequence = true;
        std::string alpha(alphabet);
        Site first; first.type = start_site; first.path_state = ends_site; first.state = -1;
        g->sites.push_back(first);
        int in_row = 1, prev_row = 1, prev_state = -1;
        g->push_edge(Edge(-1, 0));                                   // sequence.cpp:164-165 (never linked)
        for (const char *p = seq; *p; p++) {
            if (*p == '0') continue;
            Site s; s.state = (int)alpha.find(*p); s.symbol = *p;
            g->sites.push_back(s);
            int cur = (int)g->sites.size() - 1, prev = cur - 1;
            if (s.state == prev_state) { in_row++; prev_row = 1; }
            else { prev_row = in_row; in_row = 1; prev_state = s.state; }
            if ((flags & 1) && prev_row > 2) {                       // sequence.cpp:205-249
                int e = g->push_edge(Edge(prev, cur, 1.0f));
                g->sites[prev].first_fwd = g->sites[prev].cur_fwd = e;
                g->sites[cur].first_bwd = g->sites[cur].cur_bwd = e;
                int prev_ind = g->edges[g->first_bwd(prev)].start;
                int e2 = g->push_edge(Edge(prev_ind, cur, 0.9f));
                g->add_fwd(prev_ind, e2); g->add_bwd(cur, e2);
                if (prev_row >= 5) {
                    int pp = g->edges[g->first_bwd(prev_ind)].start;
                    int e3 = g->push_edge(Edge(pp, cur, 0.9f));
                    g->add_fwd(pp, e3); g->add_bwd(cur, e3);
                }
            } else if ((flags & 2) && prev_row >= 2) {               // sequence.cpp:253-278
                int e = g->push_edge(Edge(prev, cur, 1.0f));
                g->sites[prev].first_fwd = g->sites[prev].cur_fwd = e;
                g->sites[cur].first_bwd = g->sites[cur].cur_bwd = e;
                int prev_ind = g->edges[g->first_bwd(prev)].start;
                while (prev_row >= 2) {
                    int e2 = g->push_edge(Edge(prev_ind, cur, 0.25f));
                    g->add_fwd(prev_ind, e2); g->add_bwd(cur, e2);
                    prev_ind = g->edges[g->first_bwd(prev_ind)].start;
                    prev_row--;
                }
            } else {                                                 // sequence.cpp:280-287
                int e = g->push_edge(Edge(prev, cur));
                g->sites[prev].first_fwd = g->sites[prev].cur_fwd = e;
                g->sites[cur].first_bwd = g->sites[cur].cur_bwd = e;
            }
        }
        Site last; last.type = stop_site; last.path_state = ends_site; last.state = -1;
        g->sites.push_back(last);
        int cur = (int)g->sites.size() - 1, prev = cur - 1;
        int e = g->push_edge(Edge(prev, cur));
        g->sites[prev].first_fwd = g->sites[prev].cur_fwd = e;
        g->sites[cur].first_bwd = g->sites[cur].cur_bwd = e;
        return g;
    }

    // sequence.cpp:704-740 (ancestral alphabet == full alphabet one-letter symbols,
    // model_factory.cpp:1469-1472)
    std::string sequence_string(bool with_gaps, const char *alphabet) const {
        std::string out;
        int n = (int)sites.size();
        for (int j = 1; j < n - 1; j++) {
            const Site &s = sites[j];
            if (terminal_sequence) { out += s.symbol; continue; }
            if (s.path_state != xskipped && s.path_state != yskipped && s.type != non_real) out += alphabet[s.state];
            else if (with_gaps) out += '-';
        }
        return out;
    }
};

struct BuildOpts {
    float max_skip_distance = 0.5f;      // basic_alignment.h:555-557
    int max_skip_branches = 10;
    int max_match_skip_branches = 5;
    float branch_skip_probability = 0.9f; // basic_alignment.h:560
    bool reduced_terminal = true;         // basic_alignment.h:627-628
};

struct Builder {
    Graph *left, *right, *seq;
    float lbl, rbl;
    BuildOpts o;

    // basic_alignment.cpp:572-653
    void transfer2(Edge edge, const Edge &child, float branch_length) {
        if (seq->contains_bwd(edge.end, edge)) { seq->update_bwd(edge.end, edge); return; }
        if (!child.used && child.count_since_used + 1 > o.max_skip_branches) return;
        if (!child.used && child.dist_since_used + branch_length > o.max_skip_distance) return;
        float dist_start = seq->sites[edge.start].dist_since_used, dist_end = seq->sites[edge.end].dist_since_used;
        int count_start = seq->sites[edge.start].count_since_used, count_end = seq->sites[edge.end].count_since_used;
        const float branch_weight = 1.0f;
        if (dist_start != dist_end || count_start != count_end) {
            edge.dist_since_used = std::max(dist_start, dist_end);
            edge.count_since_used = std::max(count_start, count_end);
            edge.multiply_weight((float)(branch_weight * (double)child.w * o.branch_skip_probability));
        } else if (!child.used && count_start == 0 && count_end == 0) {
            edge.dist_since_used = child.dist_since_used + branch_length;
            edge.count_since_used = child.count_since_used + 1;
            edge.multiply_weight((float)(branch_weight * (double)child.w * o.branch_skip_probability));
        } else if (!child.used) {
            edge.dist_since_used = child.dist_since_used + branch_length;
            edge.count_since_used = child.count_since_used + 1;
        }
        if (seq->bwd_index_at(edge.end, edge) < 0) {
            edge.count_as_skipped = child.used ? 0 : child.count_as_skipped;
            int e = seq->push_edge(edge);
            seq->add_fwd(edge.start, e);
            seq->add_bwd(edge.end, e);
        }
    }
    // basic_alignment.cpp:510-569 (weight_edges and pair_end_reads are off: BA.h:562,565)
    void transfer(const Edge &child, const std::vector<int> &ci, float branch_length) {
        Edge edge(ci.at(child.start), ci.at(child.end), 1.0f);
        if (o.reduced_terminal) {
            if (seq->sites[edge.start].type == start_site && edge.end - edge.start > 1)
                if (child.end - child.start == 1) edge.start = edge.end - 1;
            if (seq->sites[edge.end].type == stop_site && edge.end - edge.start > 1)
                if (child.end - child.start == 1) edge.end = edge.start + 1;
        }
        transfer2(edge, child, branch_length);
    }

    // basic_alignment.cpp:61-179
    void create_sites(const pagan_col *cols, int n_cols, const int *parsimony, int S, int char_as) {
        Site first; first.type = start_site; first.path_state = ends_site; first.state = -1; first.left = 0; first.right = 0;
        seq->sites.push_back(first);
        int l_pos = 1, r_pos = 1;
        for (int k = 0; k < n_cols; k++) {
            Site s;
            int ps = cols[k].path_state;
            if (ps == xgapped || ps == xskipped) {
                const Site &c = left->sites[l_pos];
                s.state = c.state; s.ambiguous = c.ambiguous; s.path_state = ps;
                if (ps == xskipped) { s.count_since_used = c.count_since_used + 1; s.dist_since_used = c.dist_since_used + lbl; }
                s.left = l_pos; s.right = -1; l_pos++;
            } else if (ps == ygapped || ps == yskipped) {
                const Site &c = right->sites[r_pos];
                s.state = c.state; s.ambiguous = c.ambiguous; s.path_state = ps;
                if (ps == yskipped) { s.count_since_used = c.count_since_used + 1; s.dist_since_used = c.dist_since_used + rbl; }
                s.left = -1; s.right = r_pos; r_pos++;
            } else {
                int lc = left->sites[l_pos].state, rc = right->sites[r_pos].state;
                s.state = parsimony[lc + rc * S];                     // Int_matrix::g(i,j)=data[i+j*X]
                if (lc != rc || lc >= char_as) s.ambiguous = true;
                s.path_state = matched; s.left = l_pos; s.right = r_pos; l_pos++; r_pos++;
            }
            seq->sites.push_back(s);
        }
        Site last; last.type = stop_site; last.path_state = ends_site; last.state = -1;
        last.left = (int)left->sites.size() - 1; last.right = (int)right->sites.size() - 1;
        seq->sites.push_back(last);
    }

    // basic_alignment.cpp:181-368 (edges_for_skipped_flanked_by_gaps is false: BA.h:551)
    void create_edges() {
        std::vector<int> lci, rci;
        int n = (int)seq->sites.size();
        for (int i = 0; i < n; i++) {
            if (seq->sites[i].left >= 0) lci.push_back(i);
            if (seq->sites[i].right >= 0) rci.push_back(i);
        }
        int prev_state = -1;
        for (int i = 1; i < n; i++) {
            int pstate = seq->sites[i].path_state;
            int li = seq->sites[i].left, ri = seq->sites[i].right;
            if (li >= 0) {
                if (left->has_bwd(li)) {
                    int e = left->first_bwd(li);
                    transfer(left->edges[e], lci, lbl);
                    while (left->has_next_bwd(li)) { e = left->next_bwd(li); transfer(left->edges[e], lci, lbl); }
                }
                if ((pstate == xgapped || pstate == xskipped) && (prev_state == ygapped || prev_state == yskipped)) {
                    int e = seq->push_edge(Edge(i - 1, i, 1.0f));       // basic_alignment.cpp:288-296
                    seq->add_fwd(i - 1, e); seq->add_bwd(i, e);
                }
            }
            if (ri >= 0) {
                if (right->has_bwd(ri)) {
                    int e = right->first_bwd(ri);
                    transfer(right->edges[e], rci, rbl);
                    while (right->has_next_bwd(ri)) { e = right->next_bwd(ri); transfer(right->edges[e], rci, rbl); }
                }
                if ((pstate == ygapped || pstate == yskipped) && (prev_state == xgapped || prev_state == xskipped)) {
                    int e = seq->push_edge(Edge(i - 1, i, 1.0f));       // basic_alignment.cpp:351-358
                    seq->add_fwd(i - 1, e); seq->add_bwd(i, e);
                }
            }
            prev_state = pstate;
        }
    }

    // basic_alignment.cpp:370-489
    void check_skipped_boundaries() {
        int n = (int)seq->sites.size();
        auto max_start_bwd = [&](int s) {
            int e = seq->first_bwd(s);
            while (seq->has_next_bwd(s)) { int a = seq->next_bwd(s); if (seq->edges[a].start > seq->edges[e].start) e = a; }
            return e;
        };
        for (int i = 0; i < n; i++) {
            int ts = seq->sites[i].path_state;
            if (seq->has_bwd(i)) {
                int e = max_start_bwd(i);
                int ps = seq->sites[seq->edges[e].start].path_state;
                if ((ps == matched || ps == 0 /* Site::start_site compared with a path_state */) && (ts == xskipped || ts == yskipped))
                    seq->edges[e].count_as_skipped++;
            }
            if (seq->has_fwd(i)) {
                int e = seq->first_fwd(i);
                while (seq->has_next_fwd(i)) { int a = seq->next_fwd(i); if (seq->edges[a].start < seq->edges[e].start) e = a; }
                int ns = seq->sites[seq->edges[e].end].path_state;
                if ((ts == xskipped || ts == yskipped) && (ns == matched || ns == ends_site))
                    seq->edges[e].count_as_skipped++;
            }
        }
        bool non_skipped = true;
        int skip_start = -1;
        for (int i = 1; i < n; i++) {
            int ts = seq->sites[i].path_state;
            if (non_skipped && (ts == xskipped || ts == yskipped)) {
                if (seq->has_bwd(i)) {
                    int e = max_start_bwd(i);
                    if (seq->edges[e].count_as_skipped > o.max_match_skip_branches) skip_start = i;
                }
                non_skipped = false;
            }
            if (!non_skipped && skip_start >= 0 && ts == matched) {
                int edge_ind = -1;
                if (seq->has_bwd(i)) {
                    int e = seq->first_bwd(i);
                    if (seq->edges[e].count_as_skipped > o.max_match_skip_branches) edge_ind = e;
                    while (seq->has_next_bwd(i)) {
                        e = seq->next_bwd(i);
                        if (seq->edges[e].count_as_skipped > o.max_match_skip_branches) edge_ind = e;
                    }
                }
                if (edge_ind >= 0) {                                   // delete_edge_range, basic_alignment.cpp:491-508
                    int s = seq->edges[edge_ind].start;
                    while (s >= skip_start) {
                        seq->sites[s].type = non_real;
                        seq->delete_all_bwd(s);
                        seq->delete_all_fwd(s);
                        --s;
                    }
                }
                non_skipped = true; skip_start = -1;
            }
            if (ts == xgapped || ts == ygapped || ts == matched) { non_skipped = true; skip_start = -1; }
        }
    }
};

// ---- anchors ----------------------------------------------------------------------------
struct Hit { int s1, s2, len, score; };

// find_anchors.cpp:35-127.  The reference sorts pointers into two NUL-terminated copies with
// qsort+strcmp (glibc: merge sort, stable) and scans adjacent pairs.  Its `c1[n]=0` writes one
// past a VLA (find_anchors.cpp:52,61); this restates the intended behaviour: each string ends
// at its own terminator.
void prefix_hits(const std::string &a, const std::string &b, int min_length, std::vector<Hit> *hits) {
    int len1 = (int)a.size(), len2 = (int)b.size();
    std::vector<const char *> ptr;
    for (int i = 0; i < len1; i++) ptr.push_back(a.c_str() + i);
    for (int i = 0; i < len2; i++) ptr.push_back(b.c_str() + i);
    std::stable_sort(ptr.begin(), ptr.end(), [](const char *p, const char *q) { return std::strcmp(p, q) < 0; });
    auto in1 = [&](const char *p) { return p >= a.c_str() && p < a.c_str() + len1; };
    auto in2 = [&](const char *p) { return p >= b.c_str() && p < b.c_str() + len2; };
    for (size_t i = 0; i + 1 < ptr.size(); i++) {
        const char *p = ptr[i], *q = ptr[i + 1];
        if (!((in1(p) && in2(q)) || (in1(q) && in2(p)))) continue;
        int len = 0;
        { const char *x = p, *y = q; while (*x && (*x++ == *y++)) len++; }   // find_anchors.h:99-105
        if (len >= min_length) {
            Hit h; h.s1 = (int)((in1(p) ? p : q) - a.c_str()); h.s2 = (int)((in2(p) ? p : q) - b.c_str());
            h.len = len; h.score = len;
            hits->push_back(h);
        }
    }
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.len > q.len; });   // find_anchors.cpp:87
    std::vector<char> h1(len1, 0), h2(len2, 0);
    for (size_t k = 0; k < hits->size();) {
        Hit &h = (*hits)[k];
        bool overlap = false;
        for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; i++, j++)
            if (h1.at(i) || h2.at(j)) { overlap = true; break; }
        if (overlap) hits->erase(hits->begin() + k);
        else { for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; i++, j++) { h1[i] = 1; h2[j] = 1; } k++; }
    }
}

// find_anchors.cpp:225-317; `trim` = --exonerate-hit-trim (settings.cpp:155, default 5).
// The two `start+trim;` statements at find_anchors.cpp:248-251 have no effect: only the
// length shrinks.
void order_conflicts(int len1, int len2, int trim, std::vector<Hit> *hits) {
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.score > q.score; });
    std::vector<char> h1(len1, 0), h2(len2, 0);
    for (size_t k = 0; k < hits->size();) {
        Hit &h = (*hits)[k];
        h.len -= trim * 2;
        bool overlap = false;
        for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; i++, j++)
            if (h1.at(i) || h2.at(j)) { overlap = true; break; }
        if (overlap) hits->erase(hits->begin() + k);
        else { for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; i++, j++) { h1[i] = 1; h2[j] = 1; } k++; }
    }
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { if (p.s1 == q.s1) return p.s2 < q.s2; return p.s1 < q.s1; });
    size_t i1 = 0, i2 = 1;
    while (i1 < hits->size() && i2 < hits->size()) {
        if ((*hits)[i1].s2 > (*hits)[i2].s2) {
            if ((*hits)[i1].score < (*hits)[i2].score) hits->erase(hits->begin() + i1);
            else hits->erase(hits->begin() + i2);
            i2 = i1 + 1;
            continue;
        }
        i1++; i2++;
    }
}

// find_anchors.cpp:320-447; str1/str2 are the gapped strings (skipped sites as '-').
void tunnel(const std::vector<Hit> &hits, const std::string &str1, const std::string &str2, int width,
            std::vector<int> *upper, std::vector<int> *lower) {
    int length1 = (int)str1.size(), length2 = (int)str2.size();
    std::vector<int> index1, index2;
    for (int i = 0; i < length1; i++) if (str1[i] != '-') index1.push_back(i + 1);
    for (int i = 0; i < length2; i++) if (str2[i] != '-') index2.push_back(i + 1);
    std::vector<int> diag(length1 + 1, -1);
    for (const Hit &h : hits) {
        int i = 0;
        for (; i < h.len; i++) diag.at(index1.at(h.s1 + i)) = index2.at(h.s2 + i);
        if (h.s1 + i < (int)index1.size() && index1.at(h.s1 + i) < (int)diag.size()) diag.at(index1.at(h.s1 + i)) = -2;
    }
    int y1 = 0, y2 = 0, y, prev_y = 0, m_count = 0;
    for (int i = 0; i <= length1; i++) {
        if (i >= width && diag.at(i - width) >= 0) y1 = diag.at(i - width);
        if (diag.at(i) >= 0) y2 = diag.at(i) - width;
        bool run = diag.at(i) >= 0 && i > 0 && diag.at(i - 1) + 1 == diag.at(i);
        if (run) m_count++; else if (diag.at(i) == -2) m_count = 0;
        y = std::max(std::min(y1, y2), 0);
        if (run && m_count >= width) prev_y = y;
        y = std::max(std::min(y, prev_y), 0);
        upper->push_back(y);
    }
    y1 = y2 = prev_y = length2; m_count = 0;
    std::vector<int> low(length1 + 1);
    for (int i = length1; i >= 0; i--) {
        if (i <= length1 - width && diag.at(i + width) >= 0) y1 = diag.at(i + width);
        if (diag.at(i) >= 0) y2 = diag.at(i) + width;
        bool run = diag.at(i) >= 0 && i < length1 && diag.at(i + 1) - 1 == diag.at(i);
        if (run) m_count++; else if (diag.at(i) == -2) m_count = 0;
        y = std::min(std::max(y1, y2), length2);
        if (run && m_count >= width) prev_y = y;
        y = std::min(std::max(y, prev_y), length2);
        low[i] = y;
    }
    *lower = low;
}


// ---- the tunnel from overlapping hits + forced gaps ---------------------------------------------------------------
// find_anchors.cpp:497-632 (eliminate_bad_hits and its predicates), :643-843 (define_tunnel_with_overlapping_hits),
// find_anchors.h:38-70 (Coord, Tunnel_block), viterbi_alignment.cpp:467-553 (replace_largest_tunnel_block_with_gap_tunnel).
struct Coord { int x = -1, y = -1; };
struct Tunnel_block {
    Coord start, end;
    long long size() const { return (long long)(end.x - start.x) * (long long)(end.y - start.y); }
};

int overlapsAtBegin(Hit &hit, Hit &subject) {
    int overlap = 0;
    if (hit.s1 >= subject.s1 && hit.s1 + hit.len > subject.s1 + subject.len) overlap = std::max(overlap, subject.s1 + subject.len - hit.s1);
    if (hit.s2 >= subject.s2 && hit.s2 + hit.len > subject.s2 + subject.len) overlap = std::max(overlap, subject.s2 + subject.len - hit.s2);
    return std::max(0, overlap);
}
unsigned int hit_distance(Hit &hit, Hit &subject) { return abs((subject.s1 - subject.s2) - (hit.s1 - hit.s2)); }
bool probaplyBadHit(Hit &hit, Hit &subject) {
    if (hit.s1 < subject.s1 && hit.s2 > subject.s2 && hit.s1 + hit.len < subject.s1 + subject.len) return true;
    if (hit.s1 > subject.s1 && hit.s2 < subject.s2 && hit.s2 + hit.len < subject.s2 + subject.len) return true;
    return false;
}
bool totallyOverlappingHit(Hit &hit, Hit &subject) {
    if (hit.s1 >= subject.s1 && hit.s1 + hit.len <= subject.s1 + subject.len) return true;
    if (hit.s2 >= subject.s2 && hit.s2 + hit.len <= subject.s2 + subject.len) return true;
    return false;
}
bool partlyOverlappingHit(Hit &hit, Hit &subject) { return overlapsAtBegin(hit, subject) || overlapsAtBegin(subject, hit); }

// The reference keeps pointers to the good hits inside the vector it erases from; the good hits always lie before the
// erased position, so they stay put.  Indices here.
void eliminate_bad_hits(std::vector<Hit> &hits, unsigned int threshold_totally_overlapping, unsigned int threshold_partly_overlapping) {
    std::vector<size_t> good_hits;
    size_t hit = 0;
    while (hit < hits.size()) {
        bool bad_hit = false, decent_hit = false;
        for (size_t s : good_hits) {
            if (probaplyBadHit(hits[hit], hits[s]) || totallyOverlappingHit(hits[hit], hits[s])) {
                if (hit_distance(hits[hit], hits[s]) > threshold_totally_overlapping) { bad_hit = true; break; }
                else decent_hit = true;
            } else if (partlyOverlappingHit(hits[hit], hits[s])) {
                if (hit_distance(hits[hit], hits[s]) > threshold_partly_overlapping) { bad_hit = true; break; }
            }
        }
        if (bad_hit) hits.erase(hits.begin() + hit);
        else { if (!decent_hit) good_hits.push_back(hit); ++hit; }
    }
}

void define_tunnel_with_overlapping_hits(std::vector<Hit> &hits, std::vector<int> &upper, std::vector<int> &lower,
                                         const std::string &sequence1, const std::string &sequence2, int width,
                                         std::vector<Tunnel_block> &empty_blocks) {
    int l1 = sequence1.length(), l2 = sequence2.length();
    std::vector<int> i1, i2;
    for (int i = 0; i < l1; i++) if (sequence1.at(i) != '-') i1.push_back(i + 1);
    for (int i = 0; i < l2; i++) if (sequence2.at(i) != '-') i2.push_back(i + 1);
    std::vector<int> lowest_points(l1 + 1), highest_points(l1 + 1);
    int min_height = 0, max_height = l2;
    for (int i = 0; i <= l1; i++) { lowest_points[i] = max_height + 1; highest_points[i] = min_height - 1; }
    for (auto hit = hits.begin(); hit != hits.end(); ++hit)
        for (int a = 0; a < hit->len; a++) {
            if (i2.at(hit->s2 + a) < lowest_points[i1.at(hit->s1 + a)]) lowest_points[i1.at(hit->s1 + a)] = std::max(i2.at(hit->s2 + a), min_height);
            if (i2.at(hit->s2 + a) > highest_points[i1.at(hit->s1 + a)]) highest_points[i1.at(hit->s1 + a)] = std::min(i2.at(hit->s2 + a), max_height);
        }
    int previous_lowest = min_height, previous_highest = max_height;
    previous_highest = highest_points[0];
    for (int i = 0; i <= l1; i++)
        if (highest_points[i] > min_height) {
            if (highest_points[i] < previous_highest) highest_points[i] = previous_highest;
            previous_highest = highest_points[i];
        }
    previous_lowest = lowest_points[l1];
    for (int i = l1; i >= 0; i--)
        if (lowest_points[i] < max_height) {
            if (lowest_points[i] > previous_lowest) lowest_points[i] = previous_lowest;
            previous_lowest = lowest_points[i];
        }
    Tunnel_block current_block;
    current_block.start.x = 0; current_block.start.y = 0;
    for (int i = 1; i <= l1; i++) {
        if (highest_points[i - 1] >= min_height && highest_points[i] < min_height) {
            current_block.start.x = i; current_block.start.y = highest_points[i - 1];
        } else if (highest_points[i] >= min_height && highest_points[i - 1] < min_height) {
            if (lowest_points[i] > current_block.start.y) {
                current_block.end.x = i; current_block.end.y = lowest_points[i];
                if (current_block.size() > 10) empty_blocks.push_back(current_block);
            }
        } else if (i == l1 && highest_points[i] < min_height) {
            if (max_height > current_block.start.y) {
                current_block.end.x = i; current_block.end.y = max_height;
                if (current_block.size() > 10) empty_blocks.push_back(current_block);
            }
        }
    }
    // std::sort in the reference; blocks of equal size have no defined order there -- stable here
    std::stable_sort(empty_blocks.begin(), empty_blocks.end(), [](const Tunnel_block &a, const Tunnel_block &b) { return a.size() < b.size(); });
    previous_lowest = min_height; previous_highest = max_height;
    for (int i = 0; i <= l1; i++) { if (lowest_points[i] >= max_height) lowest_points[i] = previous_lowest; previous_lowest = lowest_points[i]; }
    for (int i = l1; i >= 0; i--) { if (highest_points[i] <= min_height) highest_points[i] = previous_highest; previous_highest = highest_points[i]; }
    lowest_points[0] = min_height;
    highest_points[l1] = max_height;
    for (int i = 0; i <= l1; i++) if (highest_points[i] >= min_height) highest_points[i] = std::min(max_height, highest_points[i] + width);
    for (int i = 0; i <= l1; i++) if (lowest_points[i] <= max_height) lowest_points[i] = std::max(min_height, lowest_points[i] - width);
    std::vector<std::pair<int, bool>> overflow_highest;
    for (int i = 1; i <= l1; i++) {
        if ((i + 1 > l1 || highest_points[i] == highest_points[i + 1]) && highest_points[i - 1] < highest_points[i] - 1) overflow_highest.push_back({i, true});
        else if (highest_points[i - 1] < highest_points[i] - 1) overflow_highest.push_back({i, false});
    }
    for (int a = 0; a < (int)overflow_highest.size(); a++) {
        int i = overflow_highest.at(a).first;
        if (overflow_highest.at(a).second) {
            for (int x = i - 1; x >= i - width && x >= 0 && highest_points[x] >= min_height; x--) highest_points[x] = std::max(highest_points[x], highest_points[i]);
        } else {
            for (int x = i - 1; x >= i - width && x >= 0 && highest_points[x] >= min_height; x--) highest_points[x] = std::max(highest_points[x], highest_points[x + 1] - 1);
        }
    }
    std::vector<std::pair<int, bool>> overflow_lowest;
    for (int i = l1 - 1; i >= 0; i--) {
        if ((i - 1 < 0 || lowest_points[i] == lowest_points[i - 1]) && lowest_points[i + 1] > lowest_points[i] + 1) overflow_lowest.push_back({i, true});
        else if (lowest_points[i + 1] > lowest_points[i] + 1) overflow_lowest.push_back({i, false});
    }
    for (int a = 0; a < (int)overflow_lowest.size(); a++) {
        int i = overflow_lowest.at(a).first;
        if (overflow_lowest.at(a).second) {
            for (int x = i + 1; x <= i + width && x <= l1 && lowest_points[x] <= max_height; x++) lowest_points[x] = std::min(lowest_points[x], lowest_points[i]);
        } else {
            for (int x = i + 1; x <= i + width && x <= l1 && lowest_points[x] <= max_height; x++) lowest_points[x] = std::min(lowest_points[x], lowest_points[x - 1] + 1);
        }
    }
    for (int i = 0; i <= l1; i++) { upper.push_back(lowest_points[i]); lower.push_back(highest_points[i]); }
}

bool replace_largest_tunnel_block_with_gap_tunnel(std::vector<int> &upper_bound, std::vector<int> &lower_bound,
                                                  std::vector<Tunnel_block> &empty_tunnel_blocks, int remove_threshold,
                                                  int tunnel_width, bool wide_tunnel) {
    int tunnel_end = (int)lower_bound.size() - 1;
    if (empty_tunnel_blocks.size() < 1 || empty_tunnel_blocks.back().size() < remove_threshold) return false;
    Tunnel_block &largest_block = empty_tunnel_blocks.back();
    if (wide_tunnel) {
        for (int i = largest_block.start.x; i < largest_block.end.x - tunnel_width; i++) lower_bound.at(i) = largest_block.start.y + tunnel_width;
        for (int i = largest_block.start.x - 1; i >= 0; i--) {
            if (lower_bound.at(i) > lower_bound.at(i + 1)) lower_bound.at(i) = lower_bound.at(i + 1);
            else break;
        }
    } else {
        int a = 0;
        for (int i = largest_block.start.x; i < largest_block.end.x; i++) {
            lower_bound.at(i) = largest_block.start.y;
            upper_bound.at(i) = std::min(largest_block.start.y, upper_bound.at(i) + a);
            a++;
        }
        upper_bound.at(largest_block.end.x) = largest_block.start.y;
        for (int i = largest_block.start.x - 1; i >= 0; i--) {
            if (lower_bound.at(i) > lower_bound.at(i + 1)) lower_bound.at(i) = lower_bound.at(i + 1);
            else break;
        }
        int last_i = std::min(largest_block.end.x + tunnel_width + 1, tunnel_end);
        int b = 0;
        for (int i = last_i; i >= largest_block.end.x + 1; i--) {
            upper_bound.at(i) = std::max(upper_bound.at(last_i) - b, largest_block.start.y);
            b++;
        }
    }
    empty_tunnel_blocks.pop_back();
    return true;
}

} // namespace

extern "C" {

void *oracle_graph_leaf(const char *seq, const char *alphabet, int flags) { return Graph::leaf(seq, alphabet, flags); }
void oracle_graph_free(void *g) { delete (Graph *)g; }
// Site::set_state, for Node::set_ambiguous_state (node.cpp:1661-1690) driven from the tests
void oracle_graph_set_state(void *g, int pos, int state) { ((Graph *)g)->sites[pos].state = state; }
int oracle_graph_n_sites(void *g) { return (int)((Graph *)g)->sites.size(); }
int oracle_graph_n_edges(void *g) { return (int)((Graph *)g)->edges.size(); }
int oracle_graph_n_bwd(void *gp) {
    Graph *g = (Graph *)gp; int n = 0;
    for (size_t s = 0; s < g->sites.size(); s++)
        if (g->has_bwd((int)s)) { g->first_bwd((int)s); n++; while (g->has_next_bwd((int)s)) { g->next_bwd((int)s); n++; } }
    return n;
}
// Flatten to the pagan_graph CSR layout (arrays sized by the caller from the counts above).
void oracle_graph_flatten(void *gp, int32_t *state, int32_t *bwd_off, int32_t *bwd_src, float *bwd_logw, int32_t *bwd_eid) {
    Graph *g = (Graph *)gp; int k = 0;
    for (size_t s = 0; s < g->sites.size(); s++) {
        state[s] = g->sites[s].state; bwd_off[s] = k;
        if (g->has_bwd((int)s)) {
            int e = g->first_bwd((int)s);
            for (;;) {
                bwd_src[k] = g->edges[e].start; bwd_logw[k] = g->edges[e].logw; bwd_eid[k] = e; k++;
                if (!g->has_next_bwd((int)s)) break;
                e = g->next_bwd((int)s);
            }
        }
    }
    bwd_off[g->sites.size()] = k;
}
// Per-site and per-edge attributes, for comparing graph builders field by field.
// site_attr: [n_sites][8] = state,type,path_state,left,right,count_since_used,ambiguous,n_fwd ; site_dist: [n_sites]
// edge_attr: [n_edges][6] = start,end,used,count_since_used,count_as_skipped,linked ; edge_f: [n_edges][3] = w,logw,dist
void oracle_graph_attrs(void *gp, int32_t *site_attr, float *site_dist, int32_t *edge_attr, float *edge_f) {
    Graph *g = (Graph *)gp;
    std::vector<char> linked(g->edges.size(), 0);
    for (size_t s = 0; s < g->sites.size(); s++) {
        const Site &t = g->sites[s];
        int nf = 0;
        if (g->has_fwd((int)s)) { int e = g->first_fwd((int)s); nf++; linked[e] = 1; while (g->has_next_fwd((int)s)) { e = g->next_fwd((int)s); nf++; linked[e] = 1; } }
        int32_t *a = site_attr + 8 * s;
        a[0] = t.state; a[1] = t.type; a[2] = t.path_state; a[3] = t.left; a[4] = t.right; a[5] = t.count_since_used; a[6] = t.ambiguous; a[7] = nf;
        site_dist[s] = t.dist_since_used;
    }
    for (size_t e = 0; e < g->edges.size(); e++) {
        const Edge &x = g->edges[e];
        int32_t *a = edge_attr + 6 * e;
        a[0] = x.start; a[1] = x.end; a[2] = x.used; a[3] = x.count_since_used; a[4] = x.count_as_skipped; a[5] = linked[e];
        edge_f[3 * e] = x.w; edge_f[3 * e + 1] = x.logw; edge_f[3 * e + 2] = x.dist_since_used;
    }
}
// fwd lists in iteration order, CSR (fwd_off [n_sites+1], fwd_eid [n linked edges])
void oracle_graph_fwd(void *gp, int32_t *fwd_off, int32_t *fwd_eid) {
    Graph *g = (Graph *)gp; int k = 0;
    for (size_t s = 0; s < g->sites.size(); s++) {
        fwd_off[s] = k;
        if (g->has_fwd((int)s)) { int e = g->first_fwd((int)s); fwd_eid[k++] = e; while (g->has_next_fwd((int)s)) { e = g->next_fwd((int)s); fwd_eid[k++] = e; } }
    }
    fwd_off[g->sites.size()] = k;
}
void oracle_graph_mark_used(void *gp, int n, const int32_t *eids) {
    Graph *g = (Graph *)gp;
    for (int k = 0; k < n; k++) g->edges.at(eids[k]).used = true;
}
// Builds the parent graph from the two children (with their used flags already marked) and
// the alignment columns.  flags: bit0 = reads / --keep-all-edges settings (BA.h:572-586),
// bit1 = --no-reduced-terminal-penalties.
void *oracle_graph_parent(void *lp, void *rp, const pagan_col *cols, int n_cols, float lbl, float rbl,
                          const int32_t *parsimony, int S, int char_as, int flags) {
    Builder b;
    b.left = (Graph *)lp; b.right = (Graph *)rp; b.lbl = lbl; b.rbl = rbl;
    if (flags & 1) { b.o.max_skip_distance = 5; b.o.max_skip_branches = 50000; b.o.max_match_skip_branches = 50000; b.o.branch_skip_probability = 1; }
    if (flags & 2) b.o.reduced_terminal = false;
    b.seq = new Graph();
    b.create_sites(cols, n_cols, parsimony, S, char_as);
    b.create_edges();
    b.check_skipped_boundaries();
    return b.seq;
}
// get_sequence_string into a caller buffer of n_sites bytes; returns length.
int oracle_graph_string(void *gp, int with_gaps, const char *alphabet, char *out) {
    std::string s = ((Graph *)gp)->sequence_string(with_gaps != 0, alphabet);
    std::memcpy(out, s.data(), s.size()); out[s.size()] = 0;
    return (int)s.size();
}

// Viterbi_alignment::define_tunnel with --use-prefix-anchors (viterbi_alignment.cpp:47-72,138-164):
// ungapped strings -> prefix hits -> order check -> bounds over the gapped strings.
// upper/lower must hold strlen(gapped1)+1 entries.  Returns number of surviving hits.
int oracle_define_tunnel(const char *s1, const char *s2, const char *g1, const char *g2,
                         int min_length, int trim, int width, int32_t *upper, int32_t *lower) {
    std::vector<Hit> hits;
    prefix_hits(s1, s2, min_length, &hits);
    order_conflicts((int)std::strlen(g1), (int)std::strlen(g2), trim, &hits);
    std::vector<int> up, lo;
    tunnel(hits, g1, g2, width, &up, &lo);
    for (size_t i = 0; i < up.size(); i++) { upper[i] = up[i]; lower[i] = lo[i]; }
    return (int)hits.size();
}


// Find_anchors::define_tunnel alone (find_anchors.cpp:320-447) on a given hit list: n x 4 ints (start 1, start 2, length,
// score), positions in the UNGAPPED strings; upper/lower hold strlen(g1)+1 entries.  (tests/pycheck_tunnel.py reads the
// same source a second time and is compared with this.)
int oracle_tunnel_from_hits(const int32_t *hits, int n, const char *g1, const char *g2, int width, int32_t *upper, int32_t *lower) {
    std::vector<Hit> v;
    for (int k = 0; k < n; k++) v.push_back({hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]});
    std::vector<int> up, lo;
    tunnel(v, g1, g2, width, &up, &lo);
    for (size_t i = 0; i < up.size(); i++) { upper[i] = up[i]; lower[i] = lo[i]; }
    return (int)up.size();
}

// Find_anchors::check_hits_order_conflict alone (find_anchors.cpp:225-317) on a given hit list (compacted in place; returns
// the surviving count)
int oracle_order_conflicts(int32_t *hits, int n, int len1, int len2, int trim) {
    std::vector<Hit> v;
    for (int k = 0; k < n; k++) v.push_back({hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]});
    order_conflicts(len1, len2, trim, &v);
    for (size_t k = 0; k < v.size(); k++) { hits[4 * k] = v[k].s1; hits[4 * k + 1] = v[k].s2; hits[4 * k + 2] = v[k].len; hits[4 * k + 3] = v[k].score; }
    return (int)v.size();
}

// hits: n x 4 ints (start 1, start 2, length, score) in processing order.  Returns the surviving count (compacted in place).
int oracle_eliminate_bad_hits(int32_t *hits, int n, int thr_total, int thr_partly) {
    std::vector<Hit> v;
    for (int k = 0; k < n; k++) v.push_back({hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]});
    eliminate_bad_hits(v, thr_total, thr_partly);
    for (size_t k = 0; k < v.size(); k++) { hits[4 * k] = v[k].s1; hits[4 * k + 1] = v[k].s2; hits[4 * k + 2] = v[k].len; hits[4 * k + 3] = v[k].score; }
    return (int)v.size();
}

// upper/lower: strlen(g1)+1 entries; blocks: cap x 4 ints (start x, start y, end x, end y), ascending by size.  Returns the block count.
int oracle_tunnel_overlapping(const int32_t *hits, int n, const char *g1, const char *g2, int width, int32_t *upper, int32_t *lower,
                              int32_t *blocks, int cap) {
    std::vector<Hit> v;
    for (int k = 0; k < n; k++) v.push_back({hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]});
    std::vector<int> up, lo;
    std::vector<Tunnel_block> eb;
    define_tunnel_with_overlapping_hits(v, up, lo, g1, g2, width, eb);
    for (size_t i = 0; i < up.size(); i++) { upper[i] = up[i]; lower[i] = lo[i]; }
    for (size_t k = 0; k < eb.size() && (int)k < cap; k++) { blocks[4 * k] = eb[k].start.x; blocks[4 * k + 1] = eb[k].start.y; blocks[4 * k + 2] = eb[k].end.x; blocks[4 * k + 3] = eb[k].end.y; }
    return (int)eb.size();
}

// One round of --force-gap on bounds of n entries and n_blocks blocks (ascending); returns 1 if a block was replaced.
int oracle_force_gap(int32_t *upper, int32_t *lower, int n, const int32_t *blocks, int n_blocks, int threshold, int width, int wide) {
    std::vector<int> up(upper, upper + n), lo(lower, lower + n);
    std::vector<Tunnel_block> eb(n_blocks);
    for (int k = 0; k < n_blocks; k++) { eb[k].start.x = blocks[4 * k]; eb[k].start.y = blocks[4 * k + 1]; eb[k].end.x = blocks[4 * k + 2]; eb[k].end.y = blocks[4 * k + 3]; }
    const bool done = replace_largest_tunnel_block_with_gap_tunnel(up, lo, eb, threshold, width, wide != 0);
    for (int i = 0; i < n; i++) { upper[i] = up[i]; lower[i] = lo[i]; }
    return done ? 1 : 0;
}

// Model_factory::define_dna_alphabet parsimony table (model_factory.cpp:147-227): 15x15,
// table[i + j*15]; intersection of the base sets if non-empty, else their union.
void oracle_dna_parsimony(int32_t *table) {
    const int bits[15] = {1, 2, 4, 8, 1 | 4, 2 | 8, 1 | 2, 4 | 8, 1 | 8, 2 | 4, 2 | 4 | 8, 1 | 4 | 8, 1 | 2 | 8, 1 | 2 | 4, 15};
    int pos[16]; for (int i = 0; i < 16; i++) pos[i] = -1;
    for (int i = 0; i < 15; i++) pos[bits[i]] = i;
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            int v = bits[i] & bits[j];
            table[i + j * 15] = v > 0 ? pos[v] : pos[bits[i] | bits[j]];
        }
}

} // extern "C"
