// host_graph.cpp -- leaf and parent sequence graphs for the guide-tree walk.
//
// Follows Sequence::create_default_sequence (src/main/sequence.cpp:152-303) and
// Basic_alignment::build_ancestral_sequence (src/main/basic_alignment.cpp:36-653), but on
// flat arrays: edges are created in the reference's order into one edge table, each site's
// bwd/fwd list is a chain through two `next` arrays (append at the tail, like
// Site::add_new_bwd_edge_index, src/main/sequence.h:355-364), and the chains are flattened to
// CSR once at the end.  The per-site list order -- creation order restricted to the site --
// is what the reference's lists hold, and it is what breaks ties in the DP.
#include "host_graph.h"

#include <algorithm>
#include <cmath>

namespace pagan {

namespace {

enum PathState { kEnds = 0, kTerminal = 1, kMatched = 2, kXGapped = 3, kYGapped = 4, kXSkipped = 5, kYSkipped = 6 };

// Edge chains under construction.
struct Chains {
    SeqGraph &g;
    std::vector<int32_t> bhead, btail, fhead, ftail, bnext, fnext;
    explicit Chains(SeqGraph &gr) : g(gr) {
        const size_t n = g.state.size();
        bhead.assign(n, -1); btail.assign(n, -1); fhead.assign(n, -1); ftail.assign(n, -1);
    }
    void reserve_edges(size_t n) {
        g.e_start.reserve(n); g.e_end.reserve(n); g.e_w.reserve(n); g.e_logw.reserve(n);
        g.e_count_since_used.reserve(n); g.e_count_as_skipped.reserve(n); g.e_dist_since_used.reserve(n); g.e_used.reserve(n);
        bnext.reserve(n); fnext.reserve(n);
    }
    int new_edge(int s, int e, float w, float logw) {
        const int id = (int)g.e_start.size();
        g.e_start.push_back(s); g.e_end.push_back(e); g.e_w.push_back(w); g.e_logw.push_back(logw);
        g.e_count_since_used.push_back(0); g.e_count_as_skipped.push_back(0);
        g.e_dist_since_used.push_back(0.0f); g.e_used.push_back(0);
        bnext.push_back(-1); fnext.push_back(-1);
        return id;
    }
    void link(int id) {
        const int s = g.e_start[id], e = g.e_end[id];
        if (ftail[s] < 0) fhead[s] = id; else fnext[ftail[s]] = id;
        ftail[s] = id;
        if (btail[e] < 0) bhead[e] = id; else bnext[btail[e]] = id;
        btail[e] = id;
    }
    // first bwd edge of `site` starting at `start` (Site::contains_bwd_edge, sequence.h:419-450)
    int find_bwd(int site, int start) const {
        for (int e = bhead[site]; e >= 0; e = bnext[e]) if (g.e_start[e] == start) return e;
        return -1;
    }
    void unlink_bwd(int site, int id) {          // Site::delete_bwd_edge, sequence.h:537-581
        int prev = -1;
        for (int e = bhead[site]; e >= 0; prev = e, e = bnext[e])
            if (e == id) {
                if (prev < 0) bhead[site] = bnext[e]; else bnext[prev] = bnext[e];
                if (btail[site] == e) btail[site] = prev;
                return;
            }
    }
    void unlink_fwd(int site, int id) {          // Site::delete_fwd_edge, sequence.h:583-625
        int prev = -1;
        for (int e = fhead[site]; e >= 0; prev = e, e = fnext[e])
            if (e == id) {
                if (prev < 0) fhead[site] = fnext[e]; else fnext[prev] = fnext[e];
                if (ftail[site] == e) ftail[site] = prev;
                return;
            }
    }
    void flatten() {
        const int n = g.n_sites();
        g.bwd_off.assign(n + 1, 0); g.fwd_off.assign(n + 1, 0);
        g.bwd_eid.clear(); g.fwd_eid.clear();
        for (int s = 0; s < n; ++s) {
            g.bwd_off[s] = (int)g.bwd_eid.size();
            for (int e = bhead[s]; e >= 0; e = bnext[e]) g.bwd_eid.push_back(e);
            g.fwd_off[s] = (int)g.fwd_eid.size();
            for (int e = fhead[s]; e >= 0; e = fnext[e]) g.fwd_eid.push_back(e);
        }
        g.bwd_off[n] = (int)g.bwd_eid.size(); g.fwd_off[n] = (int)g.fwd_eid.size();
        g.bwd_src.resize(g.bwd_eid.size()); g.bwd_logw.resize(g.bwd_eid.size());
        for (size_t k = 0; k < g.bwd_eid.size(); ++k) { g.bwd_src[k] = g.e_start[g.bwd_eid[k]]; g.bwd_logw[k] = g.e_logw[g.bwd_eid[k]]; }
    }
};

void reserve_sites(SeqGraph &g, size_t n) {
    g.state.reserve(n); g.site_type.reserve(n); g.path_state.reserve(n); g.child_l.reserve(n); g.child_r.reserve(n);
    g.count_since_used.reserve(n); g.dist_since_used.reserve(n); g.ambiguous.reserve(n);
}

void push_site(SeqGraph &g, int state, int type, int pstate, int cl, int cr) {
    g.state.push_back(state); g.site_type.push_back((int8_t)type); g.path_state.push_back((int8_t)pstate);
    g.child_l.push_back(cl); g.child_r.push_back(cr);
    g.count_since_used.push_back(0); g.dist_since_used.push_back(0.0f); g.ambiguous.push_back(0);
}

} // namespace

SeqGraph make_leaf(const std::string &residues, const std::string &alphabet, int flags) {
    std::vector<int32_t> states;
    std::string symbols;
    states.reserve(residues.size()); symbols.reserve(residues.size());
    for (char c : residues) {
        if (c == '0') continue;                                   // sequence.cpp:173-176
        states.push_back((int32_t)alphabet.find(c));
        symbols.push_back(c);
    }
    return make_leaf_states(states, std::move(symbols), 1, flags);
}

SeqGraph make_leaf_states(const std::vector<int32_t> &states, std::string symbols, int sym_width, int flags) {
    SeqGraph g;
    g.terminal = true;
    g.sym_width = sym_width;
    if (!(flags & (kLeaf454 | kLeafHomopolymer))) {
        // A plain leaf is a chain: site k's one bwd edge is edge k from site k-1 (edge 0 is the reference's unlinked first
        // edge, sequence.cpp:164-165), its one fwd edge is edge k+1.  Written straight into the arrays -- what the general
        // code below builds edge by edge (a 100 kb leaf: 11 ms there, 32 of them per walk).
        const int n = (int)states.size() + 2;
        g.symbols = std::move(symbols);
        g.state.resize(n); g.state[0] = -1; g.state[n - 1] = -1;
        std::copy(states.begin(), states.end(), g.state.begin() + 1);
        g.site_type.assign(n, (int8_t)kRealSite); g.site_type[0] = (int8_t)kStartSite; g.site_type[n - 1] = (int8_t)kStopSite;
        g.path_state.assign(n, (int8_t)kTerminal); g.path_state[0] = (int8_t)kEnds; g.path_state[n - 1] = (int8_t)kEnds;
        g.child_l.assign(n, -1); g.child_r.assign(n, -1);
        g.count_since_used.assign(n, 0); g.dist_since_used.assign(n, 0.0f); g.ambiguous.assign(n, 0);
        g.e_start.resize(n); g.e_end.resize(n);
        for (int e = 0; e < n; ++e) { g.e_start[e] = e - 1; g.e_end[e] = e; }
        g.e_w.assign(n, 1.0f); g.e_logw.assign(n, 0.0f);
        g.e_count_since_used.assign(n, 0); g.e_count_as_skipped.assign(n, 0); g.e_dist_since_used.assign(n, 0.0f); g.e_used.assign(n, 0);
        g.bwd_off.resize(n + 1); g.fwd_off.resize(n + 1);
        g.bwd_off[0] = 0;
        for (int k = 1; k <= n; ++k) g.bwd_off[k] = k - 1;
        for (int k = 0; k < n; ++k) g.fwd_off[k] = k;
        g.fwd_off[n] = n - 1;
        g.bwd_eid.resize(n - 1); g.fwd_eid.resize(n - 1); g.bwd_src.resize(n - 1);
        for (int k = 0; k < n - 1; ++k) { g.bwd_eid[k] = k + 1; g.fwd_eid[k] = k + 1; g.bwd_src[k] = k; }
        g.bwd_logw.assign(n - 1, 0.0f);
        return g;
    }
    reserve_sites(g, states.size() + 2);
    g.symbols = std::move(symbols);
    push_site(g, -1, kStartSite, kEnds, -1, -1);
    for (int32_t st : states) push_site(g, st, kRealSite, kTerminal, -1, -1);
    push_site(g, -1, kStopSite, kEnds, -1, -1);
    const int n = g.n_sites();
    Chains ch(g);
    ch.reserve_edges((size_t)n + 8);      // plain leaves: one edge per site (the 454 / homopolymer modes add a few)
    // Edge 0 is created but never linked (sequence.cpp:164-165); it keeps Edge::index of the
    // chain edge into site k equal to k for plain leaves.
    ch.new_edge(-1, 0, 1.0f, 0.0f);
    int in_row = 1, prev_row = 1, prev_state = -1;
    for (int cur = 1; cur < n - 1; ++cur) {
        const int prev = cur - 1;
        if (g.state[cur] == prev_state) { in_row++; prev_row = 1; }
        else { prev_row = in_row; in_row = 1; prev_state = g.state[cur]; }
        // first bwd edge of a leaf site is always the chain edge, so its start is site-1
        if ((flags & kLeaf454) && prev_row > 2) {                 // sequence.cpp:205-249
            ch.link(ch.new_edge(prev, cur, 1.0f, std::log(1.0f)));
            const int p1 = g.e_start[ch.bhead[prev]];
            ch.link(ch.new_edge(p1, cur, 0.9f, std::log(0.9f)));
            if (prev_row >= 5) {
                const int p2 = g.e_start[ch.bhead[p1]];
                ch.link(ch.new_edge(p2, cur, 0.9f, std::log(0.9f)));
            }
        } else if ((flags & kLeafHomopolymer) && prev_row >= 2) { // sequence.cpp:253-278
            ch.link(ch.new_edge(prev, cur, 1.0f, std::log(1.0f)));
            int p = g.e_start[ch.bhead[prev]];
            for (int r = prev_row; r >= 2; --r) {
                ch.link(ch.new_edge(p, cur, 0.25f, std::log(0.25f)));
                p = g.e_start[ch.bhead[p]];
            }
        } else {
            ch.link(ch.new_edge(prev, cur, 1.0f, 0.0f));          // sequence.cpp:280-287
        }
    }
    ch.link(ch.new_edge(n - 2, n - 1, 1.0f, 0.0f));               // sequence.cpp:297-301
    ch.flatten();
    return g;
}

SeqGraph make_parent(SeqGraph &left, SeqGraph &right, const pagan_result &res, float lbl, float rbl,
                     const int32_t *parsimony, int S, int char_as, const BuildSettings &bs) {
    for (int k = 0; k < res.n_left_used; ++k) left.e_used[res.left_used[k]] = 1;
    for (int k = 0; k < res.n_right_used; ++k) right.e_used[res.right_used[k]] = 1;

    SeqGraph g;
    g.sym_width = left.sym_width;
    const int n = res.n_cols + 2;
    reserve_sites(g, n);
    // ---- sites: create_ancestral_sequence, basic_alignment.cpp:61-179 ----
    std::vector<int32_t> lci(left.n_sites()), rci(right.n_sites());   // child site -> parent site
    push_site(g, -1, kStartSite, kEnds, 0, 0);
    lci[0] = 0; rci[0] = 0;
    for (int k = 0; k < res.n_cols; ++k) {
        const pagan_col &c = res.cols[k];
        const int i = k + 1;
        if (c.path_state == kMatched) {
            const int lc = left.state[c.left], rc = right.state[c.right];
            push_site(g, parsimony[lc + rc * S], kRealSite, kMatched, c.left, c.right);
            if (lc != rc || lc >= char_as) g.ambiguous[i] = 1;
            lci[c.left] = i; rci[c.right] = i;
        } else if (c.path_state == kXGapped || c.path_state == kXSkipped) {
            push_site(g, left.state[c.left], kRealSite, c.path_state, c.left, -1);
            g.ambiguous[i] = left.ambiguous[c.left];
            if (c.path_state == kXSkipped) {
                g.count_since_used[i] = left.count_since_used[c.left] + 1;
                g.dist_since_used[i] = left.dist_since_used[c.left] + lbl;
            }
            lci[c.left] = i;
        } else {
            push_site(g, right.state[c.right], kRealSite, c.path_state, -1, c.right);
            g.ambiguous[i] = right.ambiguous[c.right];
            if (c.path_state == kYSkipped) {
                g.count_since_used[i] = right.count_since_used[c.right] + 1;
                g.dist_since_used[i] = right.dist_since_used[c.right] + rbl;
            }
            rci[c.right] = i;
        }
    }
    push_site(g, -1, kStopSite, kEnds, left.n_sites() - 1, right.n_sites() - 1);
    lci[left.n_sites() - 1] = n - 1; rci[right.n_sites() - 1] = n - 1;

    // ---- edges: create_ancestral_edges, basic_alignment.cpp:181-368 ----
    Chains ch(g);
    ch.reserve_edges((size_t)left.n_edges() + right.n_edges() + 16);
    // transfer_child_edge, basic_alignment.cpp:510-653 (weight_edges / pair_end_reads off)
    auto transfer = [&](const SeqGraph &child, int ce, const std::vector<int32_t> &ci, float branch_length) {
        int s = ci[child.e_start[ce]], e = ci[child.e_end[ce]];
        const int child_span = child.e_end[ce] - child.e_start[ce];
        if (bs.reduced_terminal) {                                            // :526-541
            if (g.site_type[s] == kStartSite && e - s > 1 && child_span == 1) s = e - 1;
            if (g.site_type[e] == kStopSite && e - s > 1 && child_span == 1) e = s + 1;
        }
        const int dup = ch.find_bwd(e, s);
        if (dup >= 0) {                                                       // :579-583, sequence.h:452-502
            for (int x = ch.bhead[e]; x >= 0; x = ch.bnext[x])
                if (g.e_start[x] == s) {
                    g.e_count_as_skipped[x] = 0; g.e_count_since_used[x] = 0; g.e_dist_since_used[x] = 0.0f;
                    g.e_w[x] = 1.0f; g.e_logw[x] = std::log(1.0f);
                }
            return;
        }
        const bool used = child.e_used[ce];
        if (!used && child.e_count_since_used[ce] + 1 > bs.max_skip_branches) return;               // :587
        if (!used && child.e_dist_since_used[ce] + branch_length > bs.max_skip_distance) return;    // :591
        const float dist_s = g.dist_since_used[s], dist_e = g.dist_since_used[e];
        const int cnt_s = g.count_since_used[s], cnt_e = g.count_since_used[e];
        float w = 1.0f, dist = 0.0f;
        int cnt = 0;
        const float factor = (float)(1.0f * (double)child.e_w[ce] * bs.branch_skip_probability);    // :613
        if (dist_s != dist_e || cnt_s != cnt_e) {                                                   // :604-617
            dist = std::max(dist_s, dist_e); cnt = std::max(cnt_s, cnt_e); w *= factor;
        } else if (!used && cnt_s == 0 && cnt_e == 0) {                                             // :619-632
            dist = child.e_dist_since_used[ce] + branch_length; cnt = child.e_count_since_used[ce] + 1; w *= factor;
        } else if (!used) {                                                                         // :633-637
            dist = child.e_dist_since_used[ce] + branch_length; cnt = child.e_count_since_used[ce] + 1;
        }
        const int id = ch.new_edge(s, e, w, std::log(w));
        g.e_dist_since_used[id] = dist; g.e_count_since_used[id] = cnt;
        g.e_count_as_skipped[id] = used ? 0 : child.e_count_as_skipped[ce];                         // :643-646
        ch.link(id);
    };
    int prev_state = -1;
    for (int i = 1; i < n; ++i) {
        const int ps = g.path_state[i];
        const int li = g.child_l[i], ri = g.child_r[i];
        if (li >= 0) {
            for (int k = left.bwd_off[li]; k < left.bwd_off[li + 1]; ++k) transfer(left, left.bwd_eid[k], lci, lbl);
            if ((ps == kXGapped || ps == kXSkipped) && (prev_state == kYGapped || prev_state == kYSkipped))
                ch.link(ch.new_edge(i - 1, i, 1.0f, std::log(1.0f)));                               // :288-296
        }
        if (ri >= 0) {
            for (int k = right.bwd_off[ri]; k < right.bwd_off[ri + 1]; ++k) transfer(right, right.bwd_eid[k], rci, rbl);
            if ((ps == kYGapped || ps == kYSkipped) && (prev_state == kXGapped || prev_state == kXSkipped))
                ch.link(ch.new_edge(i - 1, i, 1.0f, std::log(1.0f)));                               // :351-358
        }
        prev_state = ps;
    }

    // ---- check_skipped_boundaries, basic_alignment.cpp:370-489 ----
    auto max_start_bwd = [&](int s) {                 // largest start index, first wins ties (:383-390)
        int best = ch.bhead[s];
        for (int e = ch.bnext[best]; e >= 0; e = ch.bnext[e]) if (g.e_start[e] > g.e_start[best]) best = e;
        return best;
    };
    auto skipped = [](int p) { return p == kXSkipped || p == kYSkipped; };
    for (int i = 0; i < n; ++i) {
        const int ts = g.path_state[i];
        if (ch.bhead[i] >= 0) {
            const int e = max_start_bwd(i);
            const int ps = g.path_state[g.e_start[e]];
            if ((ps == kMatched || ps == 0) && skipped(ts)) g.e_count_as_skipped[e]++;             // :394-398
        }
        if (ch.fhead[i] >= 0) {
            const int e = ch.fhead[i];                // all fwd edges share the start, so the first stays (:403-410)
            const int ns = g.path_state[g.e_end[e]];
            if (skipped(ts) && (ns == kMatched || ns == kEnds)) g.e_count_as_skipped[e]++;          // :414-418
        }
    }
    bool non_skipped = true;
    int skip_start = -1;
    for (int i = 1; i < n; ++i) {
        const int ts = g.path_state[i];
        if (non_skipped && skipped(ts)) {
            if (ch.bhead[i] >= 0 && g.e_count_as_skipped[max_start_bwd(i)] > bs.max_match_skip_branches) skip_start = i;
            non_skipped = false;
        }
        if (!non_skipped && skip_start >= 0 && ts == kMatched) {
            int edge_ind = -1;
            for (int e = ch.bhead[i]; e >= 0; e = ch.bnext[e])
                if (g.e_count_as_skipped[e] > bs.max_match_skip_branches) edge_ind = e;              // last such edge (:456-470)
            if (edge_ind >= 0) {                                                                    // delete_edge_range :491-508
                for (int s = g.e_start[edge_ind]; s >= skip_start; --s) {
                    g.site_type[s] = kNonReal;
                    for (int e = ch.bhead[s]; e >= 0; e = ch.bnext[e]) ch.unlink_fwd(g.e_start[e], e);
                    ch.bhead[s] = ch.btail[s] = -1;
                    for (int e = ch.fhead[s]; e >= 0; e = ch.fnext[e]) ch.unlink_bwd(g.e_end[e], e);
                    ch.fhead[s] = ch.ftail[s] = -1;
                }
            }
            non_skipped = true; skip_start = -1;
        }
        if (ts == kXGapped || ts == kYGapped || ts == kMatched) { non_skipped = true; skip_start = -1; }
    }
    ch.flatten();
    return g;
}

std::string sequence_string(const SeqGraph &g, bool with_gaps, const std::string &alphabet) {
    if (g.terminal) return g.symbols;
    std::string out;
    const int n = g.n_sites(), w = g.sym_width;
    out.reserve((size_t)n * w);
    for (int j = 1; j < n - 1; ++j) {
        const int ps = g.path_state[j];
        if (ps != kXSkipped && ps != kYSkipped && g.site_type[j] != kNonReal) out.append(alphabet, (size_t)g.state[j] * w, w);
        else if (with_gaps) out.append(w, '-');                  // "---" for codons, sequence.cpp:731-734
    }
    return out;
}

} // namespace pagan
