// host_pileup.cpp -- the pileup chain: reads added one at a time to a growing root alignment.
//
// Counterpart of Reads_aligner::pileup_alignment (src/main/reads_aligner.cpp:151-264) as run by
// `--queryfile reads.fas --pileup-alignment [--homopolymer | --454]` (BASELINE config 1): the first read is
// the one-node reference (input_output_parser.cpp:98-140); for every further read a temporary node gets the
// current root as left child at distance 0.001 and the read as right child at --query-distance
// (create_temp_node / copy_node_details, reads_aligner.h:149-183), is aligned with the reads settings
// (align_sequences_this_node(mf, true): skip limits 5 / 50000, skip probability 1, basic_alignment.h:572-586),
// and the read is kept -- the node becomes the new root -- when its overlap with the reference read and the
// identity of the overlapping columns both exceed their thresholds (compute_read_overlap ->
// read_alignment_scores, reads_aligner.h:211-220, reads_aligner.cpp:3323-3465).  A strictly serial
// caterpillar: every alignment needs the previous one's graph, so there is nothing to farm out; each step is
// one call of the GPU aligner.
//
// Not restated: --both-strands (reverse-complement attempt), the anchoring-threshold shortcut that skips the
// DP when the tunnel covers too much (node.cpp:155-186), fix_branch_lengths (output tree only).
#include <algorithm>
#include <cctype>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pagan_host.h"
#include "host_anchors.h"
#include "host_graph.h"
#include "host_model.h"

using namespace pagan;

namespace {

struct Step {
    int read = -1;
    bool accepted = false;
    float overlap = -1, identity = -1;
    int aligned = 0, matched = 0, read_length = 0;
    pagan_result res;
    bool has_res = false;
    std::shared_ptr<SeqGraph> left;          // the root the read was aligned against
    std::shared_ptr<SeqGraph> node;          // the temporary node's graph (the new root if accepted)
    std::vector<int32_t> upper, lower;
    pagan_graph gl, gr;
    pagan_band pb;
    bool banded = false;
};

} // namespace

struct pagan_pileup {
    pagan_pileup_opts opts;
    std::vector<std::string> names, seqs;
    ModelFactory mf;
    EvolModel model;
    pagan_model pm;
    std::vector<std::shared_ptr<SeqGraph>> leaf;
    std::shared_ptr<SeqGraph> root;
    std::vector<int32_t> ref_site;           // root column -> site of the reference read (read 0), -1 if none
    std::vector<Step> steps;
    std::vector<int> members;                // reads in the alignment, in the order they joined
    bool aligned = false;
    pagan_batch_fn backend = nullptr;
    void *backend_user = nullptr;
    std::vector<std::string> rows;
    ~pagan_pileup() { for (Step &s : steps) if (s.has_res) pagan_result_free(&s.res); }
};

extern "C" {

void pagan_pileup_default_opts(pagan_pileup_opts *o) {
    std::memset(o, 0, sizeof(*o));
    o->leaf_flags = 2;                       // --homopolymer
    o->query_distance = 0.1f;                // settings.cpp:107
    o->min_overlap = 0.5f; o->min_identity = 0.5f;      // settings.cpp:108,110
    o->anchors_offset = 15; o->prefix_hit_length = 30; o->hit_trim = 5;
    o->device = -1;
}

int pagan_pileup_create(int32_t n, const char *const *names, const char *const *seqs, const pagan_pileup_opts *opts,
                        pagan_pileup **out) {
    if (n < 1 || !names || !seqs || !out) return PAGAN_E_ARG;
    std::unique_ptr<pagan_pileup> p(new pagan_pileup());
    if (opts) p->opts = *opts; else pagan_pileup_default_opts(&p->opts);
    for (int k = 0; k < n; ++k) {
        if (!names[k] || !seqs[k]) return PAGAN_E_ARG;
        p->names.push_back(names[k]);
        std::string s;
        for (const char *c = seqs[k]; *c; ++c) {                   // fasta_reader.cpp:138-160, 1196-1255
            char ch = (char)std::toupper((unsigned char)*c);
            if (ch == 'U') ch = 'T';
            if (std::strchr(ModelFactory::dna_full_alphabet(), ch)) s.push_back(ch);
        }
        p->seqs.push_back(s);
    }
    float bf[4];
    ModelFactory::base_frequencies(p->seqs, bf);
    p->mf.init_dna(bf);
    // one model for the whole chain: dist = 0.001 + query distance (reads_aligner.h:151,178); the pileup rates apply
    // with --454 / --homopolymer (model_factory.cpp:1901-1905)
    const double dist = 0.001 + (double)(p->opts.query_distance <= 0 ? 0.001f : std::min(p->opts.query_distance, 0.2f));
    p->model = p->mf.alignment_model(dist, (p->opts.leaf_flags & 3) != 0);
    p->pm = p->model.view();
    for (int k = 0; k < n; ++k) p->leaf.push_back(std::make_shared<SeqGraph>(make_leaf(p->seqs[k], p->mf.leaf_alphabet, p->opts.leaf_flags)));
    *out = p.release();
    return PAGAN_OK;
}

int pagan_pileup_set_batch_backend(pagan_pileup *p, pagan_batch_fn fn, void *user) {
    if (!p) return PAGAN_E_ARG;
    p->backend = fn; p->backend_user = user;
    return PAGAN_OK;
}

int pagan_pileup_align(pagan_pileup *p) {
    if (!p || p->aligned) return PAGAN_E_ARG;
    const int n = (int)p->seqs.size();
    p->root = p->leaf[0];
    p->members.push_back(0);
    p->ref_site.resize(p->root->n_sites());
    for (int s = 0; s < p->root->n_sites(); ++s) p->ref_site[s] = s;
    BuildSettings bs;
    bs.reads_mode();                                                // is_reads_sequence = true
    if (p->opts.dp_flags & PAGAN_OPT_NO_REDUCED_TERMINAL_PEN) bs.reduced_terminal = false;
    pagan_opts po;
    po.flags = p->opts.dp_flags; po.device = p->opts.device;
    const float lbl = 0.001f;
    const float rbl = p->opts.query_distance <= 0 ? 0.001f : std::min(p->opts.query_distance, 0.2f);
    p->steps.reserve(n);
    for (int i = 1; i < n; ++i) {
        p->steps.emplace_back();
        Step &st = p->steps.back();
        st.read = i;
        st.left = p->root;
        SeqGraph &gl = *p->root, &gr = *p->leaf[i];
        st.gl = gl.view(); st.gr = gr.view();
        if (p->opts.use_anchors) {
            AnchorSettings as;
            as.offset = p->opts.anchors_offset; as.prefix_hit_length = p->opts.prefix_hit_length; as.hit_trim = p->opts.hit_trim;
            const std::string &alpha = p->mf.ancestral_alphabet;
            define_tunnel(sequence_string(gl, false, alpha), sequence_string(gr, false, alpha), sequence_string(gl, true, alpha),
                          sequence_string(gr, true, alpha), as, &st.upper, &st.lower);
            st.pb.n = (int32_t)st.upper.size(); st.pb.upper = st.upper.data(); st.pb.lower = st.lower.data();
            st.banded = true;
        }
        pagan_job jb;
        jb.left = &st.gl; jb.right = &st.gr; jb.model = &p->pm; jb.band = st.banded ? &st.pb : nullptr;
        int rc = p->backend ? p->backend(1, &jb, &po, &st.res, p->backend_user) : pagan_dp_align_batch(1, &jb, &po, &st.res);
        if (rc == PAGAN_OK && st.banded && st.res.status == PAGAN_DP_UNREACHABLE) {       // viterbi_alignment.cpp:298-317
            pagan_result_free(&st.res);
            st.banded = false; jb.band = nullptr;
            rc = p->backend ? p->backend(1, &jb, &po, &st.res, p->backend_user) : pagan_dp_align_batch(1, &jb, &po, &st.res);
        }
        if (rc != PAGAN_OK) return rc;
        st.has_res = true;
        if (st.res.status != PAGAN_DP_REACHED) continue;                                  // nothing to score: the read is dropped
        // make_parent marks the root's edges the path used (backtrack_new_path's side effect).  The reference never clears
        // Edge::used, so the marks of a rejected attempt stay on the root and count as "used" in later attempts: kept.
        st.node = std::make_shared<SeqGraph>(make_parent(gl, gr, st.res, lbl, rbl, p->mf.parsimony.data(), p->mf.S, p->mf.char_as, bs));
        // read_alignment_scores (reads_aligner.cpp:3405-3465), columns 1 .. sites-1: the stop column counts as a column
        // both have (has_site_at_alignment_column is true for a node asked about itself, node.h:1107-1113), with state -1
        const SeqGraph &g = *st.node;
        const SeqGraph &ref = *p->leaf[0];
        for (int j = 1; j < g.n_sites(); ++j) {
            const int lj = g.child_l[j], rj = g.child_r[j];
            const bool read_has = rj >= 0;
            const int rs = lj >= 0 ? p->ref_site[lj] : -1;
            const bool ref_has = rs >= 0;
            if (read_has && ref_has) {
                const int state_read = gr.state[rj], state_ref = ref.state[rs];
                if (state_read >= 0 && state_read == state_ref) st.matched++;
                st.aligned++;
            }
            if (read_has) st.read_length++;
        }
        st.overlap = (float)st.aligned / (float)st.read_length;
        st.identity = (float)st.matched / (float)st.aligned;
        const float min_ov = p->opts.min_overlap < 0 ? 0 : p->opts.min_overlap, min_id = p->opts.min_identity < 0 ? 0 : p->opts.min_identity;
        if (st.overlap > min_ov && st.identity > min_id) {                               // reads_aligner.cpp:221
            st.accepted = true;
            std::vector<int32_t> next(g.n_sites(), -1);
            for (int j = 0; j < g.n_sites(); ++j) next[j] = g.child_l[j] >= 0 ? p->ref_site[g.child_l[j]] : -1;
            p->ref_site.swap(next);
            p->root = st.node;
            p->members.push_back(i);
        }
    }
    // rows of the final alignment (Node::get_alignment over the caterpillar): walk the accepted nodes from the root down
    const int width = p->root->n_sites() - 2;
    p->rows.assign(n, std::string());
    std::vector<int32_t> col(p->root->n_sites());
    for (int s = 0; s < p->root->n_sites(); ++s) col[s] = s - 1;
    const SeqGraph *cur = p->root.get();
    for (int k = (int)p->steps.size() - 1; k >= 0; --k) {
        const Step &st = p->steps[k];
        if (!st.accepted) continue;
        const SeqGraph &g = *st.node;
        std::string row(width, '-');
        std::vector<int32_t> lcol(st.left->n_sites(), -1);
        for (int s = 1; s < g.n_sites() - 1; ++s) {
            if (g.child_r[s] >= 0) row[col[s]] = p->leaf[st.read]->symbols[g.child_r[s] - 1];
            if (g.child_l[s] >= 0) lcol[g.child_l[s]] = col[s];
        }
        p->rows[st.read] = row;
        col.swap(lcol);
        cur = st.left.get();
    }
    {
        std::string row(width, '-');
        for (int s = 1; s < cur->n_sites() - 1; ++s) row[col[s]] = p->leaf[0]->symbols[s - 1];
        p->rows[0] = row;
    }
    p->aligned = true;
    return PAGAN_OK;
}

int pagan_pileup_n_steps(const pagan_pileup *p) { return p ? (int)p->steps.size() : PAGAN_E_ARG; }

int pagan_pileup_step_info(const pagan_pileup *p, int32_t k, pagan_pileup_step *o) {
    if (!p || !o || k < 0 || k >= (int)p->steps.size()) return PAGAN_E_ARG;
    const Step &s = p->steps[k];
    std::memset(o, 0, sizeof(*o));
    o->read = s.read; o->accepted = s.accepted ? 1 : 0; o->overlap = s.overlap; o->identity = s.identity;
    o->aligned = s.aligned; o->matched = s.matched; o->read_length = s.read_length;
    o->left_sites = s.gl.n_sites; o->right_sites = s.gr.n_sites;
    if (s.has_res) { o->status = s.res.status; o->score = s.res.score; o->cells = s.res.cells; o->n_cols = s.res.n_cols; }
    return PAGAN_OK;
}

int pagan_pileup_step_job(const pagan_pileup *p, int32_t k, pagan_job *o) {
    if (!p || !o || k < 0 || k >= (int)p->steps.size()) return PAGAN_E_ARG;
    const Step &s = p->steps[k];
    o->left = &s.gl; o->right = &s.gr; o->model = &p->pm; o->band = s.banded ? &s.pb : nullptr;
    return PAGAN_OK;
}

int pagan_pileup_step_result(const pagan_pileup *p, int32_t k, pagan_result *o) {
    if (!p || !o || k < 0 || k >= (int)p->steps.size() || !p->steps[k].has_res) return PAGAN_E_ARG;
    *o = p->steps[k].res;
    return PAGAN_OK;
}

int pagan_pileup_alignment_length(const pagan_pileup *p) { return (p && p->aligned) ? (int)p->rows[0].size() : PAGAN_E_ARG; }

// Row of read k; a read that was not accepted has an empty row (returns 0 and writes an empty string).
int pagan_pileup_alignment_row(const pagan_pileup *p, int32_t k, char *buf) {
    if (!p || !p->aligned || k < 0 || k >= (int)p->rows.size() || !buf) return PAGAN_E_ARG;
    std::memcpy(buf, p->rows[k].c_str(), p->rows[k].size() + 1);
    return (int)p->rows[k].size();
}

void pagan_pileup_destroy(pagan_pileup *p) { delete p; }

} // extern "C"
