// host_tree.cpp -- guide-tree walk and the C ABI of include/pagan_host.h.
//
// Counterpart of Node::start_openmp_alignment / align_sequences_this_node
// (src/main/node.cpp:52-285): nodes whose two children carry a sequence graph are "ready"
// (build_queues, node.cpp:273-285); each round aligns every ready node -- here as ONE batched
// GPU launch per device instead of one OpenMP task per node -- then builds the parents' graphs
// and promotes the nodes that became ready.  Rounds are the guide tree's levels.  Devices are
// fed by one host thread each from a shared list of ready nodes; there is no exchange step
// between devices (parents are built on the host), hence no collective.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pagan_host.h"
#include "host_anchors.h"
#include "host_graph.h"
#include "host_model.h"

using namespace pagan;

struct pagan_hgraph { SeqGraph g; };

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F> void parallel_for(int n, int threads, F f) {
    if (threads <= 1 || n <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    const int t = std::min(threads, n);
    for (int k = 0; k < t; ++k) pool.emplace_back([&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto &th : pool) th.join();
}

struct TreeNode {
    int left = -1, right = -1, parent = -1;
    double dist = 0;             // distance to parent after correction
    std::string name;
    int leaf_index = -1;         // input sequence index for leaves
};

// Node::set_distance_to_parent, src/main/node.h:122-159 (defaults: no --scale-branches,
// no --real-branches, --truncate-branches 0.2 always active).
double corrected_branch(double d, float truncate) {
    if (d <= 0) d = 0.001;
    if (truncate > 0 && d > truncate) d = truncate;
    return d;
}

// Minimal Newick reader: rooted, strictly binary, names on leaves, optional lengths.
struct Newick {
    const char *p;
    std::vector<TreeNode> *nodes;
    bool ok = true;
    void ws() { while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p; }
    int parse() {
        ws();
        int id;
        if (*p == '(') {
            ++p;
            const int l = parse();
            ws();
            if (*p != ',') { ok = false; return -1; }
            ++p;
            const int r = parse();
            ws();
            if (*p != ')') { ok = false; return -1; }     // multifurcations are not resolved here
            ++p;
            if (!ok) return -1;
            id = (int)nodes->size();
            nodes->push_back(TreeNode());
            (*nodes)[id].left = l; (*nodes)[id].right = r;
            (*nodes)[l].parent = id; (*nodes)[r].parent = id;
            ws();
            while (*p && *p != ':' && *p != ',' && *p != ')' && *p != ';') ++p;   // internal label ignored
        } else {
            const char *s = p;
            while (*p && *p != ':' && *p != ',' && *p != ')' && *p != ';' && *p != '(') ++p;
            if (p == s) { ok = false; return -1; }
            id = (int)nodes->size();
            nodes->push_back(TreeNode());
            (*nodes)[id].name.assign(s, p - s);
            while (!(*nodes)[id].name.empty() && (*nodes)[id].name.back() == ' ') (*nodes)[id].name.pop_back();
        }
        ws();
        if (*p == ':') {
            ++p;
            char *end = nullptr;
            (*nodes)[id].dist = std::strtod(p, &end);
            if (end == p) { ok = false; return -1; }
            p = end;
        }
        return id;
    }
};

struct NodeWork {                // everything one internal node's alignment consumed / produced
    int node = -1, level = 0;
    std::shared_ptr<EvolModel> model;
    std::vector<int32_t> upper, lower;
    int n_hits = 0;
    pagan_graph gl, gr;
    pagan_model pm;
    pagan_band pb;
    bool banded = false;
    pagan_result res;
    bool has_res = false;
};

} // namespace

struct pagan_msa {
    pagan_msa_opts opts;
    std::vector<std::string> names, seqs;
    std::vector<TreeNode> tree;          // as parsed
    int root = -1;
    std::vector<int> id_of_tree;         // tree index -> public node id
    std::vector<int> tree_of_id;         // public id -> tree index
    std::vector<std::unique_ptr<pagan_hgraph>> graph;    // by public node id
    std::vector<NodeWork> work;          // by internal order k (public id = n_leaves + k)
    DnaModelFactory mf;
    pagan_msa_timing tm;
    bool aligned = false;
    int n_leaves = 0;
    std::vector<std::string> rows;
    ~pagan_msa() { for (auto &w : work) if (w.has_res) pagan_result_free(&w.res); }
};

namespace {

int device_budget(const pagan_msa_opts &o, int dev, int64_t *bytes) {
    if (o.device_mem_budget > 0) { *bytes = o.device_mem_budget; return PAGAN_OK; }
    if (hipSetDevice(dev) != hipSuccess) return PAGAN_E_NODEVICE;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return PAGAN_E_NODEVICE;
    *bytes = (int64_t)(0.8 * (double)(fr + (size_t)pagan_dp_cached_device_bytes(dev)));   // idle arenas are reused or dropped
    return PAGAN_OK;
}

// Aligns the nodes `ks` (indices into m->work) on device `dev`, splitting into sub-batches that
// fit the memory budget.
int align_on_device(pagan_msa *m, const std::vector<int> &ks, int dev, double *fill_ms, double *trace_ms) {
    int64_t budget = 0;
    int rc = device_budget(m->opts, dev, &budget);
    if (rc != PAGAN_OK) return rc;
    pagan_opts po;
    po.flags = m->opts.dp_flags; po.device = dev;
    size_t at = 0;
    while (at < ks.size()) {
        std::vector<pagan_job> jobs;
        std::vector<int> which;
        int64_t used = 0;
        while (at < ks.size()) {
            NodeWork &w = m->work[ks[at]];
            const int64_t need = pagan_dp_predict_bytes(w.gl.n_sites, w.gr.n_sites, w.banded ? &w.pb : nullptr);
            if (need < 0) return (int)need;
            if (need > budget) return PAGAN_E_MEMCAP;
            if (!jobs.empty() && used + need > budget) break;
            used += need;
            pagan_job jb; jb.left = &w.gl; jb.right = &w.gr; jb.model = &w.pm; jb.band = w.banded ? &w.pb : nullptr;
            jobs.push_back(jb); which.push_back(ks[at]); ++at;
        }
        std::vector<pagan_result> res(jobs.size());
        rc = pagan_dp_align_batch((int32_t)jobs.size(), jobs.data(), &po, res.data());
        for (size_t k = 0; k < jobs.size(); ++k) { m->work[which[k]].res = res[k]; m->work[which[k]].has_res = true; }
        if (rc != PAGAN_OK) return rc;
        if (!res.empty()) { *fill_ms += res[0].fill_ms; *trace_ms += res[0].trace_ms; }
    }
    return PAGAN_OK;
}

void build_rows(pagan_msa *m) {
    const int n = m->n_leaves;
    const int root_id = m->id_of_tree[m->root];
    const int width = m->graph[root_id]->g.n_sites() - 2;
    std::vector<std::vector<int32_t>> col(m->graph.size());
    col[root_id].resize(width + 2);
    for (int s = 0; s < width + 2; ++s) col[root_id][s] = s - 1;
    m->rows.assign(n, std::string(width, '-'));
    // internal ids grow in post-order, so walking them downwards visits parents first
    for (int id = (int)m->graph.size() - 1; id >= n; --id) {
        const TreeNode &t = m->tree[m->tree_of_id[id]];
        const SeqGraph &g = m->graph[id]->g;
        const int lid = m->id_of_tree[t.left], rid = m->id_of_tree[t.right];
        col[lid].assign(m->graph[lid]->g.n_sites(), -1);
        col[rid].assign(m->graph[rid]->g.n_sites(), -1);
        for (int s = 1; s < g.n_sites() - 1; ++s) {
            if (g.child_l[s] >= 0) col[lid][g.child_l[s]] = col[id][s];
            if (g.child_r[s] >= 0) col[rid][g.child_r[s]] = col[id][s];
        }
        col[id].clear(); col[id].shrink_to_fit();
    }
    for (int id = 0; id < n; ++id) {
        const SeqGraph &g = m->graph[id]->g;
        for (int s = 1; s < g.n_sites() - 1; ++s) m->rows[id][col[id][s]] = g.symbols[s - 1];
    }
}

} // namespace

extern "C" {

// Work-queue assignment shared by the in-process multi-device walk and the one-process-per-GPU
// bench: units sorted by cost (largest first, stable), each to the currently least-loaded worker.
void pagan_assign_units(int32_t n, const int64_t *cost, int32_t n_workers, int32_t *owner) {
    if (n <= 0 || n_workers <= 0) return;
    std::vector<int> order(n);
    for (int k = 0; k < n; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<int64_t> load(n_workers, 0);
    for (int k : order) {
        int best = 0;
        for (int d = 1; d < n_workers; ++d) if (load[d] < load[best]) best = d;
        owner[k] = best; load[best] += cost[k];
    }
}

void pagan_msa_default_opts(pagan_msa_opts *o) {
    std::memset(o, 0, sizeof(*o));
    o->use_anchors = 1; o->anchors_offset = 15; o->prefix_hit_length = 30; o->hit_trim = 5;
    o->truncate_branches = 0.2f;
}

int pagan_msa_create(int32_t n_seqs, const char *const *names, const char *const *seqs, const char *newick,
                     const pagan_msa_opts *opts, pagan_msa **out) {
    if (n_seqs < 2 || !names || !seqs || !newick || !out) return PAGAN_E_ARG;
    std::unique_ptr<pagan_msa> m(new pagan_msa());
    if (opts) m->opts = *opts; else pagan_msa_default_opts(&m->opts);
    m->n_leaves = n_seqs;
    std::map<std::string, int> by_name;
    m->seqs.resize(n_seqs);
    for (int k = 0; k < n_seqs; ++k) {
        if (!names[k] || !seqs[k]) return PAGAN_E_ARG;
        m->names.push_back(names[k]);
        by_name[m->names.back()] = k;
    }
    // per-leaf work (cleaning the residues here, the leaf graphs below) runs on a few threads: 32 x 100 kb
    // leaves are 0.2 s of the tree's 1.3 s otherwise
    auto over_leaves = [&](auto f) {
        const int hw = (int)std::thread::hardware_concurrency();
        int nt = std::max(1, std::min({(int)n_seqs, hw > 0 ? hw : 1, 8}));      // allocation-heavy: oversubscribed, it is slower than serial
        if (const char *e = std::getenv("PAGAN_HOST_THREADS")) nt = std::max(1, std::min(nt, std::atoi(e)));
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back([&] { for (int k = next++; k < n_seqs; k = next++) f(k); });
        for (auto &th : pool) th.join();
    };
    over_leaves([&](int k) {
        std::string s;
        for (const char *p = seqs[k]; *p; ++p) {             // fasta_reader.cpp:138-160,1206: upper case, U->T,
            char c = (char)std::toupper((unsigned char)*p);  // drop what is outside the DNA alphabet
            if (c == 'U') c = 'T';
            if (std::strchr(DnaModelFactory::full_alphabet(), c)) s.push_back(c);
        }
        m->seqs[k].swap(s);
    });
    Newick nw{newick, &m->tree};
    m->root = nw.parse();
    if (!nw.ok || m->root < 0) return PAGAN_E_TREE;
    nw.ws();
    if (*nw.p == ';') ++nw.p;
    int n_leaf_nodes = 0;
    for (auto &t : m->tree) {
        if (t.left < 0) {
            auto it = by_name.find(t.name);
            if (it == by_name.end() || it->second < 0) return PAGAN_E_TREE;
            t.leaf_index = it->second; it->second = -1; ++n_leaf_nodes;
        }
        t.dist = corrected_branch(t.dist, m->opts.truncate_branches);
    }
    if (n_leaf_nodes != n_seqs || (int)m->tree.size() != 2 * n_seqs - 1) return PAGAN_E_TREE;
    // public ids: leaves by input order, internal nodes in parse order -- the parser closes a
    // node after both children, i.e. post-order left->right->self, the reference's alignment
    // order and #k# naming (src/main/node.h:479-495,928-938).
    m->id_of_tree.assign(m->tree.size(), -1);
    m->tree_of_id.assign(m->tree.size(), -1);
    int next_internal = n_seqs;
    for (size_t t = 0; t < m->tree.size(); ++t) {
        const int id = m->tree[t].left < 0 ? m->tree[t].leaf_index : next_internal++;
        m->id_of_tree[t] = id; m->tree_of_id[id] = (int)t;
    }
    m->graph.resize(m->tree.size());
    float bf[4];
    DnaModelFactory::base_frequencies(m->seqs, bf);
    m->mf.init(bf);
    const std::string alpha = DnaModelFactory::full_alphabet();
    over_leaves([&](int k) {
        m->graph[k].reset(new pagan_hgraph());
        m->graph[k]->g = make_leaf(m->seqs[k], alpha, m->opts.leaf_flags);
    });
    m->work.resize(n_seqs - 1);
    std::memset(&m->tm, 0, sizeof(m->tm));
    *out = m.release();
    return PAGAN_OK;
}

int pagan_msa_align(pagan_msa *m) {
    if (!m || m->aligned) return PAGAN_E_ARG;
    const double t_start = now_s();
    const int n = m->n_leaves;
    int threads = m->opts.host_threads > 0 ? m->opts.host_threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    int ndev = m->opts.n_devices;
    if (ndev <= 0) ndev = 1;
    int first_dev = m->opts.first_device;
    if (m->opts.n_devices <= 0) { if (hipGetDevice(&first_dev) != hipSuccess) return PAGAN_E_NODEVICE; }
    const std::string alpha = DnaModelFactory::full_alphabet();
    BuildSettings bs;
    if (m->opts.keep_all_edges) bs.reads_mode();
    if (m->opts.dp_flags & PAGAN_OPT_NO_REDUCED_TERMINAL_PEN) bs.reduced_terminal = false;
    AnchorSettings as;
    as.offset = m->opts.anchors_offset; as.prefix_hit_length = m->opts.prefix_hit_length; as.hit_trim = m->opts.hit_trim;
    std::map<double, std::shared_ptr<EvolModel>> model_cache;    // one table per distinct distance
    std::vector<char> done(2 * n - 1, 0);
    for (int k = 0; k < n; ++k) done[k] = 1;
    int remaining = n - 1, level = 0;
    while (remaining > 0) {
        // build_queues, node.cpp:273-285
        std::vector<int> ready;
        for (int id = n; id < 2 * n - 1; ++id) {
            if (done[id]) continue;
            const TreeNode &t = m->tree[m->tree_of_id[id]];
            if (done[m->id_of_tree[t.left]] && done[m->id_of_tree[t.right]]) ready.push_back(id);
        }
        if (ready.empty()) return PAGAN_E_INTERNAL;
        // models (serialised in the reference too: omp critical, node.cpp:415-416)
        double t0 = now_s();
        for (int id : ready) {
            NodeWork &w = m->work[id - n];
            const TreeNode &t = m->tree[m->tree_of_id[id]];
            const double dist = m->tree[t.left].dist + m->tree[t.right].dist;         // node.cpp:70
            auto it = model_cache.find(dist);
            if (it == model_cache.end())
                it = model_cache.emplace(dist, std::make_shared<EvolModel>(m->mf.alignment_model(dist))).first;
            w.model = it->second; w.node = id; w.level = level;
        }
        m->tm.model_s += now_s() - t0;
        // anchors -> band, per node in parallel
        t0 = now_s();
        parallel_for((int)ready.size(), threads, [&](int r) {
            const int id = ready[r];
            NodeWork &w = m->work[id - n];
            const TreeNode &t = m->tree[m->tree_of_id[id]];
            const SeqGraph &gl = m->graph[m->id_of_tree[t.left]]->g, &gr = m->graph[m->id_of_tree[t.right]]->g;
            w.gl = gl.view(); w.gr = gr.view(); w.pm = w.model->view();
            w.banded = false;
            if (m->opts.use_anchors) {
                w.n_hits = define_tunnel(sequence_string(gl, false, alpha), sequence_string(gr, false, alpha),
                                         sequence_string(gl, true, alpha), sequence_string(gr, true, alpha), as,
                                         &w.upper, &w.lower);
                w.pb.n = (int32_t)w.upper.size(); w.pb.upper = w.upper.data(); w.pb.lower = w.lower.data();
                w.banded = true;
            }
        });
        m->tm.anchors_s += now_s() - t0;
        // DP on the device(s): ready nodes dealt round-robin, largest first, one thread per device
        t0 = now_s();
        std::vector<int> order(ready.size());
        for (size_t r = 0; r < ready.size(); ++r) order[r] = ready[r] - n;
        std::vector<int64_t> cost(m->work.size(), 0);
        for (int k : order) {
            NodeWork &w = m->work[k];
            cost[k] = pagan_dp_count_cells(w.gl.n_sites, w.gr.n_sites, w.banded ? &w.pb : nullptr);
            if (cost[k] < 0) return (int)cost[k];
        }
        std::vector<int64_t> oc(order.size());
        for (size_t r = 0; r < order.size(); ++r) oc[r] = cost[order[r]];
        std::vector<int32_t> owner(order.size());
        pagan_assign_units((int32_t)order.size(), oc.data(), ndev, owner.data());
        std::vector<std::vector<int>> per_dev(ndev);
        for (size_t r = 0; r < order.size(); ++r) per_dev[owner[r]].push_back(order[r]);
        for (auto &v : per_dev) std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        std::vector<int> rcs(ndev, PAGAN_OK);
        std::vector<double> fms(ndev, 0), tms(ndev, 0);
        {
            std::vector<std::thread> feeders;
            for (int d = 0; d < ndev; ++d)
                if (!per_dev[d].empty())
                    feeders.emplace_back([&, d] { rcs[d] = align_on_device(m, per_dev[d], first_dev + d, &fms[d], &tms[d]); });
            for (auto &th : feeders) th.join();
        }
        for (int d = 0; d < ndev; ++d) if (rcs[d] != PAGAN_OK) return rcs[d];
        // "anchored alignment failed: trying again" (viterbi_alignment.cpp:298-317): a node whose
        // end corner is unreachable inside its tunnel is re-aligned over the full matrix.
        std::vector<int> retry;
        for (int k : order)
            if (m->work[k].banded && m->work[k].res.status == PAGAN_DP_UNREACHABLE) retry.push_back(k);
        if (!retry.empty()) {
            for (int k : retry) { m->work[k].banded = false; pagan_result_free(&m->work[k].res); m->work[k].has_res = false; }
            double f = 0, t = 0;
            const int rc = align_on_device(m, retry, first_dev, &f, &t);
            if (rc != PAGAN_OK) return rc;
            fms[0] += f; tms[0] += t;
        }
        m->tm.dp_wall_s += now_s() - t0;
        m->tm.dp_fill_dev_s += *std::max_element(fms.begin(), fms.end()) / 1e3;
        m->tm.dp_trace_dev_s += *std::max_element(tms.begin(), tms.end()) / 1e3;
        // parents
        t0 = now_s();
        std::atomic<int> bad(0);
        parallel_for((int)ready.size(), threads, [&](int r) {
            const int id = ready[r];
            NodeWork &w = m->work[id - n];
            if (w.res.status != PAGAN_DP_REACHED) { bad = 1; return; }
            const TreeNode &t = m->tree[m->tree_of_id[id]];
            SeqGraph &gl = m->graph[m->id_of_tree[t.left]]->g, &gr = m->graph[m->id_of_tree[t.right]]->g;
            m->graph[id].reset(new pagan_hgraph());
            m->graph[id]->g = make_parent(gl, gr, w.res, (float)m->tree[t.left].dist, (float)m->tree[t.right].dist,
                                          m->mf.parsimony.data(), 15, 4, bs);
        });
        m->tm.build_s += now_s() - t0;
        if (bad) return PAGAN_E_INTERNAL;     // unreachable end corner: caller may retry with use_anchors = 0
        for (int id : ready) { done[id] = 1; --remaining; }
        ++level;
    }
    build_rows(m);
    m->tm.total_s = now_s() - t_start;
    m->aligned = true;
    return PAGAN_OK;
}

int pagan_msa_n_internal(const pagan_msa *m) { return m ? m->n_leaves - 1 : 0; }

int pagan_msa_node_info(const pagan_msa *m, int32_t k, pagan_node_info *o) {
    if (!m || !o || k < 0 || k >= m->n_leaves - 1) return PAGAN_E_ARG;
    const NodeWork &w = m->work[k];
    const int id = m->n_leaves + k;
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    std::memset(o, 0, sizeof(*o));
    o->node = id; o->left = m->id_of_tree[t.left]; o->right = m->id_of_tree[t.right];
    o->level = w.level; o->n_hits = w.n_hits;
    o->dist = m->tree[t.left].dist + m->tree[t.right].dist;
    if (w.has_res) {
        o->left_sites = w.gl.n_sites; o->right_sites = w.gr.n_sites;
        o->cells = w.res.cells; o->score = w.res.score; o->status = w.res.status;
    }
    if (m->graph[id]) o->sites = m->graph[id]->g.n_sites();
    return PAGAN_OK;
}

int pagan_msa_node_job(const pagan_msa *m, int32_t k, pagan_job *o) {
    if (!m || !o || k < 0 || k >= m->n_leaves - 1 || !m->work[k].has_res) return PAGAN_E_ARG;
    const NodeWork &w = m->work[k];
    o->left = &w.gl; o->right = &w.gr; o->model = &w.pm; o->band = w.banded ? &w.pb : nullptr;
    return PAGAN_OK;
}

int pagan_msa_node_result(const pagan_msa *m, int32_t k, pagan_result *o) {
    if (!m || !o || k < 0 || k >= m->n_leaves - 1 || !m->work[k].has_res) return PAGAN_E_ARG;
    *o = m->work[k].res;
    return PAGAN_OK;
}

int pagan_msa_timing_get(const pagan_msa *m, pagan_msa_timing *o) {
    if (!m || !o) return PAGAN_E_ARG;
    *o = m->tm;
    return PAGAN_OK;
}

int pagan_msa_alignment_length(const pagan_msa *m) { return (m && m->aligned) ? (int)m->rows[0].size() : PAGAN_E_ARG; }

int pagan_msa_alignment_row(const pagan_msa *m, int32_t leaf, char *buf) {
    if (!m || !m->aligned || leaf < 0 || leaf >= m->n_leaves || !buf) return PAGAN_E_ARG;
    std::memcpy(buf, m->rows[leaf].c_str(), m->rows[leaf].size() + 1);
    return PAGAN_OK;
}

// Fasta_reader::write_fasta (src/utils/fasta_reader.cpp:596-629) over Node::get_alignment's leaf rows
// (src/main/node.cpp:537-575): one entry per leaf in guide-tree order (left to right, as get_leaf_nodes
// collects them), `>name`, the aligned row cut into lines of chars_by_line characters (60 by default).
int pagan_msa_write_fasta(const pagan_msa *m, const char *path, int32_t chars_by_line) {
    if (!m || !m->aligned || !path) return PAGAN_E_ARG;
    const size_t width = chars_by_line > 0 ? (size_t)chars_by_line : 60;
    std::FILE *f = std::fopen(path, "w");
    if (!f) return PAGAN_E_ARG;
    std::vector<int> stack{m->root}, order;
    while (!stack.empty()) {                                   // leaves left to right
        const int t = stack.back();
        stack.pop_back();
        const TreeNode &n = m->tree[t];
        if (n.left < 0) { order.push_back(m->id_of_tree[t]); continue; }
        stack.push_back(n.right);
        stack.push_back(n.left);
    }
    for (int leaf : order) {
        std::fprintf(f, ">%s\n", m->names[leaf].c_str());
        const std::string &row = m->rows[leaf];
        for (size_t at = 0; at < row.size(); at += width) std::fprintf(f, "%.*s\n", (int)std::min(width, row.size() - at), row.c_str() + at);
    }
    return std::fclose(f) == 0 ? PAGAN_OK : PAGAN_E_ARG;
}

void *pagan_msa_node_graph(const pagan_msa *m, int32_t node) {
    if (!m || node < 0 || node >= (int)m->graph.size()) return nullptr;
    return m->graph[node].get();
}

void pagan_msa_destroy(pagan_msa *m) { delete m; }

// ---- host graphs on their own -------------------------------------------------------------
pagan_hgraph *pagan_hgraph_leaf(const char *residues, const char *alphabet, int32_t flags) {
    pagan_hgraph *h = new pagan_hgraph();
    h->g = make_leaf(residues, alphabet, flags);
    return h;
}

pagan_hgraph *pagan_hgraph_parent(pagan_hgraph *l, pagan_hgraph *r, const pagan_result *res, float lbl, float rbl,
                                  const int32_t *parsimony, int32_t S, int32_t char_as, int32_t flags) {
    BuildSettings bs;
    if (flags & 1) bs.reads_mode();
    if (flags & 2) bs.reduced_terminal = false;
    pagan_hgraph *h = new pagan_hgraph();
    h->g = make_parent(l->g, r->g, *res, lbl, rbl, parsimony, S, char_as, bs);
    return h;
}

void pagan_hgraph_view(const pagan_hgraph *g, pagan_graph *out) { *out = g->g.view(); }

void pagan_hgraph_attrs(const pagan_hgraph *h, int32_t *sa, float *sd, int32_t *ea, float *ef) {
    const SeqGraph &g = h->g;
    std::vector<char> linked(g.n_edges(), 0);
    for (int e : g.fwd_eid) linked[e] = 1;
    for (int s = 0; s < g.n_sites(); ++s) {
        int32_t *a = sa + 8 * s;
        a[0] = g.state[s]; a[1] = g.site_type[s]; a[2] = g.path_state[s]; a[3] = g.child_l[s]; a[4] = g.child_r[s];
        a[5] = g.count_since_used[s]; a[6] = g.ambiguous[s]; a[7] = g.fwd_off[s + 1] - g.fwd_off[s];
        sd[s] = g.dist_since_used[s];
    }
    for (int e = 0; e < g.n_edges(); ++e) {
        int32_t *a = ea + 6 * e;
        a[0] = g.e_start[e]; a[1] = g.e_end[e]; a[2] = g.e_used[e]; a[3] = g.e_count_since_used[e];
        a[4] = g.e_count_as_skipped[e]; a[5] = linked[e];
        ef[3 * e] = g.e_w[e]; ef[3 * e + 1] = g.e_logw[e]; ef[3 * e + 2] = g.e_dist_since_used[e];
    }
}

void pagan_hgraph_fwd(const pagan_hgraph *h, int32_t *fwd_off, int32_t *fwd_eid) {
    std::memcpy(fwd_off, h->g.fwd_off.data(), sizeof(int32_t) * h->g.fwd_off.size());
    if (!h->g.fwd_eid.empty()) std::memcpy(fwd_eid, h->g.fwd_eid.data(), sizeof(int32_t) * h->g.fwd_eid.size());
}

int pagan_hgraph_string(const pagan_hgraph *h, int32_t with_gaps, const char *alphabet, char *out) {
    const std::string s = sequence_string(h->g, with_gaps != 0, alphabet);
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

void pagan_hgraph_free(pagan_hgraph *g) { delete g; }

int pagan_define_tunnel(const char *s1, const char *s2, const char *g1, const char *g2, int32_t prefix_hit_length,
                        int32_t hit_trim, int32_t offset, int32_t *upper, int32_t *lower) {
    AnchorSettings as;
    as.prefix_hit_length = prefix_hit_length; as.hit_trim = hit_trim; as.offset = offset;
    std::vector<int32_t> up, lo;
    const int n = define_tunnel(s1, s2, g1, g2, as, &up, &lo);
    std::memcpy(upper, up.data(), sizeof(int32_t) * up.size());
    std::memcpy(lower, lo.data(), sizeof(int32_t) * lo.size());
    return n;
}

int pagan_dna_model(const float bf[4], double distance, float *table, float *params, int32_t *parsimony) {
    DnaModelFactory mf;
    mf.init(bf);
    const EvolModel em = mf.alignment_model(distance);
    std::memcpy(table, em.log_score.data(), sizeof(float) * 225);
    params[0] = em.log_gap_open; params[1] = em.log_gap_ext; params[2] = em.log_gap_end_ext; params[3] = em.log_non_gap;
    if (parsimony) std::memcpy(parsimony, mf.parsimony.data(), sizeof(int32_t) * 225);
    return PAGAN_OK;
}

} // extern "C"
