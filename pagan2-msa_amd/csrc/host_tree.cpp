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
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sched.h>
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
    std::vector<TunnelBlock> blocks;     // empty tunnel blocks, ascending by size (anchor_mode 1; --force-gap takes the last)
    int n_hits = 0, n_forced = 0;
    int imp_l = 0, imp_r = 0;    // sites of the children of a node whose result was imported (pagan_msa_import_result)
    pagan_graph gl, gr;
    pagan_model pm;
    pagan_band pb;
    bool banded = false;
    pagan_result res;
    bool has_res = false;        // res is valid (aligned here, or imported from the rank that aligned it)
    bool has_job = false;        // gl/gr/pm/pb are valid: this process prepared and aligned the node
    int  device = -1;            // device the alignment ran on (-1: another rank)
    bool parent_pending = false; // imported: the result is stored, the parent graph is built when somebody needs it (ensure_graph)
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
    ModelFactory mf;
    pagan_msa_timing tm;
    bool aligned = false;
    int n_leaves = 0;
    std::vector<std::string> rows;
    // walk state (pagan_msa_ready / align_nodes / import_result)
    std::vector<char> done;              // by public node id
    int remaining = 0, rounds = 0;
    std::map<double, std::shared_ptr<EvolModel>> model_cache;    // one table per distinct distance
    std::mutex mu;                       // model cache, timing sums
    std::vector<int32_t> state_table;    // parent state of a matched column: parsimony table, or with --mostcommon the
                                         // most-common table where it is defined (basic_alignment.cpp:146-149)
    pagan_batch_fn backend = nullptr;    // test seam (pagan_msa_set_batch_backend); null = pagan_dp_align_batch
    void *backend_user = nullptr;
    std::atomic<int> parents_built{0};   // parent graphs this process has built (pagan_msa_parents_built)
    std::atomic<int> lazy_err{0};        // first error of a deferred parent build (an imported result that does not fit the child graphs)
    bool rows_built = false;             // m->rows are valid (pagan_msa_finish builds them at once, pagan_msa_finish_lazy on first use)
    std::mutex lazy_mu;                  // deferred builds started from the accessors
    ~pagan_msa() { for (auto &w : work) if (w.has_res) pagan_result_free(&w.res); }
};

namespace {

int device_budget(const pagan_msa *m, int dev, int64_t *bytes) {
    const pagan_msa_opts &o = m->opts;
    if (o.device_mem_budget > 0) { *bytes = o.device_mem_budget; return PAGAN_OK; }
    if (m->backend) { *bytes = (int64_t)1 << 40; return PAGAN_OK; }
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
    int rc = device_budget(m, dev, &budget);
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
        rc = m->backend ? m->backend((int32_t)jobs.size(), jobs.data(), &po, res.data(), m->backend_user)
                        : pagan_dp_align_batch((int32_t)jobs.size(), jobs.data(), &po, res.data());
        if (rc != PAGAN_OK) return rc;            // nothing in res[] is valid then (the library released it)
        for (size_t k = 0; k < jobs.size(); ++k) { m->work[which[k]].res = res[k]; m->work[which[k]].has_res = true; }
        if (!res.empty()) { *fill_ms += res[0].fill_ms; *trace_ms += res[0].trace_ms; }
    }
    return PAGAN_OK;
}

int host_threads_of(const pagan_msa *m);

void build_rows(pagan_msa *m) {
    const int n = m->n_leaves;
    const int root_id = m->id_of_tree[m->root];
    const int width = m->graph[root_id]->g.n_sites() - 2;
    // a node's column map: site -> column of the alignment (-1: none).  Storage that the host's threads touch first (round 5:
    // the maps' 27 MB and the rows' 8 MB of a 32 x 100 kb walk, allocated and filled by one thread, were most of this function's time)
    std::vector<std::unique_ptr<int32_t[]>> col(m->graph.size());
    col[root_id].reset(new int32_t[width + 2]);
    for (int s = 0; s < width + 2; ++s) col[root_id][s] = s - 1;
    const int w = m->mf.type == kCodon ? 3 : 1;                 // characters per column ("---" gaps for codons)
    m->rows.assign(m->graph.size(), std::string());
    const std::string &anc = m->mf.type == kCodon ? m->mf.codon_names : m->mf.ancestral_alphabet;
    // A node's columns follow from its parent's: the tree is walked by DEPTH, the nodes of one depth side by side on the
    // host's threads (round 5: one thread walking all 2n - 1 nodes was 25 ms of a 0.7 s walk of 32 x 100 kb).
    std::vector<std::vector<int>> by_depth;
    {
        std::vector<std::pair<int, int>> stack{{m->root, 0}};      // (tree index, depth)
        while (!stack.empty()) {
            const auto [t, depth] = stack.back();
            stack.pop_back();
            if ((int)by_depth.size() <= depth) by_depth.resize(depth + 1);
            by_depth[depth].push_back(m->id_of_tree[t]);
            const TreeNode &tn = m->tree[t];
            if (tn.left >= 0) { stack.push_back({tn.left, depth + 1}); stack.push_back({tn.right, depth + 1}); }
        }
    }
    const int threads = host_threads_of(m);
    parallel_for((int)m->rows.size(), threads, [&](int id) { m->rows[id].assign((size_t)width * w, '-'); });
    for (const std::vector<int> &ids : by_depth) {
        // the upper depths hold 1, 2, 4 nodes of 10^5 sites each: a node's sites in `parts` ranges (every site writes its own
        // column of its own row and its own entries of the children's column maps)
        const int parts = std::max(1, std::min(threads, (2 * threads) / std::max(1, (int)ids.size())));
        std::vector<int> kids;
        for (int id : ids)
            if (id >= n) {
                const TreeNode &t = m->tree[m->tree_of_id[id]];
                for (int kid : {m->id_of_tree[t.left], m->id_of_tree[t.right]}) {
                    if (kid < n) continue;                     // (a leaf has no map: its parent writes the leaf's row, below)
                    col[kid].reset(new int32_t[m->graph[kid]->g.n_sites()]);
                    kids.push_back(kid);
                }
            }
        parallel_for((int)kids.size() * parts, threads, [&](int task) {
            const int kid = kids[task / parts], part = task % parts, ns = m->graph[kid]->g.n_sites();
            std::fill(col[kid].get() + (long long)ns * part / parts, col[kid].get() + (long long)ns * (part + 1) / parts, -1);
        });
        parallel_for((int)ids.size() * parts, threads, [&](int task) {
            const int id = ids[task / parts], part = task % parts;
            const SeqGraph &g = m->graph[id]->g;
            std::string &row = m->rows[id];
            const int32_t *mine = col[id].get();
            const int n_in = g.n_sites() - 2;                      // sites 1 .. n_in
            const int s_first = 1 + (int)((long long)n_in * part / parts), s_last = 1 + (int)((long long)n_in * (part + 1) / parts);
            if (id < n) {                                          // a leaf: its residues at its columns (only a tree of one leaf comes here)
                if (!mine) return;
                for (int s = s_first; s < s_last; ++s)
                    for (int c = 0; c < w; ++c) row[(size_t)mine[s] * w + c] = g.symbols[(size_t)(s - 1) * w + c];
                return;
            }
            const TreeNode &t = m->tree[m->tree_of_id[id]];
            const int lid = m->id_of_tree[t.left], rid = m->id_of_tree[t.right];
            int32_t *cl = col[lid].get(), *cr = col[rid].get();
            // a child that is a leaf has no map of its own (half of a tree's nodes, and of the maps' pages): its residues go to
            // its row from here, site c of the leaf at this site's column
            const SeqGraph *gl = lid < n ? &m->graph[lid]->g : nullptr, *gr = rid < n ? &m->graph[rid]->g : nullptr;
            std::string *rl = lid < n ? &m->rows[lid] : nullptr, *rr = rid < n ? &m->rows[rid] : nullptr;
            for (int s = s_first; s < s_last; ++s) {
                const int a = g.child_l[s], b = g.child_r[s];
                if (a >= 0) {
                    if (gl) { for (int c = 0; c < w; ++c) (*rl)[(size_t)mine[s] * w + c] = gl->symbols[(size_t)(a - 1) * w + c]; }
                    else cl[a] = mine[s];
                }
                if (b >= 0) {
                    if (gr) { for (int c = 0; c < w; ++c) (*rr)[(size_t)mine[s] * w + c] = gr->symbols[(size_t)(b - 1) * w + c]; }
                    else cr[b] = mine[s];
                }
                // the ancestor's own row (get_alignment_column_at with include_internal_nodes, node.cpp:808-818): its
                // state's character, a gap where the site is skipped or was deleted
                const int ps = g.path_state[s];
                if (!(ps == PAGAN_XSKIPPED || ps == PAGAN_YSKIPPED || g.site_type[s] == kNonReal) && g.state[s] >= 0)
                    for (int c = 0; c < w; ++c) row[(size_t)mine[s] * w + c] = anc[(size_t)g.state[s] * w + c];
            }
        });
        for (int id : ids) col[id].reset();
    }
}

const ModelFactory &codon_factory() {             // the empirical model's eigen solution and 1892 x 1892 tables: made once
    static ModelFactory mf;
    static std::once_flag once;
    std::call_once(once, [] { mf.init_codon(); });
    return mf;
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
    o->force_gap_threshold = 40000; o->overlap_total = 50; o->overlap_partly = 400;      // settings.cpp:180-189
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
    // fasta_reader.cpp:138-160: upper case, no gaps, no line ends
    over_leaves([&](int k) {
        std::string s;
        s.reserve(std::strlen(seqs[k]));
        for (const char *p = seqs[k]; *p; ++p) {
            const char c = (char)std::toupper((unsigned char)*p);
            if (c != '-' && c != '\r' && c != '\n') s.push_back(c);
        }
        m->seqs[k].swap(s);
    });
    // data_type 3: --codons on DNA input (input_output_parser.cpp:517, sequence.cpp:135-136)
    int type = m->opts.data_type == 1 ? kDna : m->opts.data_type == 2 ? kProtein : m->opts.data_type == 3 ? kCodon : ModelFactory::guess_type(m->seqs);
    // Fasta_reader::check_alphabet, fasta_reader.cpp:1180-1297: DNA U->T; protein U->X; what is outside the
    // full alphabet is dropped
    {
        char keep[256];                                            // 0: dropped, else the character it becomes
        std::memset(keep, 0, sizeof(keep));
        for (const char *p = type != kProtein ? ModelFactory::dna_full_alphabet() : ModelFactory::protein_alphabet(); *p; ++p)
            keep[(unsigned char)*p] = *p;
        if (type != kProtein) keep[(unsigned char)'U'] = 'T'; else keep[(unsigned char)'U'] = keep[(unsigned char)'X'] = 'X';
        over_leaves([&](int k) {
            std::string &s = m->seqs[k];
            size_t w = 0;
            for (size_t r = 0; r < s.size(); ++r) { const char c = keep[(unsigned char)s[r]]; if (c) s[w++] = c; }
            s.resize(w);
        });
    }
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
    if (type == kDna) {
        float bf[4];
        ModelFactory::base_frequencies(m->seqs, bf);
        m->mf.init_dna(bf);
    } else if (type == kCodon) {
        m->mf.init_codon();
    } else {
        m->mf.init_protein();
    }
    m->state_table = m->mf.parsimony;
    if (m->opts.mostcommon)      // Evol_model::mostcommon_state: defined over mc_dim x mc_dim states (20 x 20 residues for protein)
        for (int i = 0; i < m->mf.mc_dim; ++i)
            for (int j = 0; j < m->mf.mc_dim; ++j) m->state_table[i + (size_t)j * m->mf.S] = m->mf.mostcommon[i + j * m->mf.mc_dim];
    over_leaves([&](int k) {
        m->graph[k].reset(new pagan_hgraph());
        if (type == kCodon) {
            std::string symbols;
            const std::vector<int32_t> states = ModelFactory::codon_states(m->seqs[k], &symbols);
            m->graph[k]->g = make_leaf_states(states, std::move(symbols), 3, 0);
        } else {
            m->graph[k]->g = make_leaf(m->seqs[k], m->mf.leaf_alphabet, m->opts.leaf_flags);
        }
    });
    m->work.resize(n_seqs - 1);
    m->done.assign(2 * n_seqs - 1, 0);
    for (int k = 0; k < n_seqs; ++k) m->done[k] = 1;
    m->remaining = n_seqs - 1;
    std::memset(&m->tm, 0, sizeof(m->tm));
    *out = m.release();
    return PAGAN_OK;
}

} // extern "C"

namespace {

// The threads the host-side phases use when the caller names none: what this process may actually run on -- the hardware's
// threads, cut down to the affinity mask and to the cgroup's CPU quota -- and at most 16 (round 5: a GPU box reports 256
// hardware threads to a process that owns a 16-core share; 256 threads per phase there made the rows 85-97 ms instead of 20,
// and every phase is memory- or allocation-bound well before 16).
int default_host_threads() {
    static const int cached = [] {
        long long t = (long long)std::thread::hardware_concurrency();
        if (t < 1) t = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) t = std::min<long long>(t, CPU_COUNT(&set));
        long long quota = -1, period = 0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
            std::fclose(f);
        } else {                                                                       // cgroup v1
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
        }
        if (quota > 0 && period > 0) t = std::min(t, std::max(1ll, (quota + period - 1) / period));
        return (int)std::max(1ll, std::min(t, 16ll));
    }();
    return cached;
}
int host_threads_of(const pagan_msa *m) {
    const int threads = m->opts.host_threads > 0 ? m->opts.host_threads : default_host_threads();
    return threads < 1 ? 1 : threads;
}

bool node_ready(const pagan_msa *m, int id) {
    if (m->done[id]) return false;
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    return m->done[m->id_of_tree[t.left]] && m->done[m->id_of_tree[t.right]];
}

int ensure_graph(pagan_msa *m, int id);

// What align_sequences_this_node does before the aligner is called (node.cpp:70-152): the model for
// dist = d_left + d_right (serialised in the reference too: omp critical, node.cpp:415-416), the
// child graphs' views, anchors -> tunnel.
void prepare_node(pagan_msa *m, int id, int round) {
    const int n = m->n_leaves;
    NodeWork &w = m->work[id - n];
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    const double dist = m->tree[t.left].dist + m->tree[t.right].dist;         // node.cpp:70
    const bool pileup_rates = m->opts.pileup_rates != 0;
    {
        std::lock_guard<std::mutex> g(m->mu);
        auto it = m->model_cache.find(dist);
        if (it == m->model_cache.end())
            it = m->model_cache.emplace(dist, std::make_shared<EvolModel>(m->mf.alignment_model(dist, pileup_rates))).first;
        w.model = it->second;
    }
    w.node = id; w.level = round;
    {   // (rank mode: a child that another rank aligned gets its graph now -- and so does whatever is pending below it)
        int rc = ensure_graph(m, m->id_of_tree[t.left]);
        if (rc == PAGAN_OK) rc = ensure_graph(m, m->id_of_tree[t.right]);
        if (rc != PAGAN_OK) { int zero = 0; m->lazy_err.compare_exchange_strong(zero, rc); return; }
    }
    const SeqGraph &gl = m->graph[m->id_of_tree[t.left]]->g, &gr = m->graph[m->id_of_tree[t.right]]->g;
    w.gl = gl.view(); w.gr = gr.view(); w.pm = w.model->view();
    w.banded = false;
    if (m->opts.use_anchors) {
        AnchorSettings as;
        as.offset = m->opts.anchors_offset; as.prefix_hit_length = m->opts.prefix_hit_length; as.hit_trim = m->opts.hit_trim;
        // what the anchors are looked for in: the graph's string, or -- codon graphs -- its translation, one letter per site
        // (viterbi_alignment.cpp:54-60, 141-145)
        const bool codons = m->mf.type == kCodon;
        const std::string &alpha = codons ? m->mf.codon_names : m->mf.ancestral_alphabet;
        auto str = [&](const SeqGraph &g, bool with_gaps) {
            return codons ? ModelFactory::translate_codons(sequence_string(g, with_gaps, alpha)) : sequence_string(g, with_gaps, alpha);
        };
        if (m->opts.anchor_mode == 1) {
            // the reference's BLAST branch from the hit list onwards (viterbi_alignment.cpp:148-157): hits here are the
            // prefix anchors, sorted by length as find_long_substrings leaves them
            std::vector<Hit> hits;
            prefix_hits(str(gl, false), str(gr, false), as.prefix_hit_length, &hits);
            drop_bad_hits(&hits, (unsigned)m->opts.overlap_total, (unsigned)m->opts.overlap_partly);
            w.upper.clear(); w.lower.clear(); w.blocks.clear();
            hits_to_band_overlapping(hits, str(gl, true), str(gr, true), as.offset, &w.upper, &w.lower, &w.blocks);
            w.n_hits = (int)hits.size();
        } else {
            w.n_hits = define_tunnel(str(gl, false), str(gr, false), str(gl, true), str(gr, true), as, &w.upper, &w.lower);
        }
        w.pb.n = (int32_t)w.upper.size(); w.pb.upper = w.upper.data(); w.pb.lower = w.lower.data();
        w.banded = true;
    }
    w.has_job = true;
}

// Node::get_ambiguous_states (node.cpp:1638-1659): the states below an ambiguous site, down to unambiguous ones.
void ambiguous_states(const pagan_msa *m, int id, int pos, std::vector<int> *out) {
    const SeqGraph &g = m->graph[id]->g;
    if (!g.ambiguous[pos]) { out->push_back(g.state[pos]); return; }
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    if (t.left < 0) return;                                        // an ambiguous leaf site has nothing below it
    if (g.child_l[pos] >= 0) ambiguous_states(m, m->id_of_tree[t.left], g.child_l[pos], out);
    if (g.child_r[pos] >= 0) ambiguous_states(m, m->id_of_tree[t.right], g.child_r[pos], out);
}

// Node::set_ambiguous_state (node.cpp:1661-1690): pushes a resolved state down the ambiguous sites that carry it.
bool set_ambiguous_state(pagan_msa *m, int id, int pos, int state) {
    SeqGraph &g = m->graph[id]->g;
    if (!g.ambiguous[pos]) return g.state[pos] == state;
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    if (t.left < 0) return false;
    bool cont = true;
    if (g.child_l[pos] >= 0 && set_ambiguous_state(m, m->id_of_tree[t.left], g.child_l[pos], state)) { g.state[pos] = state; cont = false; }
    if (g.child_r[pos] >= 0 && cont && set_ambiguous_state(m, m->id_of_tree[t.right], g.child_r[pos], state)) g.state[pos] = state;
    return false;
}

// Node::fix_ambiguous_states (node.cpp:1610-1636), --mostcommon only: a site whose two subtrees share exactly one state
// (and bring more than two states together) takes that state, and so do the ambiguous sites below it.
void fix_ambiguous_states(pagan_msa *m, int id) {
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    const int lid = m->id_of_tree[t.left], rid = m->id_of_tree[t.right];
    const int n = m->graph[id]->g.n_sites();
    std::vector<int> ls, rs;
    for (int j = 1; j < n - 1; ++j) {
        const SeqGraph &g = m->graph[id]->g;
        ls.clear(); rs.clear();
        if (g.child_l[j] >= 0) ambiguous_states(m, lid, g.child_l[j], &ls);
        if (g.child_r[j] >= 0) ambiguous_states(m, rid, g.child_r[j], &rs);
        std::sort(ls.begin(), ls.end()); ls.erase(std::unique(ls.begin(), ls.end()), ls.end());       // std::set semantics
        std::sort(rs.begin(), rs.end()); rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
        std::vector<int> both;
        std::set_intersection(ls.begin(), ls.end(), rs.begin(), rs.end(), std::back_inserter(both));
        if (both.size() == 1 && ls.size() + rs.size() > 2) set_ambiguous_state(m, id, j, both[0]);
    }
}

std::atomic<long long> parents_on_device{0};                     // (diagnostic: pagan_parents_device_calls)

// add_ancestral_sequence(va.get_simple_sequence()) (node.cpp:166): the parent graph from the path.
int build_parent(pagan_msa *m, int id, int unit_nodes) {
    NodeWork &w = m->work[id - m->n_leaves];
    if (!w.has_res || w.res.status != PAGAN_DP_REACHED) return PAGAN_E_INTERNAL;
    BuildSettings bs;
    if (m->opts.keep_all_edges) bs.reads_mode();
    if (m->opts.dp_flags & PAGAN_OPT_NO_REDUCED_TERMINAL_PEN) bs.reduced_terminal = false;
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    SeqGraph &gl = m->graph[m->id_of_tree[t.left]]->g, &gr = m->graph[m->id_of_tree[t.right]]->g;
    m->graph[id].reset(new pagan_hgraph());
    // The parent on the device the node was aligned on (dp_parent.hip; its copy there stays for the build one level up) when
    // the graphs are long enough for a dozen launches to beat one host thread's walk over the columns; PAGAN_PARENTS=host /
    // device forces either.  --mostcommon (round 5): the graph comes from the device all the same; fix_ambiguous_states then
    // rewrites states on the host -- of this node and of the ambiguous sites below it -- and the node's device copy, which the
    // build one level up reads its children's states from, takes the node's over (the graphs below have no device copy left).
    bool on_device = false;
    if (!m->backend && w.device >= 0) {
        const char *e = std::getenv("PAGAN_PARENTS");
        const bool force_host = e && std::strcmp(e, "host") == 0, force_dev = e && std::strcmp(e, "device") == 0;
        // (a wide level's parents are built side by side on the host's threads, and as many uploads of leaf-sized children
        //  at once cost more than they save: measured on 32 x 100 kb, 16 nodes 17 ms on the device against 7 on 16 threads,
        //  one node 3 ms against 8-15; so the device takes the levels of at most four long nodes)
        if (!force_host && (force_dev || (w.res.n_cols >= 20000 && unit_nodes <= 4))) {
            SeqGraph g;
            if (make_parent_device(gl, gr, w.res, (float)m->tree[t.left].dist, (float)m->tree[t.right].dist, m->state_table.data(),
                                   m->mf.S, m->mf.char_as, bs, w.device, &g)) {
                m->graph[id]->g = std::move(g);
                on_device = true;
                parents_on_device.fetch_add(1);
            }
        }
    }
    if (!on_device)
        m->graph[id]->g = make_parent(gl, gr, w.res, (float)m->tree[t.left].dist, (float)m->tree[t.right].dist,
                                      m->state_table.data(), m->mf.S, m->mf.char_as, bs);
    // (grandchildren's device copies are no longer needed: their parents are built)
    gl.dev.reset(); gr.dev.reset();
    if (m->opts.mostcommon) {
        fix_ambiguous_states(m, id);
        if (on_device && !parent_update_states(m->graph[id]->g)) m->graph[id]->g.dev.reset();
    }
    w.parent_pending = false;
    m->parents_built.fetch_add(1);
    return PAGAN_OK;
}

// Sites of a node's graph -- of the graph itself, or, for an imported node whose parent graph has not been built yet, what
// it will have: one site per alignment column plus the start and the end site (create_ancestral_sequence,
// basic_alignment.cpp:61-179: skipped and later deleted columns stay as sites).
int sites_of(const pagan_msa *m, int id) {
    if (m->graph[id]) return m->graph[id]->g.n_sites();
    return m->work[id - m->n_leaves].res.n_cols + 2;
}

// The graph of node `id`, built now if it is an imported node's that nobody has needed so far -- together with whatever is
// pending below it, children first (iteratively: a caterpillar tree is as deep as it has leaves).  Rank mode (one process
// per GPU): a rank builds the parents of the nodes it aligns and of the imported nodes BELOW the nodes it claims; the rest
// only if it is asked for the rows (node.cpp:196-223, 273-345: a thread builds the ancestor of the node it aligned).
// Ready nodes have disjoint subtrees, so the units of a round may call this side by side.
int ensure_graph(pagan_msa *m, int id) {
    if (m->graph[id]) return PAGAN_OK;
    std::vector<int> stack{id};
    while (!stack.empty()) {
        const int cur = stack.back();
        if (m->graph[cur]) { stack.pop_back(); continue; }
        if (cur < m->n_leaves || !m->done[cur]) return PAGAN_E_INTERNAL;       // (a leaf always has its graph; a node that is not done has no path)
        const TreeNode &t = m->tree[m->tree_of_id[cur]];
        const int lid = m->id_of_tree[t.left], rid = m->id_of_tree[t.right];
        if (!m->graph[lid] || !m->graph[rid]) {
            if (!m->graph[lid]) stack.push_back(lid);
            if (!m->graph[rid]) stack.push_back(rid);
            continue;
        }
        NodeWork &w = m->work[cur - m->n_leaves];
        // what pagan_msa_import_result could not check without the child graphs: a column names a child site (1 .. n_sites - 2)
        // or none (-1) -- which of the two follows from its path state --, a used edge is an edge of the child
        const SeqGraph &gl = m->graph[lid]->g, &gr = m->graph[rid]->g;
        const pagan_result &r = w.res;
        bool good = w.has_res;
        for (int k = 0; good && k < r.n_cols; ++k) {
            const pagan_col &c = r.cols[k];
            const bool hl = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_XGAPPED || c.path_state == PAGAN_XSKIPPED;
            const bool hr = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_YGAPPED || c.path_state == PAGAN_YSKIPPED;
            if (hl ? (c.left < 1 || c.left > gl.n_sites() - 2) : c.left != -1) good = false;
            if (hr ? (c.right < 1 || c.right > gr.n_sites() - 2) : c.right != -1) good = false;
        }
        for (int k = 0; good && k < r.n_left_used; ++k) if (r.left_used[k] < 0 || r.left_used[k] >= gl.n_edges()) good = false;
        for (int k = 0; good && k < r.n_right_used; ++k) if (r.right_used[k] < 0 || r.right_used[k] >= gr.n_edges()) good = false;
        if (!good) return PAGAN_E_ARG;
        const double t0 = now_s();
        const int rc = build_parent(m, cur, 1 << 20);                          // (imported: w.device is -1, the host builder)
        {
            std::lock_guard<std::mutex> g(m->mu);
            m->tm.build_s += now_s() - t0;
        }
        if (rc != PAGAN_OK) return rc;
        stack.pop_back();
    }
    return PAGAN_OK;
}

// One unit of the work queue: the nodes `ids` (all ready) prepared, aligned on device `dev`, their parents
// built.  Runs on the calling thread (+ the host thread pool); several of these run side by side on
// different devices.  "anchored alignment failed: trying again" (viterbi_alignment.cpp:298-317): a node
// whose end corner is unreachable inside its tunnel is re-aligned over the full matrix.
int run_unit(pagan_msa *m, const std::vector<int> &ids, int dev, int round, int threads) {
    const int n = m->n_leaves;
    double t0 = now_s();
    parallel_for((int)ids.size(), threads, [&](int r) { set_anchor_device(dev); prepare_node(m, ids[r], round); });
    if (const int e = m->lazy_err.load()) return e;
    const double t_prep = now_s() - t0;
    t0 = now_s();
    std::vector<int> ks(ids.size());
    std::vector<int64_t> cost(ids.size());
    if (m->opts.force_gap) {
        // the memory guard of align_sequences_this_node (node.cpp:81-152): while the alignment does not fit the budget,
        // the largest empty tunnel block is replaced by a gap-shaped tunnel; no block left -> the node fails
        int64_t budget = 0;
        int rc = device_budget(m, dev, &budget);
        if (rc != PAGAN_OK) return rc;
        for (int id : ids) {
            NodeWork &w = m->work[id - n];
            if (!w.banded) continue;
            while (pagan_dp_predict_bytes(w.gl.n_sites, w.gr.n_sites, &w.pb) > budget) {
                if (!force_gap(&w.upper, &w.lower, &w.blocks, m->opts.force_gap_threshold, m->opts.anchors_offset, m->opts.force_gap_wide != 0))
                    return PAGAN_E_MEMCAP;
                ++w.n_forced;
            }
        }
    }
    for (size_t r = 0; r < ids.size(); ++r) {
        NodeWork &w = m->work[ids[r] - n];
        cost[r] = pagan_dp_count_cells(w.gl.n_sites, w.gr.n_sites, w.banded ? &w.pb : nullptr);
        if (cost[r] < 0) return (int)cost[r];
        w.device = dev;
    }
    std::vector<int> order(ids.size());
    for (size_t r = 0; r < ids.size(); ++r) order[r] = (int)r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    for (size_t r = 0; r < ids.size(); ++r) ks[r] = ids[order[r]] - n;
    double fill_ms = 0, trace_ms = 0;
    int rc = align_on_device(m, ks, dev, &fill_ms, &trace_ms);
    if (rc != PAGAN_OK) return rc;
    std::vector<int> retry;
    for (int k : ks) if (m->work[k].banded && m->work[k].res.status == PAGAN_DP_UNREACHABLE) retry.push_back(k);
    if (!retry.empty()) {
        for (int k : retry) { m->work[k].banded = false; pagan_result_free(&m->work[k].res); m->work[k].has_res = false; }
        rc = align_on_device(m, retry, dev, &fill_ms, &trace_ms);
        if (rc != PAGAN_OK) return rc;
    }
    const double t_dp = now_s() - t0;
    t0 = now_s();
    std::atomic<int> bad(0);
    parallel_for((int)ids.size(), threads, [&](int r) { if (build_parent(m, ids[r], (int)ids.size()) != PAGAN_OK) bad = 1; });
    const double t_build = now_s() - t0;
    if (std::getenv("PAGAN_DP_VERBOSE"))
        std::fprintf(stderr, "pagan_msa: unit of %d node(s) on device %d: prepare (model, children, anchors, tunnel) %.1f ms, DP (plan, batch, kernels, fetch) %.1f ms, parents %.1f ms\n",
                     (int)ids.size(), dev, 1e3 * t_prep, 1e3 * t_dp, 1e3 * t_build);
    {
        std::lock_guard<std::mutex> g(m->mu);
        m->tm.anchors_s += t_prep; m->tm.dp_wall_s += t_dp; m->tm.build_s += t_build;
        m->tm.dp_fill_dev_s += fill_ms / 1e3; m->tm.dp_trace_dev_s += trace_ms / 1e3;
    }
    return bad ? PAGAN_E_INTERNAL : PAGAN_OK;     // unreachable end corner: caller may retry with use_anchors = 0
}

int64_t node_cost_estimate(const pagan_msa *m, int id) {
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    const int64_t lx = sites_of(m, m->id_of_tree[t.left]), ly = sites_of(m, m->id_of_tree[t.right]);
    // before the anchors are known: a tunnel is ~ (2 * offset + a few) cells per row, a full matrix Lx * Ly
    return m->opts.use_anchors ? (lx + ly) * (int64_t)(2 * m->opts.anchors_offset + 16) : lx * ly;
}

} // namespace

extern "C" {

int pagan_msa_ready(const pagan_msa *m, int32_t *ids, int32_t cap) {
    if (!m) return PAGAN_E_ARG;
    int cnt = 0;
    for (int id = m->n_leaves; id < 2 * m->n_leaves - 1; ++id)
        if (node_ready(m, id)) { if (ids && cnt < cap) ids[cnt] = id; ++cnt; }
    return cnt;
}

int pagan_msa_remaining(const pagan_msa *m) { return m ? m->remaining : PAGAN_E_ARG; }

int64_t pagan_msa_node_cost(const pagan_msa *m, int32_t id) {
    if (!m || id < m->n_leaves || id >= 2 * m->n_leaves - 1 || !node_ready(m, id)) return PAGAN_E_ARG;
    return node_cost_estimate(m, id);
}

// The in-process work queue (Node::start_threaded_alignment, node.cpp:196-223,289-345, with devices in the
// place of threads): whenever devices are idle and nodes are ready, the ready nodes are dealt over the idle
// devices (largest first, least-loaded device) and each device's share becomes one unit on its own feeder
// thread.  A parent becomes ready the moment its two children are done, whichever devices they ran on; a
// device that finishes early picks up what is ready without waiting for the others.
int pagan_msa_align_nodes(pagan_msa *m, int32_t n_ids, const int32_t *ids) {
    if (!m || m->aligned || n_ids < 0 || (n_ids > 0 && !ids)) return PAGAN_E_ARG;
    for (int k = 0; k < n_ids; ++k)
        if (ids[k] < m->n_leaves || ids[k] >= 2 * m->n_leaves - 1 || !node_ready(m, ids[k])) return PAGAN_E_ARG;
    if (n_ids == 0) return PAGAN_OK;
    const double t_start = now_s();
    int ndev = m->opts.n_devices;
    int first_dev = m->opts.first_device;
    if (ndev <= 0) { ndev = 1; if (!m->backend && hipGetDevice(&first_dev) != hipSuccess) return PAGAN_E_NODEVICE; }
    const int threads = host_threads_of(m);
    std::vector<int64_t> cost(n_ids);
    for (int k = 0; k < n_ids; ++k) cost[k] = node_cost_estimate(m, ids[k]);
    std::vector<int32_t> owner(n_ids);
    const int workers = std::min(ndev, (int)n_ids);
    pagan_assign_units(n_ids, cost.data(), workers, owner.data());
    std::vector<std::vector<int>> share(workers);
    for (int k = 0; k < n_ids; ++k) share[owner[k]].push_back(ids[k]);
    std::vector<int> rcs(workers, PAGAN_OK);
    const int round = m->rounds++;
    if (workers == 1) {
        rcs[0] = run_unit(m, share[0], first_dev, round, threads);
    } else {
        std::vector<std::thread> feeders;
        const int per = std::max(1, threads / workers);
        for (int d = 0; d < workers; ++d)
            feeders.emplace_back([&, d] { rcs[d] = run_unit(m, share[d], first_dev + d, round, per); });
        for (auto &th : feeders) th.join();
    }
    for (int rc : rcs) if (rc != PAGAN_OK) return rc;
    for (int k = 0; k < n_ids; ++k) { m->done[ids[k]] = 1; --m->remaining; }
    m->tm.total_s += now_s() - t_start;
    return PAGAN_OK;
}

// The rows of the alignment need every node's graph (an ancestor's row is read off its sites; the leaves' columns follow
// the child maps down from the root): parents still pending (imported nodes, rank mode) are built first.
static int ensure_rows(pagan_msa *m) {
    std::lock_guard<std::mutex> g(m->lazy_mu);
    if (m->rows_built) return PAGAN_OK;
    const double t0 = now_s();
    const int rc = ensure_graph(m, m->id_of_tree[m->root]);
    if (rc != PAGAN_OK) return rc;
    const double t1 = now_s();
    build_rows(m);
    m->tm.total_s += now_s() - t0;
    if (std::getenv("PAGAN_DP_VERBOSE"))
        std::fprintf(stderr, "pagan_msa: finish: pending parents %.1f ms, rows %.1f ms\n", 1e3 * (t1 - t0), 1e3 * (now_s() - t1));
    m->rows_built = true;
    return PAGAN_OK;
}

int pagan_msa_finish(pagan_msa *m) {
    if (!m || m->aligned || m->remaining != 0) return PAGAN_E_ARG;
    m->aligned = true;
    return ensure_rows(m);
}

// Rank mode: the walk is over, but this process may never be asked for the rows -- the graphs of the nodes other ranks
// aligned above its own are then never built here.  The first call that needs them (alignment length / rows / FASTA, a
// node's graph) builds what is pending.
int pagan_msa_finish_lazy(pagan_msa *m) {
    if (!m || m->aligned || m->remaining != 0) return PAGAN_E_ARG;
    m->aligned = true;
    return PAGAN_OK;
}

int pagan_msa_parents_built(const pagan_msa *m) { return m ? m->parents_built.load() : PAGAN_E_ARG; }

int pagan_msa_align(pagan_msa *m) {
    if (!m || m->aligned) return PAGAN_E_ARG;
    int ndev = m->opts.n_devices;
    int first_dev = m->opts.first_device;
    if (ndev <= 0) { ndev = 1; if (!m->backend && hipGetDevice(&first_dev) != hipSuccess) return PAGAN_E_NODEVICE; }
    const int threads = host_threads_of(m);
    if (ndev == 1) {
        // one device: every round takes all ready nodes as one batch (the guide tree's levels)
        std::vector<int32_t> ids(m->n_leaves);
        while (m->remaining > 0) {
            const int cnt = pagan_msa_ready(m, ids.data(), (int32_t)ids.size());
            if (cnt <= 0) return PAGAN_E_INTERNAL;
            const int rc = pagan_msa_align_nodes(m, cnt, ids.data());
            if (rc != PAGAN_OK) return rc;
        }
        return pagan_msa_finish(m);
    }
    // several devices: dynamic ready queue
    const double t_start = now_s();
    std::mutex qm;
    std::condition_variable cv;
    std::vector<char> queued(m->done.size(), 0), busy(ndev, 0);
    std::vector<std::thread> running(ndev);
    std::vector<int> finished;           // devices whose unit has ended and whose thread can be joined
    int in_flight = 0, err = PAGAN_OK;
    const int per = std::max(1, threads / ndev);
    std::unique_lock<std::mutex> lk(qm);
    for (;;) {
        for (int d : finished) { running[d].join(); busy[d] = 0; }
        finished.clear();
        if (err != PAGAN_OK || (m->remaining == 0 && in_flight == 0)) break;
        std::vector<int> ready, idle;
        for (int id = m->n_leaves; id < 2 * m->n_leaves - 1; ++id) if (!queued[id] && node_ready(m, id)) ready.push_back(id);
        for (int d = 0; d < ndev; ++d) if (!busy[d]) idle.push_back(d);
        if (!ready.empty() && !idle.empty()) {
            std::vector<int64_t> cost(ready.size());
            for (size_t k = 0; k < ready.size(); ++k) cost[k] = node_cost_estimate(m, ready[k]);
            const int workers = (int)std::min(idle.size(), ready.size());
            std::vector<int32_t> owner(ready.size());
            pagan_assign_units((int32_t)ready.size(), cost.data(), workers, owner.data());
            const int round = m->rounds++;
            for (int w = 0; w < workers; ++w) {
                std::vector<int> mine;
                for (size_t k = 0; k < ready.size(); ++k) if (owner[k] == w) { mine.push_back(ready[k]); queued[ready[k]] = 1; }
                const int d = idle[w];
                busy[d] = 1; ++in_flight;
                running[d] = std::thread([&, d, mine, round] {
                    const int rc = run_unit(m, mine, first_dev + d, round, per);
                    std::lock_guard<std::mutex> g(qm);
                    if (rc != PAGAN_OK && err == PAGAN_OK) err = rc;
                    if (rc == PAGAN_OK) for (int id : mine) { m->done[id] = 1; --m->remaining; }
                    --in_flight; finished.push_back(d);
                    cv.notify_all();
                });
            }
            continue;
        }
        if (in_flight == 0) { err = PAGAN_E_INTERNAL; break; }
        cv.wait(lk);
    }
    while (in_flight > 0) cv.wait(lk);
    for (int d : finished) running[d].join();
    lk.unlock();
    if (err != PAGAN_OK) return err;
    m->tm.total_s += now_s() - t_start;
    return pagan_msa_finish(m);
}

// ---- results across processes (one rank per GPU): what a node's alignment left behind, as bytes ------------
// Layout: int32 magic, node, status, end_matrix, end_x, end_y, end_x_edge, end_y_edge, n_cols, n_left_used,
// n_right_used, packed; int64 cells; double score; then the columns -- packed: one byte of path_state per column
// (the child indices are running counters, basic_alignment.cpp:73-165), else int32 triples -- and the used edge ids.
static const int32_t kResultMagic = 0x50475232;   // "PGR2"

int64_t pagan_msa_export_result(const pagan_msa *m, int32_t id, void *buf, int64_t cap) {
    if (!m || id < m->n_leaves || id >= 2 * m->n_leaves - 1) return PAGAN_E_ARG;
    const NodeWork &w = m->work[id - m->n_leaves];
    if (!w.has_res) return PAGAN_E_ARG;
    const pagan_result &r = w.res;
    bool packed = true;
    {
        int l = 1, rr = 1;
        for (int k = 0; k < r.n_cols && packed; ++k) {
            const pagan_col &c = r.cols[k];
            const bool hl = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_XGAPPED || c.path_state == PAGAN_XSKIPPED;
            const bool hr = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_YGAPPED || c.path_state == PAGAN_YSKIPPED;
            if ((hl ? c.left != l : c.left != -1) || (hr ? c.right != rr : c.right != -1)) packed = false;
            l += hl; rr += hr;
        }
    }
    const int64_t need = 12 * 4 + 8 + 8 + (packed ? (int64_t)r.n_cols : 12 * (int64_t)r.n_cols) + 4 * ((int64_t)r.n_left_used + r.n_right_used);
    if (!buf || cap < need) return need;
    char *p = (char *)buf;
    const int32_t head[12] = {kResultMagic, id, r.status, r.end_matrix, r.end_x, r.end_y, r.end_x_edge, r.end_y_edge,
                              r.n_cols, r.n_left_used, r.n_right_used, packed ? 1 : 0};
    std::memcpy(p, head, sizeof(head)); p += sizeof(head);
    std::memcpy(p, &r.cells, 8); p += 8;
    std::memcpy(p, &r.score, 8); p += 8;
    if (packed) for (int k = 0; k < r.n_cols; ++k) *p++ = (char)r.cols[k].path_state;
    else { std::memcpy(p, r.cols, 12 * (size_t)r.n_cols); p += 12 * (size_t)r.n_cols; }
    if (r.n_left_used) std::memcpy(p, r.left_used, 4 * (size_t)r.n_left_used);
    p += 4 * (size_t)r.n_left_used;
    if (r.n_right_used) std::memcpy(p, r.right_used, 4 * (size_t)r.n_right_used);
    return need;
}

// Takes over a node aligned by another rank: stores the result and marks the node done.  The parent graph is NOT built here
// (round 5; before, every rank built every parent, on the one thread that drains the posting log): ensure_graph builds it
// when this process claims a node above it or is asked for the rows.
int pagan_msa_import_result(pagan_msa *m, const void *buf, int64_t bytes) {
    if (!m || !buf || bytes < 12 * 4 + 16) return PAGAN_E_ARG;
    const char *p = (const char *)buf;
    int32_t head[12];
    std::memcpy(head, p, sizeof(head)); p += sizeof(head);
    if (head[0] != kResultMagic) return PAGAN_E_ARG;
    const int id = head[1];
    if (id < m->n_leaves || id >= 2 * m->n_leaves - 1 || !node_ready(m, id)) return PAGAN_E_ARG;
    const int n_cols = head[8], nl = head[9], nr = head[10], packed = head[11];
    if (n_cols < 0 || nl < 0 || nr < 0) return PAGAN_E_ARG;
    const int64_t need = 12 * 4 + 16 + (packed ? (int64_t)n_cols : 12 * (int64_t)n_cols) + 4 * ((int64_t)nl + nr);
    if (bytes < need) return PAGAN_E_ARG;
    NodeWork &w = m->work[id - m->n_leaves];
    if (w.has_res) { pagan_result_free(&w.res); w.has_res = false; }
    pagan_result &r = w.res;
    std::memset(&r, 0, sizeof(r));
    r.status = head[2]; r.end_matrix = head[3]; r.end_x = head[4]; r.end_y = head[5]; r.end_x_edge = head[6]; r.end_y_edge = head[7];
    r.n_cols = n_cols; r.n_left_used = nl; r.n_right_used = nr;
    std::memcpy(&r.cells, p, 8); p += 8;
    std::memcpy(&r.score, p, 8); p += 8;
    // pagan_result_free releases these with free()
    r.cols = (pagan_col *)std::malloc(sizeof(pagan_col) * (size_t)std::max(n_cols, 1));
    r.left_used = (int32_t *)std::malloc(4 * (size_t)std::max(nl, 1));
    r.right_used = (int32_t *)std::malloc(4 * (size_t)std::max(nr, 1));
    if (!r.cols || !r.left_used || !r.right_used) { std::free(r.cols); std::free(r.left_used); std::free(r.right_used); return PAGAN_E_NOMEM; }
    if (packed) {
        int l = 1, rr = 1;
        for (int k = 0; k < n_cols; ++k) {
            const int ps = (unsigned char)*p++;
            const bool hl = ps == PAGAN_MATCHED || ps == PAGAN_XGAPPED || ps == PAGAN_XSKIPPED;
            const bool hr = ps == PAGAN_MATCHED || ps == PAGAN_YGAPPED || ps == PAGAN_YSKIPPED;
            r.cols[k].path_state = ps; r.cols[k].left = hl ? l++ : -1; r.cols[k].right = hr ? rr++ : -1;
        }
    } else { std::memcpy(r.cols, p, 12 * (size_t)n_cols); p += 12 * (size_t)n_cols; }
    if (nl) std::memcpy(r.left_used, p, 4 * (size_t)nl);
    p += 4 * (size_t)nl;
    if (nr) std::memcpy(r.right_used, p, 4 * (size_t)nr);
    // What can be checked without the child graphs is checked now: the path states, and that a column names a child site
    // exactly where its state says it has one.  The rest -- the sites against the children's sizes, the used edges against
    // their edge lists -- when the parent graph is built (ensure_graph), which happens when this process needs it: when it
    // claims a node above, or is asked for the rows.  Nothing of the payload is used as an index before that.
    const TreeNode &t = m->tree[m->tree_of_id[id]];
    bool good = r.status == PAGAN_DP_REACHED || r.status == PAGAN_DP_UNREACHABLE;
    for (int k = 0; good && k < n_cols; ++k) {
        const pagan_col &c = r.cols[k];
        if (c.path_state < PAGAN_MATCHED || c.path_state > PAGAN_YSKIPPED) { good = false; break; }
        const bool hl = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_XGAPPED || c.path_state == PAGAN_XSKIPPED;
        const bool hr = c.path_state == PAGAN_MATCHED || c.path_state == PAGAN_YGAPPED || c.path_state == PAGAN_YSKIPPED;
        if (hl ? c.left < 1 : c.left != -1) good = false;
        if (hr ? c.right < 1 : c.right != -1) good = false;
    }
    if (!good) { pagan_result_free(&r); std::memset(&r, 0, sizeof(r)); return PAGAN_E_ARG; }
    w.has_res = true; w.has_job = false; w.device = -1; w.node = id; w.level = m->rounds;
    w.imp_l = sites_of(m, m->id_of_tree[t.left]); w.imp_r = sites_of(m, m->id_of_tree[t.right]);   // (no job was prepared here: node_info reports the children's sizes from these)
    if (r.status != PAGAN_DP_REACHED) return PAGAN_E_INTERNAL;       // (the owner retries an unreachable corner itself: it never posts one)
    w.parent_pending = true;
    m->done[id] = 1; --m->remaining;
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
    o->level = w.level; o->n_hits = w.n_hits; o->n_forced_gaps = w.n_forced;
    o->dist = m->tree[t.left].dist + m->tree[t.right].dist;
    if (w.has_res) {
        o->left_sites = w.has_job ? w.gl.n_sites : w.imp_l; o->right_sites = w.has_job ? w.gr.n_sites : w.imp_r;
        o->cells = w.res.cells; o->score = w.res.score; o->status = w.res.status;
    }
    if (m->graph[id]) o->sites = m->graph[id]->g.n_sites();
    return PAGAN_OK;
}

int pagan_msa_node_job(const pagan_msa *m, int32_t k, pagan_job *o) {
    if (!m || !o || k < 0 || k >= m->n_leaves - 1 || !m->work[k].has_res || !m->work[k].has_job) return PAGAN_E_ARG;
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

int pagan_msa_alignment_length(const pagan_msa *m) {
    if (!m || !m->aligned) return PAGAN_E_ARG;
    if (const int rc = ensure_rows(const_cast<pagan_msa *>(m))) return rc;       // (pagan_msa_finish_lazy: built on first use)
    return (int)m->rows[0].size();
}

int pagan_msa_alignment_row(const pagan_msa *m, int32_t leaf, char *buf) {
    if (!m || !m->aligned || !buf) return PAGAN_E_ARG;
    if (const int rc = ensure_rows(const_cast<pagan_msa *>(m))) return rc;
    if (leaf < 0 || leaf >= (int)m->rows.size()) return PAGAN_E_ARG;
    std::memcpy(buf, m->rows[leaf].c_str(), m->rows[leaf].size() + 1);
    return PAGAN_OK;
}

// Fasta_reader::write_fasta (src/utils/fasta_reader.cpp:596-629) over Node::get_alignment's leaf rows
// (src/main/node.cpp:537-575): one entry per leaf in guide-tree order (left to right, as get_leaf_nodes
// collects them), `>name`, the aligned row cut into lines of chars_by_line characters (60 by default).
int pagan_msa_write_fasta(const pagan_msa *m, const char *path, int32_t chars_by_line) {
    return pagan_msa_write_fasta_nodes(m, path, chars_by_line, 0);
}

// ... with include_internal (--output-ancestors): every node in Node::get_all_nodes order -- left subtree, the node,
// right subtree (node.h:277-290) -- internal nodes named #k# in alignment order (node.h:479-495).
int pagan_msa_write_fasta_nodes(const pagan_msa *m, const char *path, int32_t chars_by_line, int32_t include_internal) {
    if (!m || !m->aligned || !path) return PAGAN_E_ARG;
    if (const int rc = ensure_rows(const_cast<pagan_msa *>(m))) return rc;
    const size_t width = chars_by_line > 0 ? (size_t)chars_by_line : 60;
    std::FILE *f = std::fopen(path, "w");
    if (!f) return PAGAN_E_ARG;
    std::vector<int> order;
    {
        // in-order walk without recursion: (tree index, children done?)
        std::vector<std::pair<int, bool>> stack{{m->root, false}};
        while (!stack.empty()) {
            const auto [t, visited] = stack.back();
            stack.pop_back();
            const TreeNode &n = m->tree[t];
            if (n.left < 0) { order.push_back(m->id_of_tree[t]); continue; }
            if (visited) { if (include_internal) order.push_back(m->id_of_tree[t]); continue; }
            stack.push_back({n.right, false});
            stack.push_back({t, true});
            stack.push_back({n.left, false});
        }
    }
    for (int leaf : order) {
        if (leaf < m->n_leaves) std::fprintf(f, ">%s\n", m->names[leaf].c_str());
        else std::fprintf(f, ">#%d#\n", leaf - m->n_leaves + 1);
        const std::string &row = m->rows[leaf];
        for (size_t at = 0; at < row.size(); at += width) std::fprintf(f, "%.*s\n", (int)std::min(width, row.size() - at), row.c_str() + at);
    }
    return std::fclose(f) == 0 ? PAGAN_OK : PAGAN_E_ARG;
}

void *pagan_msa_node_graph(const pagan_msa *m, int32_t node) {
    if (!m || node < 0 || node >= (int)m->graph.size()) return nullptr;
    if (!m->graph[node] && m->done[node]) {                        // (an imported node's graph: built when somebody asks for it)
        pagan_msa *mm = const_cast<pagan_msa *>(m);
        std::lock_guard<std::mutex> g(mm->lazy_mu);
        if (ensure_graph(mm, node) != PAGAN_OK) return nullptr;
    }
    return m->graph[node].get();
}

void pagan_msa_destroy(pagan_msa *m) { delete m; }

// ---- host graphs on their own -------------------------------------------------------------
pagan_hgraph *pagan_hgraph_leaf(const char *residues, const char *alphabet, int32_t flags) {
    pagan_hgraph *h = new pagan_hgraph();
    h->g = make_leaf(residues, alphabet, flags);
    return h;
}

pagan_hgraph *pagan_hgraph_leaf_codon(const char *nucleotides) {
    if (!nucleotides) return nullptr;
    pagan_hgraph *h = new pagan_hgraph();
    std::string symbols;
    const std::vector<int32_t> states = ModelFactory::codon_states(nucleotides, &symbols);
    h->g = make_leaf_states(states, std::move(symbols), 3, 0);
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

// The same graph built on the current device (dp_parent.hip).  NULL without a device or on a HIP error -- never a host-built
// graph in its place.  info (may be NULL): runs of skipped sites, fixpoint rounds of the boundary pass, deleted sites, edge
// weights outside the log table, log weights the host had to put right.
pagan_hgraph *pagan_hgraph_parent_device(pagan_hgraph *l, pagan_hgraph *r, const pagan_result *res, float lbl, float rbl,
                                         const int32_t *parsimony, int32_t S, int32_t char_as, int32_t flags, int32_t *info) {
    BuildSettings bs;
    if (flags & 1) bs.reads_mode();
    if (flags & 2) bs.reduced_terminal = false;
    pagan_hgraph *h = new pagan_hgraph();
    ParentBuildInfo pi;
    if (!make_parent_device(l->g, r->g, *res, lbl, rbl, parsimony, S, char_as, bs, -1, &h->g, &pi)) { delete h; return nullptr; }
    if (info) { info[0] = pi.runs; info[1] = pi.rounds; info[2] = pi.deleted_sites; info[3] = pi.weights_outside_table; info[4] = pi.log_weights_patched; }
    return h;
}
long long pagan_parents_device_calls(void) { return parents_on_device.load(); }

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

int pagan_prefix_hits(const char *s1, const char *s2, int32_t min_length, int32_t *hits, int32_t cap) {
    if (!s1 || !s2) return PAGAN_E_ARG;
    std::vector<Hit> v;
    prefix_hits(s1, s2, min_length, &v);
    for (size_t k = 0; k < v.size() && (int)k < cap; ++k) { hits[4 * k] = v[k].s1; hits[4 * k + 1] = v[k].s2; hits[4 * k + 2] = v[k].len; hits[4 * k + 3] = v[k].score; }
    return (int)v.size();
}

long long pagan_anchors_device_calls(void) { return device_finder_calls.load(); }

int pagan_drop_bad_hits(int32_t *hits, int32_t n, int32_t thr_total, int32_t thr_partly) {
    if (n < 0 || (n > 0 && !hits)) return PAGAN_E_ARG;
    std::vector<Hit> v(n);
    for (int k = 0; k < n; ++k) v[k] = Hit{hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]};
    drop_bad_hits(&v, (unsigned)thr_total, (unsigned)thr_partly);
    for (size_t k = 0; k < v.size(); ++k) { hits[4 * k] = v[k].s1; hits[4 * k + 1] = v[k].s2; hits[4 * k + 2] = v[k].len; hits[4 * k + 3] = v[k].score; }
    return (int)v.size();
}

int pagan_define_tunnel_overlapping(const int32_t *hits, int32_t n, const char *g1, const char *g2, int32_t width,
                                    int32_t *upper, int32_t *lower, int32_t *blocks, int32_t cap) {
    if (n < 0 || (n > 0 && !hits) || !g1 || !g2 || !upper || !lower) return PAGAN_E_ARG;
    std::vector<Hit> v(n);
    for (int k = 0; k < n; ++k) v[k] = Hit{hits[4 * k], hits[4 * k + 1], hits[4 * k + 2], hits[4 * k + 3]};
    std::vector<int32_t> up, lo;
    std::vector<TunnelBlock> eb;
    hits_to_band_overlapping(v, g1, g2, width, &up, &lo, &eb);
    std::memcpy(upper, up.data(), sizeof(int32_t) * up.size());
    std::memcpy(lower, lo.data(), sizeof(int32_t) * lo.size());
    for (size_t k = 0; k < eb.size() && (int)k < cap; ++k) { blocks[4 * k] = eb[k].sx; blocks[4 * k + 1] = eb[k].sy; blocks[4 * k + 2] = eb[k].ex; blocks[4 * k + 3] = eb[k].ey; }
    return (int)eb.size();
}

int pagan_force_gap(int32_t *upper, int32_t *lower, int32_t n, const int32_t *blocks, int32_t n_blocks, int32_t threshold,
                    int32_t width, int32_t wide) {
    if (!upper || !lower || n < 1 || n_blocks < 0 || (n_blocks > 0 && !blocks)) return PAGAN_E_ARG;
    std::vector<int32_t> up(upper, upper + n), lo(lower, lower + n);
    std::vector<TunnelBlock> eb(n_blocks);
    for (int k = 0; k < n_blocks; ++k) { eb[k].sx = blocks[4 * k]; eb[k].sy = blocks[4 * k + 1]; eb[k].ex = blocks[4 * k + 2]; eb[k].ey = blocks[4 * k + 3]; }
    const bool done = force_gap(&up, &lo, &eb, threshold, width, wide != 0);
    std::memcpy(upper, up.data(), sizeof(int32_t) * n);
    std::memcpy(lower, lo.data(), sizeof(int32_t) * n);
    return done ? 1 : 0;
}

int pagan_dna_model(const float bf[4], double distance, float *table, float *params, int32_t *parsimony) {
    if (!bf || !table || !params) return PAGAN_E_ARG;
    ModelFactory mf;
    mf.init_dna(bf);
    const EvolModel em = mf.alignment_model(distance);
    std::memcpy(table, em.log_score.data(), sizeof(float) * 225);
    params[0] = em.log_gap_open; params[1] = em.log_gap_ext; params[2] = em.log_gap_end_ext; params[3] = em.log_non_gap;
    if (parsimony) std::memcpy(parsimony, mf.parsimony.data(), sizeof(int32_t) * 225);
    return PAGAN_OK;
}

int pagan_protein_model(double distance, float *table, float *params, int32_t *parsimony) {
    if (!table || !params) return PAGAN_E_ARG;
    static ModelFactory mf;                       // the WAG eigen solution does not depend on the input
    static std::once_flag once;
    std::call_once(once, [] { mf.init_protein(); });
    const EvolModel em = mf.alignment_model(distance);
    std::memcpy(table, em.log_score.data(), sizeof(float) * em.log_score.size());
    params[0] = em.log_gap_open; params[1] = em.log_gap_ext; params[2] = em.log_gap_end_ext; params[3] = em.log_non_gap;
    if (parsimony) std::memcpy(parsimony, mf.parsimony.data(), sizeof(int32_t) * mf.parsimony.size());
    return PAGAN_OK;
}

int pagan_codon_model(double distance, float *table, float *params, int32_t *parsimony) {
    if (!table || !params) return PAGAN_E_ARG;
    const ModelFactory &mf = codon_factory();
    const EvolModel em = mf.alignment_model(distance);
    std::memcpy(table, em.log_score.data(), sizeof(float) * em.log_score.size());
    params[0] = em.log_gap_open; params[1] = em.log_gap_ext; params[2] = em.log_gap_end_ext; params[3] = em.log_non_gap;
    if (parsimony) std::memcpy(parsimony, mf.parsimony.data(), sizeof(int32_t) * mf.parsimony.size());
    return PAGAN_OK;
}

int pagan_codon_alphabet(char *names, int32_t *mostcommon) {
    const ModelFactory &mf = codon_factory();
    if (names) std::memcpy(names, mf.codon_names.c_str(), mf.codon_names.size() + 1);
    if (mostcommon) std::memcpy(mostcommon, mf.mostcommon.data(), sizeof(int32_t) * mf.mostcommon.size());
    return mf.S;
}

int pagan_codon_translate(const char *codons, char *out) {
    if (!codons || !out) return PAGAN_E_ARG;
    const std::string p = ModelFactory::translate_codons(codons);
    std::memcpy(out, p.c_str(), p.size() + 1);
    return (int)p.size();
}

int pagan_codon_states(const char *nucleotides, int32_t *states) {
    if (!nucleotides || !states) return PAGAN_E_ARG;
    const std::vector<int32_t> st = ModelFactory::codon_states(nucleotides);
    if (!st.empty()) std::memcpy(states, st.data(), sizeof(int32_t) * st.size());
    return (int)st.size();
}

// Probability-space view of the same model (Evol_model::score / gap_open / gap_ext / non_gap): score [a + b*S],
// params[3] = gap_open, gap_ext, non_gap.  data_type 1: DNA (base_freq needed), 2: protein, 3: codon.
int pagan_model_prob_table(int32_t data_type, const float *base_freq, double distance, float *score, float *params) {
    if (!score || !params || (data_type != 2 && data_type != 3 && !base_freq)) return PAGAN_E_ARG;
    ModelFactory own;
    if (data_type == 2) own.init_protein(); else if (data_type != 3) own.init_dna(base_freq);
    const ModelFactory &mf = data_type == 3 ? codon_factory() : own;
    const EvolModel em = mf.alignment_model(distance);
    std::memcpy(score, em.score.data(), sizeof(float) * em.score.size());
    params[0] = em.gap_open; params[1] = em.gap_ext; params[2] = em.non_gap;
    return PAGAN_OK;
}

int pagan_model_alphabets(int32_t data_type, char *leaf_alphabet, char *ancestral_alphabet) {
    ModelFactory mf;
    if (data_type == 2) mf.init_protein();
    else { const float bf[4] = {0.25f, 0.25f, 0.25f, 0.25f}; mf.init_dna(bf); }
    if (leaf_alphabet) std::memcpy(leaf_alphabet, mf.leaf_alphabet.c_str(), mf.leaf_alphabet.size() + 1);
    if (ancestral_alphabet) std::memcpy(ancestral_alphabet, mf.ancestral_alphabet.c_str(), mf.ancestral_alphabet.size() + 1);
    return mf.S;
}

int pagan_eigen_qrev(const double *Q, const double *pi, int32_t n, double *root, double *U, double *V) {
    if (!Q || !pi || n < 1 || !root || !U || !V) return PAGAN_E_ARG;
    return eigen_qrev(Q, pi, n, root, U, V);
}

int pagan_msa_set_batch_backend(pagan_msa *m, pagan_batch_fn fn, void *user) {
    if (!m) return PAGAN_E_ARG;
    m->backend = fn; m->backend_user = user;
    return PAGAN_OK;
}

int pagan_msa_node_device(const pagan_msa *m, int32_t k) {
    if (!m || k < 0 || k >= m->n_leaves - 1) return PAGAN_E_ARG;
    return m->work[k].device;
}

int pagan_msa_data_type(const pagan_msa *m) { return m ? (m->mf.type == kCodon ? 3 : m->mf.type == kProtein ? 2 : 1) : PAGAN_E_ARG; }

} // extern "C"
