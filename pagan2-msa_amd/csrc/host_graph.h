// host_graph.h -- host-side sequence graphs of the guide-tree walk, stored CSR-first.
//
// The reference keeps a Sequence as vector<Site> + vector<Edge> with per-site linked edge
// lists (src/main/sequence.h:34-127,216-658).  Here a graph is a set of flat arrays whose
// bwd adjacency is exactly the pagan_graph the device consumes, so handing a node to the
// aligner is a pointer assignment, not a flatten pass.  List ORDER is preserved: bwd lists
// are in the order the reference appends/iterates them, because that order decides DP ties.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pagan_dp.h"

namespace pagan {

enum SiteType { kStartSite, kRealSite, kStopSite, kBreakStart, kBreakStop, kNonReal };   // sequence.h:226

struct SeqGraph {
    // ---- sites ----
    std::vector<int32_t> state;          // Site::character_state
    std::vector<int8_t>  site_type;      // SiteType
    std::vector<int8_t>  path_state;     // Site::Path_state (0 = ends_site, 1 = terminal, 2..6)
    std::vector<int32_t> child_l, child_r;
    std::vector<int32_t> count_since_used;   // Site::branch_count_since_last_used
    std::vector<float>   dist_since_used;    // Site::branch_distance_since_last_used
    std::vector<uint8_t> ambiguous;
    // ---- edges (index = Edge::index) ----
    std::vector<int32_t> e_start, e_end;
    std::vector<float>   e_w, e_logw;        // posterior_weight, log_posterior_weight (logf)
    std::vector<int32_t> e_count_since_used, e_count_as_skipped;
    std::vector<float>   e_dist_since_used;
    std::vector<uint8_t> e_used;             // set from the child's point of view by the parent's alignment
    // ---- adjacency, iteration order preserved ----
    std::vector<int32_t> bwd_off, bwd_eid;   // CSR over sites
    std::vector<int32_t> bwd_src;            // = e_start[bwd_eid]   (device input)
    std::vector<float>   bwd_logw;           // = e_logw[bwd_eid]    (device input)
    std::vector<int32_t> fwd_off, fwd_eid;
    bool terminal = false;                   // leaf (Sequence::is_terminal_sequence)
    std::string symbols;                     // leaf residues, sites 1..n-2 (sym_width characters each)
    int sym_width = 1;                       // characters a state prints as: 1, or 3 for codon graphs
    // the graph's copy on a device, where the parent-graph builder of dp_parent.hip reads a child and leaves a parent (opaque
    // here; released with the last copy of the graph).  The host arrays stay the authority for everything on the host.
    std::shared_ptr<void> dev;

    int n_sites() const { return (int)state.size(); }
    int n_edges() const { return (int)e_start.size(); }
    pagan_graph view() const {
        pagan_graph g;
        g.n_sites = n_sites(); g.n_edges = n_edges();
        g.state = state.data(); g.bwd_off = bwd_off.data(); g.bwd_src = bwd_src.data();
        g.bwd_logw = bwd_logw.data(); g.bwd_eid = bwd_eid.data();
        return g;
    }
};

struct BuildSettings {                       // basic_alignment.h:546-586,627-628
    float max_skip_distance = 0.5f;
    int   max_skip_branches = 10;
    int   max_match_skip_branches = 5;
    float branch_skip_probability = 0.9f;
    bool  reduced_terminal = true;
    void reads_mode() { max_skip_distance = 5; max_skip_branches = 50000; max_match_skip_branches = 50000; branch_skip_probability = 1; }
};

enum LeafFlags { kLeaf454 = 1, kLeafHomopolymer = 2 };

// Sequence::create_default_sequence (src/main/sequence.cpp:152-303)
SeqGraph make_leaf(const std::string &residues, const std::string &full_alphabet, int flags);
// the same from states already looked up; symbols holds sym_width characters per state.  Codon leaves
// (Sequence::create_codon_sequence, sequence.cpp:306-359) come this way with flags 0: a plain chain.
SeqGraph make_leaf_states(const std::vector<int32_t> &states, std::string symbols, int sym_width, int flags);

// Basic_alignment::build_ancestral_sequence (src/main/basic_alignment.cpp:36-59): path -> parent.
// Marks `left`/`right` edges used from the result first (the traceback's side effect,
// viterbi_alignment.cpp:1054-1057,1079-1101,1128,1155).
SeqGraph make_parent(SeqGraph &left, SeqGraph &right, const pagan_result &res, float left_branch,
                     float right_branch, const int32_t *parsimony, int n_states, int char_as,
                     const BuildSettings &bs);

// The same graph built on a device (dp_parent.hip): sites as a map over the path's columns, a site's
// edges by one thread, ids and lists by scans, the boundary pass as a fixpoint over runs of skipped sites.  Returns false
// (and leaves *out alone) without a device or on a HIP error: the caller then takes make_parent.  device < 0: the current one.
struct ParentBuildInfo { int runs = 0, rounds = 0, deleted_sites = 0, weights_outside_table = 0, log_weights_patched = 0; };
bool make_parent_device(SeqGraph &left, SeqGraph &right, const pagan_result &res, float left_branch, float right_branch,
                        const int32_t *parsimony, int n_states, int char_as, const BuildSettings &bs, int device, SeqGraph *out,
                        ParentBuildInfo *info = nullptr);
void parent_release_cache();
bool parent_update_states(SeqGraph &g);      // g's states changed on the host (--mostcommon): its device copy takes them over

// Sequence::get_sequence_string (src/main/sequence.cpp:704-740)
std::string sequence_string(const SeqGraph &g, bool with_gaps, const std::string &full_alphabet);

} // namespace pagan
