/*
 * pagan_host.h -- C ABI of the host-side guide-tree walk that drives pagan_dp.h.
 *
 * Mirrors the part of the reference's Node class that calls the aligner:
 * Node::start_alignment / start_openmp_alignment -> align_sequences_this_node
 * (src/main/node.h:880-938, src/main/node.cpp:52-285): post-order over a rooted binary guide
 * tree; at every internal node build the Evol_model for dist = d_left + d_right, define the
 * tunnel from anchors, call the pairwise aligner (GPU), build the parent Sequence graph.
 * Nodes whose children are finished are independent and are aligned as one batch
 * (node.cpp:227-285), optionally spread over several devices or ranks (no collectives in the
 * data path: what ranks exchange is the finished path of a node, host bytes).
 *
 * Also exposes the host graph builder on its own (pagan_hgraph_*), the counterpart of
 * Sequence construction (src/main/sequence.cpp:152-303) and
 * Basic_alignment::build_ancestral_sequence (src/main/basic_alignment.cpp:36-653).
 */
#ifndef PAGAN_HOST_H
#define PAGAN_HOST_H

#include <stdint.h>
#include "pagan_dp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PAGAN_E_TREE   -20   /* Newick parse error / tree not rooted binary / unknown leaf */
#define PAGAN_E_MEMCAP -21   /* one alignment would not fit the device memory budget         */

typedef struct pagan_msa pagan_msa;

typedef struct pagan_msa_opts {
    int32_t  use_anchors;        /* 0 = --no-anchors (full matrix), 1 = --use-prefix-anchors      */
    int32_t  anchors_offset;     /* --anchors-offset, default 15 (settings.cpp:157)              */
    int32_t  prefix_hit_length;  /* --prefix-hit-length, default 30 (settings.cpp:160)           */
    int32_t  hit_trim;           /* --exonerate-hit-trim, default 5 (settings.cpp:155)           */
    uint32_t dp_flags;           /* PAGAN_OPT_*                                                  */
    int32_t  leaf_flags;         /* 1 = --454, 2 = --homopolymer (sequence.cpp:205,253)          */
    int32_t  keep_all_edges;     /* --keep-all-edges (basic_alignment.h:572-586)                 */
    int32_t  n_devices;          /* devices to farm ready nodes over; 0 = just the current one   */
    int32_t  first_device;       /* ordinal of the first device used                             */
    int32_t  host_threads;       /* threads for anchors / graph building; 0 = hardware           */
    float    truncate_branches;  /* --truncate-branches, default 0.2 (settings.cpp:228)          */
    int64_t  device_mem_budget;  /* bytes of HBM one batch may use; 0 = 80% of free memory       */
    int32_t  data_type;          /* 0 = guess (Fasta_reader::check_sequence_data_type,
                                    fasta_reader.cpp:1303-1336), 1 = DNA, 2 = protein (WAG,
                                    211-letter alphabet, model_factory.cpp:304-632,1478-1595),
                                    3 = DNA read as codons (--codons: 1892 states, rows of three
                                    characters per column; anchors are found in the codon
                                    strings' translation, viterbi_alignment.cpp:54-60)            */
    int32_t  pileup_rates;       /* ins = del = 0.25: --454/--homopolymer with --pileup-alignment
                                    (model_factory.cpp:1901-1905)                                  */
    int32_t  anchor_mode;        /* 0 = order-conflict filter + Find_anchors::define_tunnel
                                    (viterbi_alignment.cpp:159-164); 1 = eliminate_bad_hits +
                                    define_tunnel_with_overlapping_hits (:148-157, the BLAST branch;
                                    the hits here are the prefix anchors)                          */
    int32_t  overlap_total;      /* --ncbi-threshold-overlap-total, default 50 (settings.cpp:180)  */
    int32_t  overlap_partly;     /* --ncbi-threshold-overlap-partly, default 400 (settings.cpp:181) */
    int32_t  force_gap;          /* --force-gap (node.cpp:124-141): over the memory budget, replace the
                                    largest empty tunnel block by a gap until the node fits         */
    int32_t  force_gap_threshold;/* --force-gap-threshold, default 40000 (settings.cpp:189)        */
    int32_t  force_gap_wide;     /* --force-gap-wide-tunnel                                        */
    int32_t  mostcommon;         /* --mostcommon: a matched column's parent state from the most-common table
                                    (basic_alignment.cpp:146-147) and Node::fix_ambiguous_states
                                    (node.cpp:1610-1690) after every node                            */
} pagan_msa_opts;

void pagan_msa_default_opts(pagan_msa_opts *o);

/* The work-queue rule that spreads a tree level's ready nodes (node.cpp:273-285 run_nodes) over
 * devices or ranks: largest cost first, each unit to the least-loaded worker.  owner[k] in
 * [0, n_workers).  Pure host function (no GPU).                                             */
void pagan_assign_units(int32_t n, const int64_t *cost, int32_t n_workers, int32_t *owner);

typedef struct pagan_node_info {
    int32_t node, left, right;   /* node ids: leaves 0..n-1 in input order, internal n..2n-2
                                    in alignment (post-order) order                              */
    int32_t level;               /* batch (tree level) the node was aligned in                   */
    int32_t left_sites, right_sites, sites;
    int32_t n_hits;
    int64_t cells;
    double  dist;                /* d_left + d_right after truncation (node.cpp:70)               */
    double  score;
    int32_t status;
    int32_t n_forced_gaps;       /* empty tunnel blocks replaced by a gap (--force-gap)             */
} pagan_node_info;

typedef struct pagan_msa_timing {
    double total_s, model_s, anchors_s, dp_wall_s, dp_fill_dev_s, dp_trace_dev_s, build_s;
} pagan_msa_timing;

int  pagan_msa_create(int32_t n_seqs, const char *const *names, const char *const *seqs,
                      const char *newick, const pagan_msa_opts *opts, pagan_msa **out);
/* Progressive alignment of every internal node; the DP runs on the GPU(s).  With n_devices > 1 the
 * ready nodes are a work queue over the devices (Node::start_threaded_alignment, node.cpp:196-223,
 * 289-345): a parent is ready the moment its two children are done, idle devices take what is
 * ready.  No collective: parents are built on the host.                                          */
int  pagan_msa_align(pagan_msa *m);

/* ---- the same walk, one step at a time (one process per GPU: tree replicated, DP sharded) -------
 * Every rank holds the whole tree.  Per round: pagan_msa_ready lists the nodes whose children are
 * done (build_queues, node.cpp:273-285; the same list on every rank), the ranks split it with
 * pagan_assign_units over pagan_msa_node_cost, each aligns its share (pagan_msa_align_nodes:
 * model, anchors, DP on this rank's device(s), parent graph), exports what the alignment left behind
 * (pagan_msa_export_result: path columns + used child edges + max_end, a few bytes per column) and
 * imports the other ranks' results (pagan_msa_import_result stores them; the parent graph of an
 * imported node is built when this process needs it -- when it claims a node above, or is asked for
 * the rows -- so the ranks shard the parent graphs as they shard the alignments: node.cpp:196-223,
 * 273-345, a thread builds the ancestor of the node it aligned).  pagan_msa_finish after the last
 * round builds what is pending and the rows; pagan_msa_finish_lazy leaves both to the first call
 * that needs them (a rank that is never asked for the alignment never builds them).
 * pagan_msa_align is this loop with one rank.                                                  */
int     pagan_msa_ready(const pagan_msa *m, int32_t *ids, int32_t cap);    /* count (may exceed cap) */
int     pagan_msa_remaining(const pagan_msa *m);                           /* internal nodes not yet done */
int64_t pagan_msa_node_cost(const pagan_msa *m, int32_t node);             /* cells estimate of a ready node */
int     pagan_msa_align_nodes(pagan_msa *m, int32_t n, const int32_t *nodes);
int64_t pagan_msa_export_result(const pagan_msa *m, int32_t node, void *buf, int64_t cap);  /* bytes needed/written */
int     pagan_msa_import_result(pagan_msa *m, const void *buf, int64_t bytes);
int     pagan_msa_finish(pagan_msa *m);
int     pagan_msa_finish_lazy(pagan_msa *m);
int     pagan_msa_parents_built(const pagan_msa *m);                       /* parent graphs this process has built so far */
int     pagan_msa_data_type(const pagan_msa *m);                           /* 1 DNA, 2 protein */
int     pagan_msa_node_device(const pagan_msa *m, int32_t k);              /* device internal node k ran on; -1: another rank */
/* TEST SEAM, never set by the product: replaces pagan_dp_align_batch as the thing the work queue hands its
 * batches to, so that the queue (dealing, feeder threads, retries, result exchange) can be exercised on a
 * machine without a GPU.  fn has pagan_dp_align_batch's contract (results released with free()); the device a
 * batch was dealt to arrives in opts->device.  NULL restores the HIP path.                                    */
typedef int (*pagan_batch_fn)(int32_t n, const pagan_job *jobs, const pagan_opts *opts, pagan_result *out, void *user);
int     pagan_msa_set_batch_backend(pagan_msa *m, pagan_batch_fn fn, void *user);
int  pagan_msa_n_internal(const pagan_msa *m);
int  pagan_msa_node_info(const pagan_msa *m, int32_t k, pagan_node_info *out);
/* Borrowed views (valid until pagan_msa_destroy) of what node k's alignment consumed and
 * produced: usable as a pagan_job for pagan_batch_create / the oracle.                   */
int  pagan_msa_node_job(const pagan_msa *m, int32_t k, pagan_job *out);
int  pagan_msa_node_result(const pagan_msa *m, int32_t k, pagan_result *out);
int  pagan_msa_timing_get(const pagan_msa *m, pagan_msa_timing *out);
int  pagan_msa_alignment_length(const pagan_msa *m);
/* Row of node `node` of the final alignment, '-' for gaps; buf >= length+1.  Leaves 0..n-1 in input order;
 * internal nodes n..2n-2 (the ancestors' rows of --output-ancestors: the state's character, a gap where the
 * site is skipped or deleted; get_alignment_column_at, node.cpp:779-834).                                  */
int  pagan_msa_alignment_row(const pagan_msa *m, int32_t node, char *buf);
/* The leaf rows as FASTA, leaves in guide-tree order, `>name` + the row cut into lines of chars_by_line
 * characters (<= 0: 60): Fasta_reader::write_fasta over Node::get_alignment
 * (src/utils/fasta_reader.cpp:596-629, src/main/node.cpp:537-575).                            */
int  pagan_msa_write_fasta(const pagan_msa *m, const char *path, int32_t chars_by_line);
/* ... with include_internal != 0 every node, in Node::get_all_nodes order (left subtree, node, right subtree;
 * node.h:277-290), internal nodes named #k# (node.h:479-495): Node::get_alignment(.., true), node.cpp:537-555. */
int  pagan_msa_write_fasta_nodes(const pagan_msa *m, const char *path, int32_t chars_by_line, int32_t include_internal);
void *pagan_msa_node_graph(const pagan_msa *m, int32_t node);   /* a pagan_hgraph (borrowed)  */
void pagan_msa_destroy(pagan_msa *m);

/* ---- pileup chain (BASELINE config 1: --queryfile reads --pileup-alignment --homopolymer) -----------------
 * Reads_aligner::pileup_alignment (src/main/reads_aligner.cpp:151-264): read 0 is the reference; every further
 * read is aligned (GPU) against the current root as a temporary two-child node -- root at distance 0.001, read at
 * query_distance, reads settings (basic_alignment.h:572-586) -- and joins the alignment when
 * overlap > min_overlap and identity > min_identity (read_alignment_scores, reads_aligner.cpp:3323-3465).      */
typedef struct pagan_pileup pagan_pileup;
typedef struct pagan_pileup_opts {
    int32_t  leaf_flags;         /* 1 = --454, 2 = --homopolymer (default): leaf graphs and ins = del = 0.25 */
    uint32_t dp_flags;           /* PAGAN_OPT_*                                                           */
    float    query_distance;     /* --query-distance, default 0.1 (settings.cpp:107)                      */
    float    min_overlap;        /* --min-query-overlap, default 0.5                                      */
    float    min_identity;       /* --min-query-identity, default 0.5                                     */
    int32_t  use_anchors;        /* 0 = full matrices (default), 1 = prefix anchors                       */
    int32_t  anchors_offset, prefix_hit_length, hit_trim;
    int32_t  device;             /* HIP device, -1 = current                                              */
} pagan_pileup_opts;
typedef struct pagan_pileup_step {
    int32_t read;                /* index of the read this step tried to add                              */
    int32_t accepted;
    float   overlap, identity;   /* aligned / read_length, matched / aligned                              */
    int32_t aligned, matched, read_length;
    int32_t left_sites, right_sites, status, n_cols;
    double  score;
    int64_t cells;
} pagan_pileup_step;
void pagan_pileup_default_opts(pagan_pileup_opts *o);
int  pagan_pileup_create(int32_t n_reads, const char *const *names, const char *const *seqs,
                         const pagan_pileup_opts *opts, pagan_pileup **out);
int  pagan_pileup_align(pagan_pileup *p);
int  pagan_pileup_n_steps(const pagan_pileup *p);                              /* n_reads - 1 */
int  pagan_pileup_step_info(const pagan_pileup *p, int32_t k, pagan_pileup_step *out);
int  pagan_pileup_step_job(const pagan_pileup *p, int32_t k, pagan_job *out);  /* borrowed views */
int  pagan_pileup_step_result(const pagan_pileup *p, int32_t k, pagan_result *out);
int  pagan_pileup_alignment_length(const pagan_pileup *p);
int  pagan_pileup_alignment_row(const pagan_pileup *p, int32_t read, char *buf);   /* "" for a rejected read */
int  pagan_pileup_set_batch_backend(pagan_pileup *p, pagan_batch_fn fn, void *user);  /* TEST SEAM, as for pagan_msa */
void pagan_pileup_destroy(pagan_pileup *p);

/* ---- host graphs on their own ---------------------------------------------------------- */
typedef struct pagan_hgraph pagan_hgraph;
pagan_hgraph *pagan_hgraph_leaf(const char *residues, const char *full_alphabet, int32_t flags);
/* Sequence::create_codon_sequence (src/main/sequence.cpp:306-359): one site per triplet of the nucleotide string,
 * state 61 (NNN) for what is not a sense codon; a plain chain.  pagan_hgraph_string on such a graph (and on its
 * parents) takes the three-letter names of pagan_codon_alphabet and writes three characters per site.            */
pagan_hgraph *pagan_hgraph_leaf_codon(const char *nucleotides);
/* flags: bit0 reads/keep-all-edges settings, bit1 --no-reduced-terminal-penalties            */
pagan_hgraph *pagan_hgraph_parent(pagan_hgraph *left, pagan_hgraph *right, const pagan_result *res,
                                  float left_branch, float right_branch, const int32_t *parsimony,
                                  int32_t n_states, int32_t char_as, int32_t flags);
/* The same parent built on the current HIP device (csrc/dp_parent.hip: Basic_alignment::build_ancestral_sequence,
 * basic_alignment.cpp:36-653, as maps, scans and per-site loops).  Field for field the graph
 * pagan_hgraph_parent returns; NULL without a device or on a HIP error (never a host-built graph in its place).
 * info[5] (may be NULL): runs of skipped sites, rounds of the boundary pass's fixpoint, deleted sites, edge weights outside
 * the log-weight table, log weights the host had to correct.  The tree walk uses it for nodes of 20,000 columns and more
 * (PAGAN_PARENTS=host / device forces either); pagan_parents_device_calls counts those.                            */
pagan_hgraph *pagan_hgraph_parent_device(pagan_hgraph *left, pagan_hgraph *right, const pagan_result *res,
                                         float left_branch, float right_branch, const int32_t *parsimony,
                                         int32_t n_states, int32_t char_as, int32_t flags, int32_t *info);
long long pagan_parents_device_calls(void);
void pagan_hgraph_view(const pagan_hgraph *g, pagan_graph *out);
/* site_attr [n_sites][8] = state,type,path_state,left,right,count_since_used,ambiguous,n_fwd;
 * site_dist [n_sites]; edge_attr [n_edges][6] = start,end,used,count_since_used,
 * count_as_skipped,linked; edge_f [n_edges][3] = weight,log_weight,dist_since_used.       */
void pagan_hgraph_attrs(const pagan_hgraph *g, int32_t *site_attr, float *site_dist,
                        int32_t *edge_attr, float *edge_f);
void pagan_hgraph_fwd(const pagan_hgraph *g, int32_t *fwd_off, int32_t *fwd_eid);
int  pagan_hgraph_string(const pagan_hgraph *g, int32_t with_gaps, const char *full_alphabet, char *out);
void pagan_hgraph_free(pagan_hgraph *g);

/* Viterbi_alignment::define_tunnel for --use-prefix-anchors; upper/lower hold
 * strlen(gapped1)+1 entries.  Returns the number of anchors used.                        */
int  pagan_define_tunnel(const char *s1, const char *s2, const char *gapped1, const char *gapped2,
                         int32_t prefix_hit_length, int32_t hit_trim, int32_t offset,
                         int32_t *upper, int32_t *lower);

/* The tunnel from a list of possibly overlapping hits, the reference's BLAST branch from the hit list onwards.
 * hits: n x 4 ints (start in sequence 1, start in sequence 2, length, score), positions in the UNGAPPED strings.
 * pagan_prefix_hits: Find_anchors::find_long_substrings (find_anchors.cpp:35-127); returns the number of hits.
 * pagan_drop_bad_hits: Find_anchors::eliminate_bad_hits (find_anchors.cpp:497-545), in place, returns the count.
 * pagan_define_tunnel_overlapping: Find_anchors::define_tunnel_with_overlapping_hits (find_anchors.cpp:643-843);
 *   upper/lower hold strlen(gapped1)+1 entries; blocks (cap x 4 ints: start x, start y, end x, end y) are the empty
 *   tunnel blocks ascending by size; returns their number.
 * pagan_force_gap: Viterbi_alignment::replace_largest_tunnel_block_with_gap_tunnel (viterbi_alignment.cpp:467-553)
 *   on the last (largest) block; returns 1 if it was replaced, 0 if no block of at least `threshold` cells is left. */
int  pagan_prefix_hits(const char *s1, const char *s2, int32_t min_length, int32_t *hits, int32_t cap);
/* diagnostic: how often the prefix-anchor finder ran on the device (csrc/dp_anchors.hip: suffix array by prefix doubling on the
 * GPU; PAGAN_ANCHORS=host keeps it on the host, which is also where it runs without a device)                               */
long long pagan_anchors_device_calls(void);
int  pagan_drop_bad_hits(int32_t *hits, int32_t n, int32_t thr_total, int32_t thr_partly);
int  pagan_define_tunnel_overlapping(const int32_t *hits, int32_t n, const char *gapped1, const char *gapped2, int32_t width,
                                     int32_t *upper, int32_t *lower, int32_t *blocks, int32_t cap);
int  pagan_force_gap(int32_t *upper, int32_t *lower, int32_t n, const int32_t *blocks, int32_t n_blocks, int32_t threshold,
                     int32_t width, int32_t wide);

/* DNA Evol_model for a distance: table [a + b*15] (15x15 floats), params[4] =
 * log_gap_open, log_gap_ext, log_gap_end_ext, log_non_gap; parsimony [i + j*15].          */
int  pagan_dna_model(const float base_freq[4], double distance, float *table, float *params,
                     int32_t *parsimony);
/* Protein (WAG) Evol_model for a distance: table [a + b*211] (211x211 floats), params as above,
 * parsimony [i + j*211] (model_factory.cpp:304-541, 1478-1595, 1871-1960, 2155-2219).              */
int  pagan_protein_model(double distance, float *table, float *params, int32_t *parsimony);
/* Codon (Kosiol & Goldman's empirical model) Evol_model for a distance: 1892 states = 61 sense codons, NNN,
 * 1830 codon pairs; table [a + b*1892], params as above, parsimony [i + j*1892]
 * (Model_factory::define_codon_alphabet / codon_model / alignment_model, model_factory.cpp:839-1217, 1599-1805,
 * 1871-1960, 2022-2092).                                                                                       */
int  pagan_codon_model(double distance, float *table, float *params, int32_t *parsimony);
/* names: 3 characters per state (>= 3*1892 + 1 bytes) -- what a state prints as (the ancestral character
 * alphabet, model_factory.cpp:1739-1803; its first 62 entries are the leaf alphabet, model_factory.h:209-221);
 * mostcommon [i + j*61] (model_factory.cpp:1208-1217).  Either may be NULL.  Returns the number of states.      */
int  pagan_codon_alphabet(char *names, int32_t *mostcommon);
/* Codon_translation::gapped_DNA_to_protein (src/utils/codon_translation.cpp:32-107): one amino-acid letter per triplet of
 * a codon string ("---" -> '-', X for what the table does not hold) -- the strings the anchors of a codon walk are found in
 * (viterbi_alignment.cpp:54-60, 141-145).  out holds strlen/3 + 2 bytes.  Returns the number of letters.               */
int  pagan_codon_translate(const char *codons, char *out);
/* Leaf states of a nucleotide string read as codons (sequence.cpp:318-336); states holds (strlen+2)/3 entries.
 * Returns the number of states.                                                                                */
int  pagan_codon_states(const char *nucleotides, int32_t *states);
/* The same model in probability space, what the forward/backward pass takes (pagan_model_prob):
 * score [a + b*S] = Evol_model::score, params[3] = gap_open, gap_ext, non_gap.  data_type 1 DNA, 2 protein,
 * 3 codon.                                                                                                    */
int  pagan_model_prob_table(int32_t data_type, const float *base_freq, double distance, float *score, float *params);
/* Alphabets of a data type (1 DNA, 2 protein): leaf_alphabet (state = position of the residue,
 * Sequence::full_char_alphabet) and ancestral_alphabet (the character an internal state prints as,
 * Model_factory::ancestral_character_alphabet); both >= 212 bytes.  Returns the number of states.  */
int  pagan_model_alphabets(int32_t data_type, char *leaf_alphabet, char *ancestral_alphabet);
/* Eigen::eigenQREV (src/utils/eigen.cpp:48-128) for a row-major n x n reversible rate matrix:
 * Q = U diag(root) V.  Exposed for the tests.                                                     */
int  pagan_eigen_qrev(const double *Q, const double *pi, int32_t n, double *root, double *U, double *V);

#ifdef __cplusplus
}
#endif
#endif /* PAGAN_HOST_H */
