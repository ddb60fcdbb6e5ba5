// dp_parent.hip -- the parent sequence graph from the alignment path, on the device.
//
// Reference: Basic_alignment::build_ancestral_sequence (src/main/basic_alignment.cpp:36-59) =
//   create_ancestral_sequence   :61-179    one parent site per path column
//   create_ancestral_edges      :181-368   the children's bwd edges carried over, left child first (transfer_child_edge :510-653)
//   check_skipped_boundaries    :370-489   skip counters (pass 1) and deletion of over-skipped ranges (pass 2, delete_edge_range :491-508)
// The host builder (host_graph.cpp: make_parent) walks the columns once and appends to linked lists.  Here every step is a
// map, a scan or a small per-site loop:
//   pp_sites      one thread per column: the site's fields and the child -> parent index maps          (:61-179)
//   pp_edges      one thread per parent site: the site's candidate edges in the reference's order -- left child's bwd list,
//                 the X-after-Y edge, right child's list, the Y-after-X edge -- with the duplicate rule, the skip limits and the
//                 weight / history rules, into a scratch segment of the site.  A site's edges depend on that site only: every
//                 transferred edge ENDS at the site whose child site it enters.  One exception, the stop site: a span-1 child
//                 edge into the child's stop site that maps to a longer parent edge is shortened to (s, s+1) (:526-541) and so
//                 joins site s+1's list, behind everything that site got; pp_edges_stop does the stop site alone, afterwards.
//   scans         edge ids are creation order = site-major order of the scratch segments; bwd lists are the segments themselves
//   pp_emit       segments -> edge arrays, bwd CSR, counts of the fwd lists
//   pp_fwd_*      a site's fwd list is its edges in creation order: scatter by start, then each (short) segment sorted by id
//   pp_pass1      :377-420, increments by atomics (they commute)
//   pp_bound      :424-488.  The scan over sites is a state machine that resets at every gapped / matched site, so it falls into
//                 RUNS of skipped sites; a run's decision reads the bwd lists of its first site and of the matched site behind
//                 it, and those lists may have lost edges to EARLIER runs' deletions -- nothing else couples the runs.  All runs
//                 are evaluated side by side against the current deletion flags, again and again until nothing changes; run k is
//                 final after k+1 rounds at the latest (it depends on earlier runs only), in practice after one or two, and a
//                 fixpoint is the sequential result by the same induction.
//   pp_final_*    deleted sites become non_real; every list loses the edges that touch one (:491-508), order kept.
// float arithmetic as on the host (same operations on the same values; -ffp-contract=off); an edge's log weight is looked up
// in a table the host makes with ITS logf for the weights the rules can produce (1, 0.9, 0.25 times powers of the skip
// probability) -- a device logf is not glibc's; a weight outside the table is counted by the kernels, and the host then re-derives
// the logarithm of every weight != 1 after the download (only then: with the table complete the device's words are the host's).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "host_graph.h"

namespace pagan {

namespace {

enum { kEnds = 0, kMatched = 2, kXGapped = 3, kYGapped = 4, kXSkipped = 5, kYSkipped = 6 };

struct PPChild {
    int n_sites;
    const int32_t *state; const uint8_t *ambiguous; const int32_t *count_since_used; const float *dist_since_used;
    const int32_t *bwd_off, *bwd_eid;
    const int32_t *e_start, *e_end; const float *e_w; const int32_t *e_cnt, *e_cas; const float *e_dist; uint8_t *e_used;
};

struct PPCand { int32_t s, e, cnt, cas; float w, dist; };

#define PP_TAB 96
struct PPArgs {
    PPChild L, R;
    const int32_t *cols;            // [n_cols][3] left, right, path_state
    int n;                          // parent sites
    float lbl, rbl;
    const int32_t *pars; int S, char_as;
    float max_skip_distance; int max_skip_branches, max_match_skip_branches; float skip_prob; int reduced_terminal;
    // parent arrays
    int32_t *state; int8_t *site_type, *path_state; int32_t *child_l, *child_r, *count_since_used; float *dist_since_used; uint8_t *ambiguous;
    int32_t *e_start, *e_end; float *e_w, *e_logw; int32_t *e_cnt, *e_cas; float *e_dist; uint8_t *e_used;
    int32_t *bwd_off, *bwd_eid, *bwd_src; float *bwd_logw; int32_t *fwd_off, *fwd_eid;
    // scratch
    int32_t *lci, *rci;             // child site -> parent site
    int32_t *cap, *coff;            // candidate capacity of a site, its scan
    PPCand *cand;
    int32_t *ncr, *eoff;            // edges a site created, their scan (edge ids)
    int32_t *nb, *boff;             // bwd list length, its scan (first CSR)
    int32_t *nf, *foff, *ffill;     // fwd list length, its scan, scatter cursors
    int32_t *beid_t, *feid_t;       // first CSR (before deletions)
    int32_t *run_f, *run_m, *run_a, *run_b;   // runs of skipped sites: first site, the non-skipped site behind, deleted interval
    uint8_t *deleted;
    int32_t *nb2, *nf2;
    int32_t *info;                  // [0] edges, [1] bwd entries, [2] fwd entries, [3] runs, [4] rounds, [5] deleted sites, [6] weights not in the table
    int n_tab;
    float tab_w[PP_TAB], tab_logw[PP_TAB];
};

__device__ __forceinline__ bool pp_skipped(int p) { return p == kXSkipped || p == kYSkipped; }

// ---- sites: create_ancestral_sequence, basic_alignment.cpp:61-179 ----
__global__ __launch_bounds__(256) void pp_sites(PPArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    int st = -1, type = kRealSite, ps = kEnds, cl = -1, cr = -1, cnt = 0, amb = 0;
    float dist = 0.0f;
    if (i == 0) { type = kStartSite; cl = 0; cr = 0; A.lci[0] = 0; A.rci[0] = 0; }
    else if (i == A.n - 1) { type = kStopSite; cl = A.L.n_sites - 1; cr = A.R.n_sites - 1; A.lci[cl] = i; A.rci[cr] = i; }
    else {
        const int l = A.cols[3 * (i - 1)], r = A.cols[3 * (i - 1) + 1];
        ps = A.cols[3 * (i - 1) + 2];
        if (ps == kMatched) {
            const int lc = A.L.state[l], rc = A.R.state[r];
            st = A.pars[lc + rc * A.S]; cl = l; cr = r;
            amb = (lc != rc || lc >= A.char_as) ? 1 : 0;
            A.lci[l] = i; A.rci[r] = i;
        } else if (ps == kXGapped || ps == kXSkipped) {
            st = A.L.state[l]; cl = l; amb = A.L.ambiguous[l];
            if (ps == kXSkipped) { cnt = A.L.count_since_used[l] + 1; dist = A.L.dist_since_used[l] + A.lbl; }
            A.lci[l] = i;
        } else {
            st = A.R.state[r]; cr = r; amb = A.R.ambiguous[r];
            if (ps == kYSkipped) { cnt = A.R.count_since_used[r] + 1; dist = A.R.dist_since_used[r] + A.rbl; }
            A.rci[r] = i;
        }
    }
    A.state[i] = st; A.site_type[i] = (int8_t)type; A.path_state[i] = (int8_t)ps; A.child_l[i] = cl; A.child_r[i] = cr;
    A.count_since_used[i] = cnt; A.dist_since_used[i] = dist; A.ambiguous[i] = (uint8_t)amb;
    int cap = 0;
    if (i > 0) {
        if (cl >= 0) cap += A.L.bwd_off[cl + 1] - A.L.bwd_off[cl];
        if (cr >= 0) cap += A.R.bwd_off[cr + 1] - A.R.bwd_off[cr];
        cap += 2;
    }
    A.cap[i] = cap;
}

__global__ void pp_mark_used(uint8_t *used, const int32_t *ids, int n) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) used[ids[k]] = 1;
}

// ---- one block, any length: exclusive scan (out[n] = total) ----
__global__ __launch_bounds__(1024) void pp_scan(const int32_t *in, int32_t *out, int n, int32_t *total) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024;
    const int a = min(t * per, n), b = min(a + per, n);
    int s = 0;
    for (int k = a; k < b; ++k) s += in[k];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int k = a; k < b; ++k) { const int v = in[k]; out[k] = run; run += v; }
    if (t == 1023) { out[n] = part[1023]; if (total) *total = part[1023]; }
}

// transfer_child_edge (basic_alignment.cpp:510-653; weight_edges / pair_end_reads off) is the `transfer` lambda below.
// One site's edges.  STOP: the site is the stop site -- its shortened edges join the list of site s+1.
template <bool STOP>
__device__ __forceinline__ void pp_site_edges(const PPArgs &A, int i) {
    const int n = A.n;
    PPCand *own = A.cand + A.coff[i];
    int n_own = 0;
    const int ps = A.path_state[i], prev = A.path_state[i - 1];
    const int li = A.child_l[i], ri = A.child_r[i];
    auto reset = [](PPCand &x) { x.cas = 0; x.cnt = 0; x.dist = 0.0f; x.w = 1.0f; };      // :579-583, sequence.h:452-502
    auto transfer = [&](const PPChild &C, int ce, const int32_t *ci, float bl) {
        int s = ci[C.e_start[ce]], e = ci[C.e_end[ce]];
        const int span = C.e_end[ce] - C.e_start[ce];
        if (A.reduced_terminal) {                                             // :526-541
            if (s == 0 && e - s > 1 && span == 1) s = e - 1;
            if (e == n - 1 && e - s > 1 && span == 1) e = s + 1;
        }
        // Site::contains_bwd_edge on the list of site e: every entry with this start is reset, and nothing is added
        bool dup = false;
        if (STOP && e != i) {
            PPCand *theirs = A.cand + A.coff[e];
            for (int k = 0; k < A.ncr[e]; ++k) if (theirs[k].s == s) { reset(theirs[k]); dup = true; }
            for (int k = 0; k < n_own; ++k) if (own[k].e == e && own[k].s == s) { reset(own[k]); dup = true; }
        } else {
            for (int k = 0; k < n_own; ++k) if (own[k].e == e && own[k].s == s) { reset(own[k]); dup = true; }
        }
        if (dup) return;
        const bool used = C.e_used[ce] != 0;
        if (!used && C.e_cnt[ce] + 1 > A.max_skip_branches) return;            // :587
        if (!used && C.e_dist[ce] + bl > A.max_skip_distance) return;          // :591
        const float ds = A.dist_since_used[s], de = A.dist_since_used[e];
        const int cs = A.count_since_used[s], cend = A.count_since_used[e];
        float w = 1.0f, dist = 0.0f;
        int cnt = 0;
        const float factor = (float)(1.0f * (double)C.e_w[ce] * A.skip_prob);  // :613
        if (ds != de || cs != cend) {                                          // :604-617
            dist = fmaxf(ds, de); cnt = max(cs, cend); w *= factor;
        } else if (!used && cs == 0 && cend == 0) {                            // :619-632
            dist = C.e_dist[ce] + bl; cnt = C.e_cnt[ce] + 1; w *= factor;
        } else if (!used) {                                                    // :633-637
            dist = C.e_dist[ce] + bl; cnt = C.e_cnt[ce] + 1;
        }
        PPCand x; x.s = s; x.e = e; x.cnt = cnt; x.cas = used ? 0 : C.e_cas[ce]; x.w = w; x.dist = dist;      // :643-646
        own[n_own++] = x;
    };
    auto plain = [&]() { PPCand x; x.s = i - 1; x.e = i; x.cnt = 0; x.cas = 0; x.w = 1.0f; x.dist = 0.0f; own[n_own++] = x; };
    if (li >= 0) {
        for (int k = A.L.bwd_off[li]; k < A.L.bwd_off[li + 1]; ++k) transfer(A.L, A.L.bwd_eid[k], A.lci, A.lbl);
        if ((ps == kXGapped || ps == kXSkipped) && (prev == kYGapped || prev == kYSkipped)) plain();          // :288-296
    }
    if (ri >= 0) {
        for (int k = A.R.bwd_off[ri]; k < A.R.bwd_off[ri + 1]; ++k) transfer(A.R, A.R.bwd_eid[k], A.rci, A.rbl);
        if ((ps == kYGapped || ps == kYSkipped) && (prev == kXGapped || prev == kXSkipped)) plain();          // :351-358
    }
    A.ncr[i] = n_own;
    if (!STOP) { A.nb[i] = n_own; return; }
    int stay = 0;
    for (int k = 0; k < n_own; ++k) { if (own[k].e == i) ++stay; else A.nb[own[k].e] += 1; }
    A.nb[i] = stay;
}

__global__ __launch_bounds__(128) void pp_edges(PPArgs A) {
    const int i = blockIdx.x * 128 + threadIdx.x + 1;
    if (i >= A.n - 1) return;
    pp_site_edges<false>(A, i);
}
__global__ void pp_edges_stop(PPArgs A) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        A.ncr[0] = 0; A.nb[0] = 0;
        if (A.n >= 2) pp_site_edges<true>(A, A.n - 1);
    }
}

__device__ __forceinline__ float pp_logw(const PPArgs &A, float w) {
    if (w == 1.0f) return 0.0f;
    for (int k = 0; k < A.n_tab; ++k) if (A.tab_w[k] == w) return A.tab_logw[k];
    atomicAdd(A.info + 6, 1);
    return (float)log((double)w);                                              // (the host puts its own logf there: make_parent_device)
}

// segments -> edge arrays and the first bwd CSR; fwd list lengths
__global__ __launch_bounds__(128) void pp_emit(PPArgs A) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= A.n) return;
    const PPCand *own = A.cand + A.coff[i];
    const int base = A.eoff[i], nc = A.ncr[i];
    int at = A.boff[i];
    for (int k = 0; k < nc; ++k) {
        const PPCand x = own[k];
        const int id = base + k;
        A.e_start[id] = x.s; A.e_end[id] = x.e; A.e_w[id] = x.w; A.e_logw[id] = pp_logw(A, x.w);
        A.e_cnt[id] = x.cnt; A.e_cas[id] = x.cas; A.e_dist[id] = x.dist; A.e_used[id] = 0;
        atomicAdd(A.nf + x.s, 1);
        if (x.e == i) A.beid_t[at++] = id;
    }
    if (i == A.n - 1) {
        // the stop site's shortened edges: at the tail of their end sites' lists, in creation order
        for (int k = 0; k < nc; ++k) {
            const PPCand x = own[k];
            if (x.e == i) continue;
            int pos = A.boff[x.e] + A.ncr[x.e];
            for (int q = 0; q < k; ++q) if (own[q].e == x.e) ++pos;
            A.beid_t[pos] = base + k;
        }
    }
}

__global__ __launch_bounds__(256) void pp_fwd_scatter(PPArgs A) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= A.info[0]) return;                                               // (the number of edges is the scan's total)
    const int s = A.e_start[id];
    A.feid_t[A.foff[s] + atomicAdd(A.ffill + s, 1)] = id;
}
__global__ __launch_bounds__(256) void pp_fwd_sort(PPArgs A) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= A.n) return;
    int32_t *v = A.feid_t + A.foff[s];
    const int c = A.foff[s + 1] - A.foff[s];
    for (int a = 1; a < c; ++a) {                                              // creation order = ascending id
        const int x = v[a];
        int b = a - 1;
        while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
        v[b + 1] = x;
    }
}

// largest start index among the bwd edges of site i whose start is not deleted, first wins ties (:383-390); -1: none
// (`before`: only deletions of sites below it count -- a run decides BEFORE its own sites go)
__device__ __forceinline__ int pp_max_start_bwd(const PPArgs &A, int i, const uint8_t *deleted, int before) {
    int best = -1, bs = -1;
    for (int k = A.boff[i]; k < A.boff[i + 1]; ++k) {
        const int e = A.beid_t[k], s = A.e_start[e];
        if (deleted && s < before && deleted[s]) continue;
        if (s > bs) { bs = s; best = e; }
    }
    return best;
}

// check_skipped_boundaries, pass 1 (:377-420)
__global__ __launch_bounds__(256) void pp_pass1(PPArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const int ts = A.path_state[i];
    if (!pp_skipped(ts)) return;
    const int e = pp_max_start_bwd(A, i, nullptr, 0);
    if (e >= 0) {
        const int ps = A.path_state[A.e_start[e]];
        if (ps == kMatched || ps == kEnds) atomicAdd(A.e_cas + e, 1);          // :394-398
    }
    if (A.foff[i + 1] > A.foff[i]) {
        const int f = A.feid_t[A.foff[i]];                                     // the first fwd edge stays (:403-410)
        const int ns = A.path_state[A.e_end[f]];
        if (ns == kMatched || ns == kEnds) atomicAdd(A.e_cas + f, 1);          // :414-418
    }
}

// check_skipped_boundaries, pass 2 (:424-488) with delete_edge_range (:491-508): see the file header
__global__ __launch_bounds__(1024) void pp_bound(PPArgs A) {
    __shared__ int n_runs, changed;
    const int t = threadIdx.x, n = A.n;
    if (t == 0) n_runs = 0;
    __syncthreads();
    for (int i = 1 + t; i < n - 1; i += 1024) {
        if (!pp_skipped(A.path_state[i]) || pp_skipped(A.path_state[i - 1])) continue;
        int m = i + 1;
        while (m < n - 1 && pp_skipped(A.path_state[m])) ++m;
        const int r = atomicAdd(&n_runs, 1);
        A.run_f[r] = i; A.run_m[r] = m; A.run_a[r] = 0; A.run_b[r] = -1;
    }
    __syncthreads();
    const int R = n_runs;
    int rounds = 0;
    for (;;) {
        if (t == 0) changed = 0;
        __syncthreads();
        for (int r = t; r < R; r += 1024) {
            const int f = A.run_f[r], m = A.run_m[r];
            int a = 0, b = -1;
            const int e0 = pp_max_start_bwd(A, f, A.deleted, f);
            if (e0 >= 0 && A.e_cas[e0] > A.max_match_skip_branches && A.path_state[m] == kMatched) {
                int edge = -1;
                for (int k = A.boff[m]; k < A.boff[m + 1]; ++k) {
                    const int e = A.beid_t[k];
                    if (A.e_start[e] < f && A.deleted[A.e_start[e]]) continue;  // (gone with an earlier run; this run's own sites are still there)
                    if (A.e_cas[e] > A.max_match_skip_branches) edge = e;      // the last such edge (:456-470)
                }
                if (edge >= 0) { a = f; b = A.e_start[edge]; }                 // sites b, b-1, .., a (nothing if b < a)
            }
            if (b < a) { a = 0; b = -1; }
            if (a != A.run_a[r] || b != A.run_b[r]) {
                A.run_a[r] = a; A.run_b[r] = b;
                for (int s = f; s < m; ++s) A.deleted[s] = (s >= a && s <= b) ? 1 : 0;
                changed = 1;
            }
        }
        __threadfence_block();
        __syncthreads();
        ++rounds;
        const int c = changed;
        __syncthreads();
        if (!c || rounds > R + 2) break;
    }
    if (t == 0) { A.info[3] = R; A.info[4] = rounds; }
}

// what is left of the lists once the deleted sites are gone
__global__ __launch_bounds__(256) void pp_final_count(PPArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    int b = 0, f = 0;
    if (A.deleted[i]) { A.site_type[i] = (int8_t)kNonReal; atomicAdd(A.info + 5, 1); }
    else {
        for (int k = A.boff[i]; k < A.boff[i + 1]; ++k) b += A.deleted[A.e_start[A.beid_t[k]]] ? 0 : 1;
        for (int k = A.foff[i]; k < A.foff[i + 1]; ++k) f += A.deleted[A.e_end[A.feid_t[k]]] ? 0 : 1;
    }
    A.nb2[i] = b; A.nf2[i] = f;
}
__global__ __launch_bounds__(256) void pp_final_lists(PPArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n || A.deleted[i]) return;
    int at = A.bwd_off[i];
    for (int k = A.boff[i]; k < A.boff[i + 1]; ++k) {
        const int e = A.beid_t[k], s = A.e_start[e];
        if (A.deleted[s]) continue;
        A.bwd_eid[at] = e; A.bwd_src[at] = s; A.bwd_logw[at] = A.e_logw[e];
        ++at;
    }
    at = A.fwd_off[i];
    for (int k = A.foff[i]; k < A.foff[i + 1]; ++k) {
        const int e = A.feid_t[k];
        if (!A.deleted[A.e_end[e]]) A.fwd_eid[at++] = e;
    }
}

#define PPHIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return false; } } while (0)

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// device memory of finished builds, kept for the next one (a level of a walk is followed by the next)
struct MemPool {
    std::mutex m;
    struct Slab { int device; char *p; size_t cap; };
    std::vector<Slab> idle;
    char *take(int device, size_t need, size_t *cap) {
        {
            std::lock_guard<std::mutex> g(m);
            int best = -1;
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].device == device && idle[k].cap >= need && (best < 0 || idle[k].cap < idle[best].cap)) best = (int)k;
            if (best >= 0 && idle[best].cap <= 4 * need + (1u << 20)) {
                Slab s = idle[best]; idle.erase(idle.begin() + best); *cap = s.cap; return s.p;
            }
        }
        char *p = nullptr;
        *cap = need + need / 8;
        if (hipMalloc((void **)&p, *cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void give(int device, char *p, size_t cap) {
        if (!p) return;
        std::lock_guard<std::mutex> g(m);
        if (idle.size() >= 48) {                                               // (bounded: the oldest goes)
            int cur = -1;
            (void)hipGetDevice(&cur);                                          // (the calling thread stays on the device it was on)
            (void)hipSetDevice(idle.front().device); (void)hipFree(idle.front().p); idle.erase(idle.begin());
            if (cur >= 0) (void)hipSetDevice(cur);
        }
        idle.push_back({device, p, cap});
    }
    void clear() {
        std::lock_guard<std::mutex> g(m);
        int cur = -1;
        (void)hipGetDevice(&cur);
        for (auto &s : idle) { (void)hipSetDevice(s.device); (void)hipFree(s.p); }
        idle.clear();
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};
MemPool pp_pool;

struct Carve {
    char *base; size_t cur = 0;
    template <class T> T *take(size_t n) { T *p = base ? (T *)(base + cur) : nullptr; cur += up256(sizeof(T) * (n ? n : 1)); return p; }
};

} // namespace

// A SeqGraph's copy on a device: what the builder reads of a child (and writes of a parent).
struct DevGraph {
    int device = -1, n_sites = 0, n_edges = 0;
    char *mem = nullptr; size_t cap = 0;
    int32_t *state = nullptr; int8_t *site_type = nullptr, *path_state = nullptr; int32_t *child_l = nullptr, *child_r = nullptr, *count_since_used = nullptr;
    float *dist_since_used = nullptr; uint8_t *ambiguous = nullptr;
    int32_t *e_start = nullptr, *e_end = nullptr; float *e_w = nullptr, *e_logw = nullptr; int32_t *e_cnt = nullptr, *e_cas = nullptr; float *e_dist = nullptr; uint8_t *e_used = nullptr;
    int32_t *bwd_off = nullptr, *bwd_eid = nullptr, *bwd_src = nullptr; float *bwd_logw = nullptr; int32_t *fwd_off = nullptr, *fwd_eid = nullptr;
    ~DevGraph() { if (mem) pp_pool.give(device, mem, cap); }
    void carve(int n, int e_cap) {
        Carve c{mem};
        lay(c, n, e_cap);
    }
    static size_t bytes(int n, int e_cap) { Carve c{nullptr}; DevGraph d; d.lay(c, n, e_cap); return c.cur; }
    void lay(Carve &c, int n, int ec) {
        state = c.take<int32_t>(n); site_type = c.take<int8_t>(n); path_state = c.take<int8_t>(n); child_l = c.take<int32_t>(n);
        child_r = c.take<int32_t>(n); count_since_used = c.take<int32_t>(n); dist_since_used = c.take<float>(n); ambiguous = c.take<uint8_t>(n);
        e_start = c.take<int32_t>(ec); e_end = c.take<int32_t>(ec); e_w = c.take<float>(ec); e_logw = c.take<float>(ec);
        e_cnt = c.take<int32_t>(ec); e_cas = c.take<int32_t>(ec); e_dist = c.take<float>(ec); e_used = c.take<uint8_t>(ec);
        bwd_off = c.take<int32_t>(n + 1); bwd_eid = c.take<int32_t>(ec); bwd_src = c.take<int32_t>(ec); bwd_logw = c.take<float>(ec);
        fwd_off = c.take<int32_t>(n + 1); fwd_eid = c.take<int32_t>(ec);
    }
};

void parent_release_cache() { pp_pool.clear(); }

// The states of a graph changed on the host after it was built (--mostcommon: Node::fix_ambiguous_states, node.cpp:1610-1690,
// rewrites the states of a node and of the ambiguous sites below it): its copy on the device -- which the build one level up
// reads the children's states from -- takes them over.  false: the copy could not be updated (the caller drops it).
bool parent_update_states(SeqGraph &g) {
    DevGraph *d = static_cast<DevGraph *>(g.dev.get());
    if (!d) return true;
    if (d->n_sites != g.n_sites()) return false;
    int cur = -1;
    (void)hipGetDevice(&cur);
    bool ok = hipSetDevice(d->device) == hipSuccess &&
              hipMemcpy(d->state, g.state.data(), sizeof(int32_t) * (size_t)d->n_sites, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    if (cur >= 0) (void)hipSetDevice(cur);
    return ok;
}

namespace {

// the child's arrays on `device` (uploaded once; a parent built there is resident already)
bool resident(SeqGraph &g, int device, hipStream_t st, DevGraph **out) {
    DevGraph *d = static_cast<DevGraph *>(g.dev.get());
    if (d && d->device == device && d->n_sites == g.n_sites() && d->n_edges == g.n_edges()) { *out = d; return true; }
    std::shared_ptr<DevGraph> nd(new DevGraph());
    nd->device = device; nd->n_sites = g.n_sites(); nd->n_edges = g.n_edges();
    const int n = g.n_sites(), ne = std::max(g.n_edges(), 1);
    nd->mem = pp_pool.take(device, DevGraph::bytes(n, ne), &nd->cap);
    if (!nd->mem) return false;
    nd->carve(n, ne);
    auto up = [&](void *dst, const void *src, size_t bytes) { return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess; };
    bool ok = up(nd->state, g.state.data(), 4 * (size_t)n) && up(nd->ambiguous, g.ambiguous.data(), n) &&
              up(nd->count_since_used, g.count_since_used.data(), 4 * (size_t)n) && up(nd->dist_since_used, g.dist_since_used.data(), 4 * (size_t)n) &&
              up(nd->bwd_off, g.bwd_off.data(), 4 * (size_t)(n + 1)) && up(nd->bwd_eid, g.bwd_eid.data(), 4 * g.bwd_eid.size()) &&
              up(nd->e_start, g.e_start.data(), 4 * (size_t)g.n_edges()) && up(nd->e_end, g.e_end.data(), 4 * (size_t)g.n_edges()) &&
              up(nd->e_w, g.e_w.data(), 4 * (size_t)g.n_edges()) && up(nd->e_cnt, g.e_count_since_used.data(), 4 * (size_t)g.n_edges()) &&
              up(nd->e_cas, g.e_count_as_skipped.data(), 4 * (size_t)g.n_edges()) && up(nd->e_dist, g.e_dist_since_used.data(), 4 * (size_t)g.n_edges()) &&
              up(nd->e_used, g.e_used.data(), (size_t)g.n_edges());
    if (!ok) { (void)hipGetLastError(); return false; }
    g.dev = nd;
    *out = nd.get();
    return true;
}

PPChild child_view(const DevGraph &d) {
    PPChild c;
    c.n_sites = d.n_sites; c.state = d.state; c.ambiguous = d.ambiguous; c.count_since_used = d.count_since_used; c.dist_since_used = d.dist_since_used;
    c.bwd_off = d.bwd_off; c.bwd_eid = d.bwd_eid; c.e_start = d.e_start; c.e_end = d.e_end; c.e_w = d.e_w; c.e_cnt = d.e_cnt; c.e_cas = d.e_cas;
    c.e_dist = d.e_dist; c.e_used = d.e_used;
    return c;
}

struct StreamPool {
    std::mutex m;
    std::vector<std::pair<int, hipStream_t>> idle;
    hipStream_t take(int device) {
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t k = 0; k < idle.size(); ++k) if (idle[k].first == device) { hipStream_t s = idle[k].second; idle.erase(idle.begin() + k); return s; }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return s;
    }
    void give(int device, hipStream_t s) { if (s) { std::lock_guard<std::mutex> g(m); idle.emplace_back(device, s); } }
};
StreamPool pp_streams;

} // namespace

// make_parent (host_graph.cpp) on `device`: the same graph, field for field.  false: no device / a HIP error / a device
// status -- nothing has been changed that the host builder would not change too (the children's used marks), the caller
// takes the host builder.  The parent's device copy stays with it (SeqGraph::dev) for the build one level up.
bool make_parent_device(SeqGraph &left, SeqGraph &right, const pagan_result &res, float lbl, float rbl, const int32_t *parsimony,
                        int S, int char_as, const BuildSettings &bs, int device, SeqGraph *out, ParentBuildInfo *info_out) {
    const bool verbose = std::getenv("PAGAN_DP_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return false; } }
    else PPHIP(hipSetDevice(device));
    for (int k = 0; k < res.n_left_used; ++k) left.e_used[res.left_used[k]] = 1;
    for (int k = 0; k < res.n_right_used; ++k) right.e_used[res.right_used[k]] = 1;
    hipStream_t st = pp_streams.take(device);
    if (!st) return false;
    struct Lease { int d; hipStream_t s; ~Lease() { pp_streams.give(d, s); } } lease{device, st};
    DevGraph *dl = nullptr, *dr = nullptr;
    const bool l_was = left.dev && static_cast<DevGraph *>(left.dev.get())->device == device;
    const bool r_was = right.dev && static_cast<DevGraph *>(right.dev.get())->device == device;
    if (!resident(left, device, st, &dl) || !resident(right, device, st, &dr)) return false;

    const double t1 = now();
    const int n = res.n_cols + 2;
    const int nbL = left.bwd_off[left.n_sites()], nbR = right.bwd_off[right.n_sites()];
    const int e_cap = nbL + nbR + 2 * n;
    std::shared_ptr<DevGraph> pd(new DevGraph());
    pd->device = device; pd->n_sites = n;
    pd->mem = pp_pool.take(device, DevGraph::bytes(n, e_cap), &pd->cap);
    if (!pd->mem) return false;
    pd->carve(n, e_cap);

    PPArgs A;
    std::memset(&A, 0, sizeof(A));
    // scratch (one slab, returned to the pool at the end)
    size_t scap = 0;
    char *smem = nullptr;
    const int nu_l = res.n_left_used, nu_r = res.n_right_used;
    int32_t *d_cols = nullptr, *d_pars = nullptr, *d_lu = nullptr, *d_ru = nullptr;
    auto lay = [&](char *base) {
        Carve c{base};
        d_cols = c.take<int32_t>(3 * (size_t)res.n_cols); d_pars = c.take<int32_t>((size_t)S * S);
        d_lu = c.take<int32_t>(nu_l); d_ru = c.take<int32_t>(nu_r);
        A.lci = c.take<int32_t>(left.n_sites()); A.rci = c.take<int32_t>(right.n_sites());
        A.cap = c.take<int32_t>(n + 1); A.coff = c.take<int32_t>(n + 1);
        A.cand = c.take<PPCand>(e_cap);
        A.ncr = c.take<int32_t>(n + 1); A.eoff = c.take<int32_t>(n + 1); A.nb = c.take<int32_t>(n + 1); A.boff = c.take<int32_t>(n + 1);
        A.nf = c.take<int32_t>(n + 1); A.foff = c.take<int32_t>(n + 1); A.ffill = c.take<int32_t>(n + 1);
        A.beid_t = c.take<int32_t>(e_cap); A.feid_t = c.take<int32_t>(e_cap);
        A.run_f = c.take<int32_t>(n); A.run_m = c.take<int32_t>(n); A.run_a = c.take<int32_t>(n); A.run_b = c.take<int32_t>(n);
        A.deleted = c.take<uint8_t>(n + 1); A.nb2 = c.take<int32_t>(n + 1); A.nf2 = c.take<int32_t>(n + 1);
        A.info = c.take<int32_t>(16);
        return c.cur;
    };
    const size_t sneed = lay(nullptr);
    smem = pp_pool.take(device, sneed, &scap);
    if (!smem) return false;
    struct Scratch { int d; char *p; size_t c; ~Scratch() { pp_pool.give(d, p, c); } } scratch{device, smem, scap};
    lay(smem);
    // Whatever way this function is left (a failed launch or copy returns early), nothing goes back to the pools -- the scratch
    // slab, the parent's slab, the stream -- while kernels or copies queued on the stream may still touch it: this object is the
    // last one declared, so it is destroyed first.  (round 4 advisor: the early returns handed memory in flight to the next build)
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); (void)hipGetLastError(); } } drain{st};

    A.L = child_view(*dl); A.R = child_view(*dr);
    A.cols = d_cols; A.n = n; A.lbl = lbl; A.rbl = rbl; A.pars = d_pars; A.S = S; A.char_as = char_as;
    A.max_skip_distance = bs.max_skip_distance; A.max_skip_branches = bs.max_skip_branches; A.max_match_skip_branches = bs.max_match_skip_branches;
    A.skip_prob = bs.branch_skip_probability; A.reduced_terminal = bs.reduced_terminal ? 1 : 0;
    A.state = pd->state; A.site_type = pd->site_type; A.path_state = pd->path_state; A.child_l = pd->child_l; A.child_r = pd->child_r;
    A.count_since_used = pd->count_since_used; A.dist_since_used = pd->dist_since_used; A.ambiguous = pd->ambiguous;
    A.e_start = pd->e_start; A.e_end = pd->e_end; A.e_w = pd->e_w; A.e_logw = pd->e_logw; A.e_cnt = pd->e_cnt; A.e_cas = pd->e_cas;
    A.e_dist = pd->e_dist; A.e_used = pd->e_used;
    A.bwd_off = pd->bwd_off; A.bwd_eid = pd->bwd_eid; A.bwd_src = pd->bwd_src; A.bwd_logw = pd->bwd_logw; A.fwd_off = pd->fwd_off; A.fwd_eid = pd->fwd_eid;
    // weights the rules can produce, with the host's logf: w -> (float)(1 * (double) w * p) from the leaves' constants
    {
        const float starts[3] = {1.0f, 0.9f, 0.25f};
        int nt = 0;
        for (float w0 : starts) {
            float w = w0;
            for (int k = 0; k < PP_TAB / 3 && nt < PP_TAB; ++k) {
                bool have = false;
                for (int q = 0; q < nt; ++q) have = have || A.tab_w[q] == w;
                if (!have) { A.tab_w[nt] = w; A.tab_logw[nt] = std::log(w); ++nt; }
                w = (float)(1.0f * (double)w * bs.branch_skip_probability);
            }
        }
        A.n_tab = nt;
    }
    PPHIP(hipMemcpyAsync(d_cols, res.cols, sizeof(int32_t) * 3 * (size_t)res.n_cols, hipMemcpyHostToDevice, st));
    PPHIP(hipMemcpyAsync(d_pars, parsimony, sizeof(int32_t) * (size_t)S * S, hipMemcpyHostToDevice, st));
    if (nu_l) PPHIP(hipMemcpyAsync(d_lu, res.left_used, sizeof(int32_t) * nu_l, hipMemcpyHostToDevice, st));
    if (nu_r) PPHIP(hipMemcpyAsync(d_ru, res.right_used, sizeof(int32_t) * nu_r, hipMemcpyHostToDevice, st));
    // (a child uploaded just now carries the marks already; one that was resident gets them here)
    if (l_was && nu_l) hipLaunchKernelGGL(pp_mark_used, dim3((nu_l + 255) / 256), dim3(256), 0, st, dl->e_used, d_lu, nu_l);
    if (r_was && nu_r) hipLaunchKernelGGL(pp_mark_used, dim3((nu_r + 255) / 256), dim3(256), 0, st, dr->e_used, d_ru, nu_r);
    PPHIP(hipMemsetAsync(A.nf, 0, sizeof(int32_t) * (n + 1), st));
    PPHIP(hipMemsetAsync(A.ffill, 0, sizeof(int32_t) * (n + 1), st));
    PPHIP(hipMemsetAsync(A.deleted, 0, n + 1, st));
    PPHIP(hipMemsetAsync(A.info, 0, sizeof(int32_t) * 16, st));
    hipLaunchKernelGGL(pp_sites, dim3((n + 255) / 256), dim3(256), 0, st, A);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.cap, A.coff, n, (int32_t *)nullptr);
    if (n > 2) hipLaunchKernelGGL(pp_edges, dim3((n - 2 + 127) / 128), dim3(128), 0, st, A);
    hipLaunchKernelGGL(pp_edges_stop, dim3(1), dim3(64), 0, st, A);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.ncr, A.eoff, n, A.info + 0);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.nb, A.boff, n, (int32_t *)nullptr);
    hipLaunchKernelGGL(pp_emit, dim3((n + 127) / 128), dim3(128), 0, st, A);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.nf, A.foff, n, (int32_t *)nullptr);
    hipLaunchKernelGGL(pp_fwd_scatter, dim3((e_cap + 255) / 256), dim3(256), 0, st, A);
    hipLaunchKernelGGL(pp_fwd_sort, dim3((n + 255) / 256), dim3(256), 0, st, A);
    hipLaunchKernelGGL(pp_pass1, dim3((n + 255) / 256), dim3(256), 0, st, A);
    hipLaunchKernelGGL(pp_bound, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(pp_final_count, dim3((n + 255) / 256), dim3(256), 0, st, A);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.nb2, A.bwd_off, n, A.info + 1);
    hipLaunchKernelGGL(pp_scan, dim3(1), dim3(1024), 0, st, A.nf2, A.fwd_off, n, A.info + 2);
    hipLaunchKernelGGL(pp_final_lists, dim3((n + 255) / 256), dim3(256), 0, st, A);
    PPHIP(hipGetLastError());
    int32_t info[16];
    const double t2 = now();
    PPHIP(hipMemcpyAsync(info, A.info, sizeof(info), hipMemcpyDeviceToHost, st));
    PPHIP(hipStreamSynchronize(st));
    const double t3 = now();
    const int E = info[0], nbw = info[1], nfw = info[2];
    if (E < 0 || E > e_cap || nbw < 0 || nbw > E || nfw < 0 || nfw > E) return false;
    pd->n_edges = E;
    SeqGraph g;
    g.sym_width = left.sym_width;
    g.state.resize(n); g.site_type.resize(n); g.path_state.resize(n); g.child_l.resize(n); g.child_r.resize(n);
    g.count_since_used.resize(n); g.dist_since_used.resize(n); g.ambiguous.resize(n);
    g.e_start.resize(E); g.e_end.resize(E); g.e_w.resize(E); g.e_logw.resize(E); g.e_count_since_used.resize(E);
    g.e_count_as_skipped.resize(E); g.e_dist_since_used.resize(E); g.e_used.assign(E, 0);
    g.bwd_off.resize(n + 1); g.fwd_off.resize(n + 1); g.bwd_eid.resize(nbw); g.bwd_src.resize(nbw); g.bwd_logw.resize(nbw); g.fwd_eid.resize(nfw);
    auto down = [&](void *dst, const void *src, size_t bytes) { return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess; };
    const bool ok = down(g.state.data(), pd->state, 4 * (size_t)n) && down(g.site_type.data(), pd->site_type, n) && down(g.path_state.data(), pd->path_state, n) &&
                    down(g.child_l.data(), pd->child_l, 4 * (size_t)n) && down(g.child_r.data(), pd->child_r, 4 * (size_t)n) &&
                    down(g.count_since_used.data(), pd->count_since_used, 4 * (size_t)n) && down(g.dist_since_used.data(), pd->dist_since_used, 4 * (size_t)n) &&
                    down(g.ambiguous.data(), pd->ambiguous, n) &&
                    down(g.e_start.data(), pd->e_start, 4 * (size_t)E) && down(g.e_end.data(), pd->e_end, 4 * (size_t)E) && down(g.e_w.data(), pd->e_w, 4 * (size_t)E) &&
                    down(g.e_logw.data(), pd->e_logw, 4 * (size_t)E) && down(g.e_count_since_used.data(), pd->e_cnt, 4 * (size_t)E) &&
                    down(g.e_count_as_skipped.data(), pd->e_cas, 4 * (size_t)E) && down(g.e_dist_since_used.data(), pd->e_dist, 4 * (size_t)E) &&
                    down(g.bwd_off.data(), pd->bwd_off, 4 * (size_t)(n + 1)) && down(g.fwd_off.data(), pd->fwd_off, 4 * (size_t)(n + 1)) &&
                    down(g.bwd_eid.data(), pd->bwd_eid, 4 * (size_t)nbw) && down(g.bwd_src.data(), pd->bwd_src, 4 * (size_t)nbw) &&
                    down(g.bwd_logw.data(), pd->bwd_logw, 4 * (size_t)nbw) && down(g.fwd_eid.data(), pd->fwd_eid, 4 * (size_t)nfw);
    if (!ok) { (void)hipGetLastError(); return false; }
    PPHIP(hipStreamSynchronize(st));
    if (verbose)
        std::fprintf(stderr, "pagan_dp: parent of %d sites on device %d: children resident %.2f ms (%d %d), launches %.2f ms, kernels %.2f ms, download %.2f ms\n",
                     n, device, 1e3 * (t1 - t0), (int)l_was, (int)r_was, 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (now() - t3));
    // the host's logf is the authority for a weight's logarithm (the table covers what the rules produce; a weight outside
    // it is counted, and put right here and on the device)
    int patched = 0;
    if (info[6] != 0) {
        for (int e = 0; e < E; ++e) {
            if (g.e_w[e] == 1.0f) continue;
            const float lw = std::log(g.e_w[e]);
            if (std::memcmp(&lw, &g.e_logw[e], 4) != 0) { g.e_logw[e] = lw; ++patched; }
        }
        for (int k = 0; k < nbw; ++k) g.bwd_logw[k] = g.e_logw[g.bwd_eid[k]];
        if (patched) {
            PPHIP(hipMemcpyAsync(pd->e_logw, g.e_logw.data(), 4 * (size_t)E, hipMemcpyHostToDevice, st));
            PPHIP(hipMemcpyAsync(pd->bwd_logw, g.bwd_logw.data(), 4 * (size_t)nbw, hipMemcpyHostToDevice, st));
            PPHIP(hipStreamSynchronize(st));
        }
    }
    if (info_out) {
        info_out->runs = info[3]; info_out->rounds = info[4]; info_out->deleted_sites = info[5];
        info_out->weights_outside_table = info[6]; info_out->log_weights_patched = patched;
    }
    g.dev = pd;
    *out = std::move(g);
    return true;
}

} // namespace pagan
