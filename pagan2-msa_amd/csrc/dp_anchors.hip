// dp_anchors.hip -- prefix anchors on the device: the exact shared substrings that
// Find_anchors::find_long_substrings reports (src/utils/find_anchors.cpp:35-127), found with a suffix array built on the
// GPU.
//
// The reference sorts the suffix pointers of two NUL-terminated copies (qsort + strcmp: a stable merge sort in glibc) and
// reports every adjacent cross-string pair with a common prefix >= min_length (:66-85).  host_anchors.cpp shows that this
// order is the suffix array of  a + '\0' + b + '\1'  (both sentinels below every residue, '\0' < '\1': every suffix is a
// different string, so the order is unique -- no tie for a sort to break) and builds it on the host.  Here:
//   * prefix doubling: round r sorts the suffixes by (rank of the first 2^r symbols, rank of the next 2^r) -- one 40-bit
//     key per suffix, a library radix sort (rocPRIM) of n keys with their indices -- and renumbers: a flag where a key differs
//     from its predecessor, an inclusive scan of the flags.  Rounds until all n ranks differ (log2 of the longest repeat);
//     the rank arrays of all rounds are kept;
//   * the common prefix of two suffixes from those arrays, longest round first: ranks of round r equal <=> 2^r symbols equal
//     (the unique sentinels end every comparison where strcmp ends it);
//   * the adjacent cross-string pairs with a common prefix >= min_length are selected in suffix-array order (a flag per
//     rank, a library select) and written out as (start in a, start in b, length).
// What follows in the reference -- the sort by length, the overlap filter (:87-126), check_hits_order_conflict and
// define_tunnel -- stays with host_anchors.cpp: the hit list that enters it is the same, element for element
// (tests/test_anchors_gpu.py compares the two on homologous sequences up to 2 x 100 kb).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "host_anchors.h"

namespace pagan {

namespace {

#define HIPA(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { if (std::getenv("PAGAN_DP_VERBOSE")) std::fprintf(stderr, "pagan anchors: %s -> %s\n", #x, hipGetErrorString(e_)); return false; } } while (0)

__global__ void pa_symbols(const char *a, int len1, const char *b, int len2, int *rk0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = len1 + len2 + 2;
    if (i >= n) return;
    int v;
    if (i < len1) v = (unsigned char)a[i] + 2;
    else if (i == len1) v = 0;
    else if (i < n - 1) v = (unsigned char)b[i - len1 - 1] + 2;
    else v = 1;
    rk0[i] = v;
}

// key of suffix i in a round with half-length k: (rank of its first k symbols, rank of the next k; 0 where there are none)
__global__ void pa_keys(const int *rk, int n, int k, unsigned long long *key, int *idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long hi = (unsigned long long)(rk[i] + 1), lo = i + k < n ? (unsigned long long)(rk[i + k] + 1) : 0ull;
    key[i] = (hi << 20) | lo;
    idx[i] = i;
}

__global__ void pa_flags(const unsigned long long *key_sorted, int n, int *flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    flag[r] = r > 0 && key_sorted[r] != key_sorted[r - 1] ? 1 : 0;
}

__global__ void pa_ranks(const int *sa, const int *flag_scan, int n, int *rk_new) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    rk_new[sa[r]] = flag_scan[r];
}

struct Rounds { const int *rk[24]; int n_rounds; };

// common prefix of the suffixes at ranks r-1 and r, for the adjacent pairs that lie in different strings; 1 where it
// reaches min_length
__global__ void pa_pairs(const int *sa, int n, int len1, Rounds R, int min_length, int *lcp, unsigned char *hit) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    unsigned char h = 0;
    int len = 0;
    if (r > 0) {
        const int p = sa[r - 1], q = sa[r];
        const bool sentinel = p == len1 || p == n - 1 || q == len1 || q == n - 1;
        if (!sentinel && (p < len1) != (q < len1)) {
            int i = p, j = q;
            for (int t = R.n_rounds - 1; t >= 0; --t) {
                const int step = 1 << t;
                if (i + step <= n && j + step <= n && R.rk[t][i] == R.rk[t][j]) { i += step; j += step; len += step; if (i >= n || j >= n) break; }
            }
            h = len >= min_length;
        }
    }
    lcp[r] = len;
    hit[r] = h;
}

__global__ void pa_emit(const int *sel, int n_sel, const int *sa, const int *lcp, int len1, int *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sel) return;
    const int r = sel[k], p = sa[r - 1], q = sa[r];
    const bool p1 = p < len1;
    out[3 * k] = p1 ? p : q;
    out[3 * k + 1] = (p1 ? q : p) - (len1 + 1);
    out[3 * k + 2] = lcp[r];
}

// Device scratch and a stream per finder in flight, kept between calls in a small pool (the tree walk prepares a level's nodes
// on threads that live for that level only: per-thread storage would be allocated and freed -- a device synchronisation --
// once per node).  At most four are in flight (host_anchors.cpp), so at most four exist.
struct Scratch {
    int device = -1;
    size_t cap = 0;
    char *mem = nullptr;
    hipStream_t stream = nullptr;
};
struct ScratchPool {
    std::mutex m;
    std::vector<Scratch> idle;
    Scratch take(int device) {
        std::lock_guard<std::mutex> g(m);
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].device == device) { Scratch s = idle[k]; idle.erase(idle.begin() + k); return s; }
        Scratch s; s.device = device; return s;
    }
    void give(const Scratch &s) { std::lock_guard<std::mutex> g(m); idle.push_back(s); }
};
ScratchPool scratch_pool;
struct ScratchLease {
    Scratch s;
    explicit ScratchLease(int device) : s(scratch_pool.take(device)) {}
    ~ScratchLease() { scratch_pool.give(s); }
};

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

} // namespace

// The cross-string adjacent pairs of the suffix array with a common prefix >= min_length, in suffix-array order -- the list
// prefix_hits() builds before it sorts by length.  false: no device / a HIP error (the caller takes the host's finder).
// (pagan_dp_release_cache: the finders' idle scratch and streams go as well)
void anchors_release_cache() {
    std::vector<Scratch> all;
    { std::lock_guard<std::mutex> g(scratch_pool.m); all.swap(scratch_pool.idle); }
    for (Scratch &s : all) {
        (void)hipSetDevice(s.device);
        if (s.mem) (void)hipFree(s.mem);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
}

bool prefix_hits_device(const std::string &a, const std::string &b, int min_length, std::vector<Hit> *hits, int device) {
    const int len1 = (int)a.size(), len2 = (int)b.size(), n = len1 + len2 + 2;
    if (n >= (1 << 20) - 1 || min_length < 1) return false;               // (20-bit ranks in the sort key)
    // (a thread made for one level of the walk starts on device 0: the unit's device is handed down, host_tree.cpp)
    if (device >= 0) HIPA(hipSetDevice(device)); else HIPA(hipGetDevice(&device));
    ScratchLease lease(device);
    Scratch &S = lease.s;
    if (!S.stream) HIPA(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    hipStream_t st = S.stream;
    const int max_rounds = 21;
    // library scratch sizes
    size_t tmp_sort = 0, tmp_scan = 0, tmp_sel = 0;
    HIPA(rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr,
                                   (size_t)n, 0u, 40u, st));
    HIPA(rocprim::inclusive_scan(nullptr, tmp_scan, (int *)nullptr, (int *)nullptr, (size_t)n, rocprim::plus<int>(), st));
    HIPA(rocprim::select(nullptr, tmp_sel, rocprim::counting_iterator<int>(0), (unsigned char *)nullptr, (int *)nullptr, (int *)nullptr, (size_t)n, st));
    const size_t tmp = up256(std::max(tmp_sort, std::max(tmp_scan, tmp_sel)));
    const size_t need = up256(len1 + 1) + up256(len2 + 1) + 2 * up256(8 * (size_t)n) + (size_t)(6 + max_rounds) * up256(4 * (size_t)n) + up256(n) + tmp + 4096;
    if (S.cap < need) {
        if (S.mem) { (void)hipFree(S.mem); S.mem = nullptr; S.cap = 0; }
        HIPA(hipMalloc((void **)&S.mem, need));
        S.cap = need;
    }
    char *m = S.mem;
    auto take = [&](size_t bytes) { char *p = m; m += up256(bytes); return p; };
    char *d_a = take(len1 + 1), *d_b = take(len2 + 1);
    unsigned long long *key = (unsigned long long *)take(8 * (size_t)n), *key2 = (unsigned long long *)take(8 * (size_t)n);
    int *idx = (int *)take(4 * (size_t)n), *sa = (int *)take(4 * (size_t)n), *flag = (int *)take(4 * (size_t)n), *fscan = (int *)take(4 * (size_t)n);
    int *lcp = (int *)take(4 * (size_t)n), *sel = (int *)take(4 * (size_t)n);
    int *rk[max_rounds + 1];
    for (int t = 0; t <= max_rounds - 1; ++t) rk[t] = (int *)take(4 * (size_t)n);
    unsigned char *hit = (unsigned char *)take(n);
    int *d_count = (int *)take(256);
    void *d_tmp = take(tmp);
    if (len1) HIPA(hipMemcpyAsync(d_a, a.data(), len1, hipMemcpyHostToDevice, st));
    if (len2) HIPA(hipMemcpyAsync(d_b, b.data(), len2, hipMemcpyHostToDevice, st));
    const int B = 256, G = (n + B - 1) / B;
    hipLaunchKernelGGL(pa_symbols, dim3(G), dim3(B), 0, st, d_a, len1, d_b, len2, rk[0]);
    Rounds R;
    R.n_rounds = 1; R.rk[0] = rk[0];
    int classes = 0;
    for (int t = 0; t < max_rounds - 1; ++t) {
        const int k = 1 << t;
        // (round t sorts by 2^(t+1) symbols: rank arrays rk[t] (2^t symbols) -> rk[t+1])
        hipLaunchKernelGGL(pa_keys, dim3(G), dim3(B), 0, st, rk[t], n, k, key, idx);
        size_t ts = tmp;
        HIPA(rocprim::radix_sort_pairs(d_tmp, ts, key, key2, idx, sa, (size_t)n, 0u, 40u, st));
        hipLaunchKernelGGL(pa_flags, dim3(G), dim3(B), 0, st, key2, n, flag);
        ts = tmp;
        HIPA(rocprim::inclusive_scan(d_tmp, ts, flag, fscan, (size_t)n, rocprim::plus<int>(), st));
        hipLaunchKernelGGL(pa_ranks, dim3(G), dim3(B), 0, st, sa, fscan, n, rk[t + 1]);
        HIPA(hipMemcpyAsync(&classes, fscan + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        HIPA(hipStreamSynchronize(st));
        R.rk[R.n_rounds++] = rk[t + 1];
        if (classes + 1 == n) break;
        if (k >= n) break;
    }
    if (classes + 1 != n) return false;                                    // (cannot happen: every suffix is a different string)
    // round t's ranks tell 2^t symbols apart for t >= 1 only up to the sort's depth; rk[0] are the symbols themselves
    hipLaunchKernelGGL(pa_pairs, dim3(G), dim3(B), 0, st, sa, n, len1, R, min_length, lcp, hit);
    size_t ts = tmp;
    HIPA(rocprim::select(d_tmp, ts, rocprim::counting_iterator<int>(0), hit, sel, d_count, (size_t)n, st));
    int n_sel = 0;
    HIPA(hipMemcpyAsync(&n_sel, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPA(hipStreamSynchronize(st));
    hits->clear();
    if (n_sel > 0) {
        int *out = (int *)key;                                             // (the key array is free now: 3 ints per hit, n_sel <= n)
        hipLaunchKernelGGL(pa_emit, dim3((n_sel + B - 1) / B), dim3(B), 0, st, sel, n_sel, sa, lcp, len1, out);
        std::vector<int> h(3 * (size_t)n_sel);
        HIPA(hipMemcpyAsync(h.data(), out, h.size() * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPA(hipStreamSynchronize(st));
        hits->resize(n_sel);
        for (int k = 0; k < n_sel; ++k) { Hit x; x.s1 = h[3 * k]; x.s2 = h[3 * k + 1]; x.len = h[3 * k + 2]; x.score = x.len; (*hits)[k] = x; }
    }
    return true;
}

} // namespace pagan
