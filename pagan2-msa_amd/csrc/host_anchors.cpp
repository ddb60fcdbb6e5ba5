// host_anchors.cpp -- see host_anchors.h.
//
// The reference sorts char* suffix pointers of two NUL-terminated copies with qsort+strcmp
// (glibc's qsort is a stable merge sort here) and reports every adjacent cross-string pair
// with a common prefix >= min_length (find_anchors.cpp:66-85).  That order is the suffix
// array of  a + '\0' + b + '\1'  over an alphabet where both sentinels sort below every
// residue and '\0' < '\1': a shorter suffix sorts first, and of two identical suffixes the
// one from `a` comes first, which is what the stable sort of [a-suffixes..., b-suffixes...]
// yields.  So the suffix array is built directly (prefix doubling with counting sorts,
// O(n log maxLCP)) and the common prefixes come from Kasai's LCP pass instead of strcmp
// walks; the hit list is identical and in identical order.
#include "host_anchors.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace pagan {

namespace {

// Suffix array of `t` (values < sigma), prefix doubling.
void suffix_array(const std::vector<int> &t, int sigma, std::vector<int> *sa_out, std::vector<int> *rank_out) {
    const int n = (int)t.size();
    std::vector<int> sa(n), rk(n), tmp(n), cnt(std::max(sigma, n) + 1);
    // initial sort by first symbol
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int i = 0; i < n; ++i) cnt[t[i] + 1]++;
    for (int c = 0; c < sigma; ++c) cnt[c + 1] += cnt[c];
    for (int i = 0; i < n; ++i) sa[cnt[t[i]]++] = i;
    rk[sa[0]] = 0;
    int classes = 1;
    for (int i = 1; i < n; ++i) { if (t[sa[i]] != t[sa[i - 1]]) classes++; rk[sa[i]] = classes - 1; }
    std::vector<int> sa2(n);
    for (int k = 1; classes < n; k <<= 1) {
        // order by second key: suffixes i with i+k >= n come first (empty second key), then by sa order
        int p = 0;
        for (int i = n - k; i < n; ++i) sa2[p++] = i;
        for (int i = 0; i < n; ++i) if (sa[i] >= k) sa2[p++] = sa[i] - k;
        // stable counting sort by first key
        std::fill(cnt.begin(), cnt.begin() + classes + 1, 0);
        for (int i = 0; i < n; ++i) cnt[rk[i] + 1]++;
        for (int c = 0; c < classes; ++c) cnt[c + 1] += cnt[c];
        for (int i = 0; i < n; ++i) sa[cnt[rk[sa2[i]]]++] = sa2[i];
        tmp[sa[0]] = 0;
        int nc = 1;
        for (int i = 1; i < n; ++i) {
            const int a = sa[i - 1], b = sa[i];
            const int ra2 = a + k < n ? rk[a + k] : -1, rb2 = b + k < n ? rk[b + k] : -1;
            if (rk[a] != rk[b] || ra2 != rb2) nc++;
            tmp[b] = nc - 1;
        }
        rk.swap(tmp);
        classes = nc;
    }
    *sa_out = std::move(sa);
    *rank_out = std::move(rk);
}

// Overlap filter shared by find_long_substrings (:89-126) and check_hits_order_conflict (:232-278):
// walk the hits in order, drop one that touches an already covered site of either string.
void drop_overlapping(std::vector<Hit> *hits, int len1, int len2) {
    std::vector<uint8_t> h1(len1, 0), h2(len2, 0);
    size_t out = 0;
    for (size_t k = 0; k < hits->size(); ++k) {
        const Hit h = (*hits)[k];
        bool overlap = false;
        for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; ++i, ++j)
            if (h1[i] || h2[j]) { overlap = true; break; }
        if (overlap) continue;
        for (int i = h.s1, j = h.s2; i < h.s1 + h.len && j < h.s2 + h.len; ++i, ++j) { h1[i] = 1; h2[j] = 1; }
        (*hits)[out++] = h;
    }
    hits->resize(out);
}

} // namespace

// PAGAN_ANCHORS=host keeps the finder on the host; otherwise the suffix array is built on the calling thread's device
// (dp_anchors.hip) and the host's builder below only runs where there is no device
std::atomic<long long> device_finder_calls{0};                  // (diagnostic: pagan_anchors_device_calls)
// The device the calling thread's finders run on.  The tree walk prepares a unit's nodes on fresh threads, and a new thread's
// HIP device is 0 whatever device the unit belongs to: run_unit says which one it is (-1: whatever is current).
static thread_local int tl_anchor_device = -1;
void set_anchor_device(int device) { tl_anchor_device = device; }
int anchor_device() { return tl_anchor_device; }
static bool anchors_on_device() {
    const char *e = std::getenv("PAGAN_ANCHORS");
    return !(e && std::strcmp(e, "host") == 0);
}

void prefix_hits(const std::string &a, const std::string &b, int min_length, std::vector<Hit> *hits) {
    const int len1 = (int)a.size(), len2 = (int)b.size();
    // The device's finder wins on long sequences (2 x 100 kb: 2.9 ms against 11 ms on one host thread) and loses on short ones
    // (launch overheads: 2 x 3 kb 0.75 against 0.22 ms); a wide tree level prepares its nodes on as many host threads at once,
    // which then beat a device they would have to share: at most four finders on the device at a time.
    struct InFlight {
        std::atomic<int> &c; const bool ok;
        explicit InFlight(std::atomic<int> &c_) : c(c_), ok(c_.fetch_add(1) < 4) {}
        ~InFlight() { c.fetch_sub(1); }
    };
    static std::atomic<int> on_device[64];                         // per device: the cap is about sharing ONE device
    bool done = false;
    if (anchors_on_device() && len1 + len2 >= 16384) {
        const int dev = anchor_device();
        InFlight slot(on_device[(dev < 0 ? 0 : dev) & 63]);
        done = slot.ok && prefix_hits_device(a, b, min_length, hits, dev);
    }
    if (done) {
        device_finder_calls.fetch_add(1);
        std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.len > q.len; });   // :87
        drop_overlapping(hits, len1, len2);
        return;
    }
    hits->clear();
    const int n = len1 + len2 + 2;
    std::vector<int> t(n);
    for (int i = 0; i < len1; ++i) t[i] = (unsigned char)a[i] + 2;
    t[len1] = 0;
    for (int i = 0; i < len2; ++i) t[len1 + 1 + i] = (unsigned char)b[i] + 2;
    t[n - 1] = 1;
    std::vector<int> sa, rk;
    suffix_array(t, 258, &sa, &rk);
    // Kasai: lcp[r] = common prefix of suffixes at ranks r-1 and r.  The sentinels are unique,
    // so a common prefix never runs through one -- it is the strcmp prefix of the two strings.
    std::vector<int> lcp(n, 0);
    for (int i = 0, h = 0; i < n; ++i) {
        if (rk[i] == 0) { h = 0; continue; }
        const int j = sa[rk[i] - 1];
        while (i + h < n && j + h < n && t[i + h] == t[j + h] && t[i + h] > 1) ++h;
        lcp[rk[i]] = h;
        if (h > 0) --h;
    }
    // The two sentinel suffixes sort first and are not part of the reference's pointer array.
    for (int r = 1; r < n; ++r) {
        const int p = sa[r - 1], q = sa[r];
        if (p == len1 || p == n - 1 || q == len1 || q == n - 1) continue;
        const bool p1 = p < len1, q1 = q < len1;
        if (p1 == q1) continue;                                   // different_strings, find_anchors.h:107-115
        if (lcp[r] < min_length) continue;
        Hit h;
        h.s1 = p1 ? p : q;
        h.s2 = (p1 ? q : p) - (len1 + 1);
        h.len = lcp[r]; h.score = lcp[r];
        hits->push_back(h);
    }
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.len > q.len; });   // :87
    drop_overlapping(hits, len1, len2);
}

void resolve_conflicts(int len1, int len2, int trim, std::vector<Hit> *hits) {
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.score > q.score; });   // :230
    for (Hit &h : *hits) h.len -= trim * 2;          // :252; the start shifts at :248-251 are no-ops
    drop_overlapping(hits, len1, len2);
    std::sort(hits->begin(), hits->end(), [](Hit p, Hit q) { return p.s1 == q.s1 ? p.s2 < q.s2 : p.s1 < q.s1; });   // :280
    // :282-304: neighbours out of order in the second string -> drop the lower-scoring one and
    // re-test the survivor against the next hit.  `kept` is the hits before it1, in order.
    std::vector<Hit> kept;
    kept.reserve(hits->size());
    size_t k = 0;
    const size_t n = hits->size();
    if (n == 0) return;
    Hit cur = (*hits)[k++];
    while (k < n) {
        const Hit nxt = (*hits)[k];
        if (cur.s2 > nxt.s2) {
            if (cur.score < nxt.score) cur = nxt;    // erase it1; it1 now names the old it2
            ++k;                                     // either way the pair shrinks to one hit
            continue;
        }
        kept.push_back(cur);
        cur = nxt;
        ++k;
    }
    kept.push_back(cur);
    hits->swap(kept);
}

void hits_to_band(const std::vector<Hit> &hits, const std::string &str1, const std::string &str2, int width,
                  std::vector<int32_t> *upper, std::vector<int32_t> *lower) {
    const int length1 = (int)str1.size(), length2 = (int)str2.size();
    std::vector<int> index1, index2;
    index1.reserve(length1); index2.reserve(length2);
    for (int i = 0; i < length1; ++i) if (str1[i] != '-') index1.push_back(i + 1);
    for (int i = 0; i < length2; ++i) if (str2[i] != '-') index2.push_back(i + 1);
    std::vector<int> diag(length1 + 1, -1);
    for (const Hit &h : hits) {
        int i = 0;
        for (; i < h.len; ++i) diag[index1[h.s1 + i]] = index2[h.s2 + i];
        if (h.s1 + i < (int)index1.size() && index1[h.s1 + i] < (int)diag.size()) diag[index1[h.s1 + i]] = -2;
    }
    upper->assign(length1 + 1, 0);
    lower->assign(length1 + 1, 0);
    int y1 = 0, y2 = 0, prev_y = 0, m_count = 0;
    for (int i = 0; i <= length1; ++i) {                                   // :373-408
        if (i >= width && diag[i - width] >= 0) y1 = diag[i - width];
        if (diag[i] >= 0) y2 = diag[i] - width;
        const bool run = diag[i] >= 0 && i > 0 && diag[i - 1] + 1 == diag[i];
        if (run) m_count++; else if (diag[i] == -2) m_count = 0;
        int y = std::max(std::min(y1, y2), 0);
        if (run && m_count >= width) prev_y = y;
        (*upper)[i] = std::max(std::min(y, prev_y), 0);
    }
    y1 = y2 = prev_y = length2; m_count = 0;
    for (int i = length1; i >= 0; --i) {                                   // :416-447
        if (i <= length1 - width && diag[i + width] >= 0) y1 = diag[i + width];
        if (diag[i] >= 0) y2 = diag[i] + width;
        const bool run = diag[i] >= 0 && i < length1 && diag[i + 1] - 1 == diag[i];
        if (run) m_count++; else if (diag[i] == -2) m_count = 0;
        int y = std::min(std::max(y1, y2), length2);
        if (run && m_count >= width) prev_y = y;
        (*lower)[i] = std::min(std::max(y, prev_y), length2);
    }
}

// ---- Find_anchors::eliminate_bad_hits and its predicates (find_anchors.cpp:497-632) ----
namespace {
inline int end1(const Hit &h) { return h.s1 + h.len; }
inline int end2(const Hit &h) { return h.s2 + h.len; }
// length of `h`'s head lying over `o`'s tail on either axis (overlapsAtBegin, :552-565)
int head_overlap(const Hit &h, const Hit &o) {
    int ov = 0;
    if (h.s1 >= o.s1 && end1(h) > end1(o)) ov = std::max(ov, end1(o) - h.s1);
    if (h.s2 >= o.s2 && end2(h) > end2(o)) ov = std::max(ov, end2(o) - h.s2);
    return std::max(0, ov);
}
unsigned diagonal_distance(const Hit &h, const Hit &o) { return (unsigned)std::abs((o.s1 - o.s2) - (h.s1 - h.s2)); }   // :573-575
bool crosses(const Hit &h, const Hit &o) {                                                                              // probaplyBadHit, :583-591
    if (h.s1 < o.s1 && h.s2 > o.s2 && end1(h) < end1(o)) return true;
    if (h.s1 > o.s1 && h.s2 < o.s2 && end2(h) < end2(o)) return true;
    return false;
}
bool inside(const Hit &h, const Hit &o) {                                                                               // totallyOverlappingHit, :598-606
    return (h.s1 >= o.s1 && end1(h) <= end1(o)) || (h.s2 >= o.s2 && end2(h) <= end2(o));
}
} // namespace

void drop_bad_hits(std::vector<Hit> *hits, unsigned max_inside, unsigned max_partly) {
    // hits are judged in input order against the hits accepted so far as "good"; a hit that crosses or lies inside
    // a good hit survives only close to its diagonal (and then is kept without becoming a yardstick itself)
    std::vector<Hit> good, out;
    out.reserve(hits->size());
    for (const Hit &h : *hits) {
        bool bad = false, tolerated = false;
        for (const Hit &g : good) {
            if (crosses(h, g) || inside(h, g)) {
                if (diagonal_distance(h, g) > max_inside) { bad = true; break; }
                tolerated = true;
            } else if (head_overlap(h, g) || head_overlap(g, h)) {                    // partlyOverlappingHit, :616-624
                if (diagonal_distance(h, g) > max_partly) { bad = true; break; }
            }
        }
        if (bad) continue;
        if (!tolerated) good.push_back(h);
        out.push_back(h);
    }
    hits->swap(out);
}

void hits_to_band_overlapping(const std::vector<Hit> &hits, const std::string &g1, const std::string &g2, int width,
                              std::vector<int32_t> *upper, std::vector<int32_t> *lower, std::vector<TunnelBlock> *blocks) {
    const int l1 = (int)g1.size(), l2 = (int)g2.size();
    std::vector<int> pos1, pos2;                                   // 1-based column of every residue in the gapped strings
    for (int i = 0; i < l1; ++i) if (g1[i] != '-') pos1.push_back(i + 1);
    for (int i = 0; i < l2; ++i) if (g2[i] != '-') pos2.push_back(i + 1);
    const int floor_y = 0, top_y = l2;
    std::vector<int> lo(l1 + 1, top_y + 1), hi(l1 + 1, floor_y - 1);    // per row: lowest / highest anchored column; unset
    for (const Hit &h : hits)                                           // :683-692
        for (int a = 0; a < h.len; ++a) {
            const int x = pos1.at(h.s1 + a), y = pos2.at(h.s2 + a);
            if (y < lo[x]) lo[x] = std::max(y, floor_y);
            if (y > hi[x]) hi[x] = std::min(y, top_y);
        }
    {   // monotone: the upper envelope never falls, the lower never rises (:699-717); unset rows are skipped
        int run = hi[0];
        for (int i = 0; i <= l1; ++i) if (hi[i] > floor_y) { if (hi[i] < run) hi[i] = run; run = hi[i]; }
        run = lo[l1];
        for (int i = l1; i >= 0; --i) if (lo[i] < top_y) { if (lo[i] > run) lo[i] = run; run = lo[i]; }
    }
    {   // empty blocks between anchored stretches (:720-748)
        TunnelBlock cur;
        cur.sx = 0; cur.sy = 0;
        for (int i = 1; i <= l1; ++i) {
            const bool here = hi[i] >= floor_y, before = hi[i - 1] >= floor_y;
            if (before && !here) { cur.sx = i; cur.sy = hi[i - 1]; }
            else if (here && !before) {
                if (lo[i] > cur.sy) { cur.ex = i; cur.ey = lo[i]; if (cur.size() > 10) blocks->push_back(cur); }
            } else if (i == l1 && !here) {
                if (top_y > cur.sy) { cur.ex = i; cur.ey = top_y; if (cur.size() > 10) blocks->push_back(cur); }
            }
        }
        // ascending by size (std::sort over Tunnel_block::operator<, :750); stable here so that equal sizes keep their order
        std::stable_sort(blocks->begin(), blocks->end(), [](const TunnelBlock &a, const TunnelBlock &b) { return a.size() < b.size(); });
    }
    {   // unset rows take their neighbour's bound (:757-771), corners pinned (:774-775)
        int run = floor_y;
        for (int i = 0; i <= l1; ++i) { if (lo[i] >= top_y) lo[i] = run; run = lo[i]; }
        run = top_y;
        for (int i = l1; i >= 0; --i) { if (hi[i] <= floor_y) hi[i] = run; run = hi[i]; }
        lo[0] = floor_y; hi[l1] = top_y;
    }
    for (int i = 0; i <= l1; ++i) if (hi[i] >= floor_y) hi[i] = std::min(top_y, hi[i] + width);      // :780-789
    for (int i = 0; i <= l1; ++i) if (lo[i] <= top_y) lo[i] = std::max(floor_y, lo[i] - width);
    {   // the same margin along the other axis (:793-827): where a bound jumps by more than one, the rows before (after) the
        // jump are lifted -- to the jump's level across a gap, along a unit-slope ramp otherwise
        std::vector<std::pair<int, bool>> jumps;
        for (int i = 1; i <= l1; ++i) {
            if ((i + 1 > l1 || hi[i] == hi[i + 1]) && hi[i - 1] < hi[i] - 1) jumps.emplace_back(i, true);
            else if (hi[i - 1] < hi[i] - 1) jumps.emplace_back(i, false);
        }
        for (const auto &jp : jumps) {
            const int i = jp.first;
            for (int x = i - 1; x >= i - width && x >= 0 && hi[x] >= floor_y; --x)
                hi[x] = std::max(hi[x], jp.second ? hi[i] : hi[x + 1] - 1);
        }
        jumps.clear();
        for (int i = l1 - 1; i >= 0; --i) {
            if ((i - 1 < 0 || lo[i] == lo[i - 1]) && lo[i + 1] > lo[i] + 1) jumps.emplace_back(i, true);
            else if (lo[i + 1] > lo[i] + 1) jumps.emplace_back(i, false);
        }
        for (const auto &jp : jumps) {
            const int i = jp.first;
            for (int x = i + 1; x <= i + width && x <= l1 && lo[x] <= top_y; ++x)
                lo[x] = std::min(lo[x], jp.second ? lo[i] : lo[x - 1] + 1);
        }
    }
    upper->assign(lo.begin(), lo.end());                                                              // :839-842
    lower->assign(hi.begin(), hi.end());
}

bool force_gap(std::vector<int32_t> *upper, std::vector<int32_t> *lower, std::vector<TunnelBlock> *blocks, int min_size,
               int width, bool wide) {
    if (blocks->empty() || blocks->back().size() < min_size) return false;                            // :481-486
    const TunnelBlock b = blocks->back();
    std::vector<int32_t> &up = *upper, &lo = *lower;
    const int last = (int)lo.size() - 1;
    auto pull_down_before = [&](int from) {                    // keep the lower bound monotone below the block (:499-506, 519-526)
        for (int i = from; i >= 0; --i) { if (lo.at(i) > lo.at(i + 1)) lo.at(i) = lo.at(i + 1); else break; }
    };
    if (wide) {                                                                                       // :490-506
        for (int i = b.sx; i < b.ex - width; ++i) lo.at(i) = b.sy + width;
        pull_down_before(b.sx - 1);
    } else {                                                                                          // :508-541
        int a = 0;
        for (int i = b.sx; i < b.ex; ++i, ++a) { lo.at(i) = b.sy; up.at(i) = std::min(b.sy, up.at(i) + a); }
        up.at(b.ex) = b.sy;
        pull_down_before(b.sx - 1);
        const int last_i = std::min(b.ex + width + 1, last);
        int back = 0;
        for (int i = last_i; i >= b.ex + 1; --i, ++back) up.at(i) = std::max(up.at(last_i) - back, b.sy);
    }
    blocks->pop_back();
    return true;
}

int define_tunnel(const std::string &s1, const std::string &s2, const std::string &g1, const std::string &g2,
                  const AnchorSettings &as, std::vector<int32_t> *upper, std::vector<int32_t> *lower) {
    std::vector<Hit> hits;
    prefix_hits(s1, s2, as.prefix_hit_length, &hits);
    // check_hits_order_conflict receives the GAPPED strings' lengths (viterbi_alignment.cpp:138-161)
    resolve_conflicts((int)g1.size(), (int)g2.size(), as.hit_trim, &hits);
    hits_to_band(hits, g1, g2, as.offset, upper, lower);
    return (int)hits.size();
}

} // namespace pagan
