// host_anchors.h -- anchors -> band ("tunnel") for one pairwise alignment.
//
// Restates the in-tree anchor path of Viterbi_alignment::define_tunnel
// (src/main/viterbi_alignment.cpp:44-185) for --use-prefix-anchors:
//   prefix_hits      <- Find_anchors::find_long_substrings      src/utils/find_anchors.cpp:35-127
//   resolve_conflicts<- Find_anchors::check_hits_order_conflict find_anchors.cpp:225-317
//   hits_to_band     <- Find_anchors::define_tunnel             find_anchors.cpp:320-447
// The default anchor source of the reference (NCBI BLAST) is outside its tree and cannot be
// reproduced; the band arrays are an INPUT of the aligner ABI for that reason.
#pragma once
#include <cstdint>
#include <atomic>
#include <string>
#include <vector>

namespace pagan {

struct Hit { int s1, s2, len, score; };

struct AnchorSettings {
    int prefix_hit_length = 30;   // --prefix-hit-length  (settings.cpp:160)
    int hit_trim = 5;             // --exonerate-hit-trim (settings.cpp:155)
    int offset = 15;              // --anchors-offset     (settings.cpp:157)
};

void prefix_hits(const std::string &a, const std::string &b, int min_length, std::vector<Hit> *hits);
// the same list before its sort by length and overlap filter, from a suffix array built on the device (dp_anchors.hip);
// false where there is no device
extern std::atomic<long long> device_finder_calls;    // how often prefix_hits took the device's finder
void anchors_release_cache();                          // frees the device finder's idle scratch (pagan_dp_release_cache)
// device < 0: the calling thread's current device
bool prefix_hits_device(const std::string &a, const std::string &b, int min_length, std::vector<Hit> *hits, int device = -1);
// which device prefix_hits() on this thread builds its suffix arrays on (host_tree.cpp: run_unit's device)
void set_anchor_device(int device);
int anchor_device();
void resolve_conflicts(int len1, int len2, int trim, std::vector<Hit> *hits);
void hits_to_band(const std::vector<Hit> &hits, const std::string &gapped1, const std::string &gapped2, int width,
                  std::vector<int32_t> *upper, std::vector<int32_t> *lower);

// ---- the tunnel from possibly overlapping hits (the reference's BLAST branch) and the --force-gap rescue ----
//   drop_bad_hits       <- Find_anchors::eliminate_bad_hits                    find_anchors.cpp:497-545 (+ predicates :552-632)
//   hits_to_band_overlapping <- Find_anchors::define_tunnel_with_overlapping_hits   find_anchors.cpp:643-843
//   force_gap           <- Viterbi_alignment::replace_largest_tunnel_block_with_gap_tunnel  viterbi_alignment.cpp:467-553
// The hits themselves come from NCBI BLAST in the reference (outside its tree: unpinned); everything from the hit
// list onwards is restated here.
struct TunnelBlock {                   // find_anchors.h:51-70: an empty rectangle between two anchored stretches
    int sx = -1, sy = -1, ex = -1, ey = -1;
    long long size() const { return (long long)(ex - sx) * (long long)(ey - sy); }
};
void drop_bad_hits(std::vector<Hit> *hits, unsigned max_dist_inside, unsigned max_dist_partly);
void hits_to_band_overlapping(const std::vector<Hit> &hits, const std::string &gapped1, const std::string &gapped2, int width,
                              std::vector<int32_t> *upper, std::vector<int32_t> *lower, std::vector<TunnelBlock> *empty_blocks);
// Replaces the largest empty block (blocks sorted ascending by size; the last one) by a gap-shaped tunnel.  Returns false
// when there is no block of at least `min_size` cells left.
bool force_gap(std::vector<int32_t> *upper, std::vector<int32_t> *lower, std::vector<TunnelBlock> *blocks, int min_size,
               int width, bool wide_tunnel);

// define_tunnel end to end: ungapped strings for the hits, gapped strings for the band.
int define_tunnel(const std::string &s1, const std::string &s2, const std::string &g1, const std::string &g2,
                  const AnchorSettings &as, std::vector<int32_t> *upper, std::vector<int32_t> *lower);

} // namespace pagan
