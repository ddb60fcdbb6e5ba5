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
void resolve_conflicts(int len1, int len2, int trim, std::vector<Hit> *hits);
void hits_to_band(const std::vector<Hit> &hits, const std::string &gapped1, const std::string &gapped2, int width,
                  std::vector<int32_t> *upper, std::vector<int32_t> *lower);

// define_tunnel end to end: ungapped strings for the hits, gapped strings for the band.
int define_tunnel(const std::string &s1, const std::string &s2, const std::string &g1, const std::string &g2,
                  const AnchorSettings &as, std::vector<int32_t> *upper, std::vector<int32_t> *lower);

} // namespace pagan
