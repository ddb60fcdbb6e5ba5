// host_model.h -- per-alignment scoring model (the reference's Evol_model) for DNA.
//
// DnaModelFactory restates Model_factory::dna_model (src/utils/model_factory.cpp:1299-1474) and
// Model_factory::alignment_model (model_factory.cpp:1871-2016): Q from base frequencies
// (kappa 2, rho 1, float arithmetic as written there), P(t) = U exp(t Lambda) V, log-odds
// table with the 15-letter ambiguity extension, and the indel parameters.  The symmetric
// eigenproblem is solved with cyclic Jacobi rotations instead of the reference's PAML
// Householder/QL routine (src/utils/eigen.cpp:132-330), so P(t) can differ from the
// reference's in the last bits of a double; the table is an INPUT of the aligner ABI.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pagan_dp.h"

namespace pagan {

struct EvolModel {
    int S = 0, char_as = 0;
    std::vector<float> log_score;        // [a + b*S]  (Evol_model::log_score, evol_model.h:87)
    float log_gap_open = 0, log_gap_ext = 0, log_gap_end_ext = 0, log_non_gap = 0;
    pagan_model view() const {
        pagan_model m;
        m.n_states = S; m.log_score = log_score.data();
        m.log_gap_open = log_gap_open; m.log_gap_ext = log_gap_ext;
        m.log_gap_end_ext = log_gap_end_ext; m.log_non_gap = log_non_gap;
        return m;
    }
};

struct DnaModelFactory {
    static const char *full_alphabet() { return "ACGTRYMKWSBDHVN"; }   // model_factory.cpp:103
    float ins_rate = 0.01f, del_rate = 0.01f, ext_prob = 0.8f, end_ext_prob = 0.95f;   // :1303-1306
    double pi[4];
    double U[16], V[16], root[4];
    std::vector<int32_t> parsimony;      // 15x15, [i + j*15] (model_factory.cpp:147-227)

    // Empirical base frequencies the way Fasta_reader::check_alphabet counts them
    // (src/utils/fasta_reader.cpp:1196-1255): float counters over all input sequences.
    static void base_frequencies(const std::vector<std::string> &seqs, float out[4]);
    void init(const float base_freq[4], float kappa = 2.0f, float rho = 1.0f);
    EvolModel alignment_model(double distance) const;
};

} // namespace pagan
