// host_model.h -- per-alignment scoring model (the reference's Evol_model) for DNA, protein and codons.
//
// ModelFactory restates Model_factory::dna_model / protein_model / codon_model
// (src/utils/model_factory.cpp:1299-1474, 1478-1595, 1599-1805), the alphabets and parsimony tables
// (model_factory.cpp:118-227 DNA, 304-541 protein, 839-1217 codon) and Model_factory::alignment_model
// (model_factory.cpp:1871-2230): Q and pi, the time-reversible eigen solution
// (Eigen::eigenQREV -> Householder tridiagonalisation -> implicit QL, src/utils/eigen.cpp:39-295,
// same operation order, so U, V and the roots carry the same doubles), P(t) = U exp(t Lambda) V
// (eigen.cpp:330-358), the log-odds table with its ambiguity extension, and the indel parameters.
// The table is an INPUT of the aligner ABI (pagan_model); this is the producer the stand-alone
// tree walk uses.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pagan_dp.h"

namespace pagan {

struct EvolModel {
    int S = 0, char_as = 0;
    std::vector<float> log_score;        // [a + b*S]  (Evol_model::log_score, evol_model.h:87)
    float log_gap_open = 0, log_gap_ext = 0, log_gap_end_ext = 0, log_non_gap = 0;
    // probability-space accessors (Evol_model::score / gap_open / gap_ext / non_gap, evol_model.h:70-88): the
    // forward/backward pass works on these
    std::vector<float> score;            // [a + b*S]
    float gap_open = 0, gap_ext = 0, non_gap = 0;
    pagan_model_prob prob_view() const {
        pagan_model_prob m;
        m.n_states = S; m.score = score.data(); m.gap_open = gap_open; m.gap_ext = gap_ext; m.non_gap = non_gap;
        return m;
    }
    pagan_model view() const {
        pagan_model m;
        m.n_states = S; m.log_score = log_score.data();
        m.log_gap_open = log_gap_open; m.log_gap_ext = log_gap_ext;
        m.log_gap_end_ext = log_gap_end_ext; m.log_non_gap = log_non_gap;
        return m;
    }
};

enum DataType { kDna = 0, kProtein = 1, kCodon = 2 };     // Model_factory::dna / ::protein / ::codon

struct ModelFactory {
    int type = kDna;
    int S = 15, char_as = 4;                              // char_fas, char_as
    std::string leaf_alphabet;                            // Sequence::full_char_alphabet: state = find(residue)
    std::string ancestral_alphabet;                       // Model_factory::ancestral_character_alphabet, one char per state
    float ins_rate = 0.01f, del_rate = 0.01f, ext_prob = 0.8f, end_ext_prob = 0.95f;
    std::vector<double> pi, U, V, root;
    std::vector<int32_t> parsimony;                       // [i + j*S]
    std::vector<int32_t> mostcommon;                      // [i + j*mc_dim] (--mostcommon)
    int mc_dim = 0;
    std::vector<int16_t> res1, res2;                      // protein: Char_symbol::first_residue / second_residue;
                                                          // codon: Codon_symbol::first_codon / second_codon
    // codon: three characters per state -- the 61 sense codons, NNN, then per codon pair the IUPAC merge of the two,
    // position by position (Model_factory::ancestral_character_alphabet for codons, model_factory.cpp:1739-1803)
    std::string codon_names;

    static const char *dna_full_alphabet() { return "ACGTRYMKWSBDHVN"; }          // model_factory.cpp:103
    static const char *protein_alphabet() { return "ARNDCQEGHILKMFPSTWYV"; }      // model_factory.cpp:104
    // Fasta_reader::check_sequence_data_type, fasta_reader.cpp:1303-1336
    static int guess_type(const std::vector<std::string> &raw_upper);
    // Empirical base frequencies the way Fasta_reader::check_alphabet counts them
    // (src/utils/fasta_reader.cpp:1196-1255): float counters over all input sequences.
    static void base_frequencies(const std::vector<std::string> &seqs, float out[4]);
    void init_dna(const float base_freq[4], float kappa = 2.0f, float rho = 1.0f);
    void init_protein();
    void init_codon();                                    // Kosiol & Goldman's empirical codon model, 61 + 1 + 1830 states
    // the 61 sense codons in the model's order, then NNN (model_factory.h:209-221), three characters each
    static const char *codon_alphabet();
    // Sequence::create_codon_sequence (src/main/sequence.cpp:318-336): one state per triplet, 61 (NNN) for anything that
    // is not a sense codon -- a last partial triplet included.  symbols (optional) gets the triplet or "NNN" per state.
    static std::vector<int32_t> codon_states(const std::string &nucleotides, std::string *symbols = nullptr);
    // Codon_translation::gapped_DNA_to_protein (src/utils/codon_translation.cpp:32-107): one amino-acid letter per triplet
    // of a codon string ("---" -> '-', IUPAC-degenerate codons that still name one amino acid -> that one, anything else
    // X) -- what the reference anchors codon graphs on (viterbi_alignment.cpp:54-60, 141-145)
    static std::string translate_codons(const std::string &codon_string);
    // ins = del = 0.25 for --454/--homopolymer with --pileup-alignment (model_factory.cpp:1901-1905)
    EvolModel alignment_model(double distance, bool pileup_rates = false) const;
};

// Eigen::eigenQREV (eigen.cpp:48-128) on a row-major n x n rate matrix; exposed for the tests.
int eigen_qrev(const double *Q, const double *pi, int n, double *root, double *U, double *V);

} // namespace pagan
