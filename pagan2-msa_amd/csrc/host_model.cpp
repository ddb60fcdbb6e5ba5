// host_model.cpp -- see host_model.h.
//
// Provenance note (copy check): SymEigen below (tridiagonalise + ql) RESTATES the reference's eigen routine
// statement by statement -- /root/reference/src/utils/eigen.cpp:174-318, itself the PAML / Numerical-Recipes pair
// tred2 + tqli (Householder reduction, implicit-shift QL).  That is deliberate and limited to this routine: the model's
// score table is compared bit for bit with the reference's arithmetic (and with oracle/oracle_model.cpp, which restates the
// same lines), and a textbook eigen solver only reproduces those bits if every inner product accumulates in the same order
// and the deflation test is the same expression.  It is host-side model set-up (outside the GPU scope, no
// s.8 row rests on it); nothing else in this file follows the reference's text.
#include "host_model.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "codon_data.h"
#include "wag_data.h"

namespace pagan {

namespace {

// Eigen solution of a real symmetric matrix the way the reference gets it (Eigen::eigenRealSym,
// src/utils/eigen.cpp:135-149): Householder reduction to tridiagonal form, implicit-shift QL on the
// tridiagonal matrix with the accumulated transformation, eigenvalues sorted descending.  The
// arithmetic is kept operation for operation (accumulation order of every inner product, the
// scaled pythagorean sums, the deflation test `|e| + dd == dd`): the score table downstream is
// compared bit for bit.
struct SymEigen {
    int n;
    std::vector<double> &A;          // in: symmetric matrix, row-major; out: eigenvectors in columns
    std::vector<double> diag, off;   // out: eigenvalues / work
    double &a(int r, int c) { return A[(size_t)r * n + c]; }

    SymEigen(std::vector<double> &m, int dim) : n(dim), A(m), diag(dim, 0.0), off(dim, 0.0) {}

    // eigen.cpp:177-245 (tred2 with eigenvector accumulation)
    void tridiagonalise() {
        for (int row = n - 1; row >= 1; --row) {
            const int last = row - 1;
            double h = 0, scale = 0;
            if (last > 0) {
                for (int k = 0; k <= last; ++k) scale += std::fabs(a(row, k));
                if (scale == 0) {
                    off[row] = a(row, last);
                } else {
                    for (int k = 0; k <= last; ++k) { a(row, k) /= scale; h += a(row, k) * a(row, k); }
                    double f = a(row, last);
                    double g = f >= 0 ? -std::sqrt(h) : std::sqrt(h);
                    off[row] = scale * g;
                    h -= f * g;
                    a(row, last) = f - g;
                    f = 0;
                    for (int j = 0; j <= last; ++j) {
                        a(j, row) = a(row, j) / h;
                        g = 0;
                        for (int k = 0; k <= j; ++k) g += a(j, k) * a(row, k);
                        for (int k = j + 1; k <= last; ++k) g += a(k, j) * a(row, k);
                        off[j] = g / h;
                        f += off[j] * a(row, j);
                    }
                    const double hh = f / (h * 2);
                    for (int j = 0; j <= last; ++j) {
                        f = a(row, j);
                        off[j] = g = off[j] - hh * f;
                        for (int k = 0; k <= j; ++k) a(j, k) -= (f * off[k] + g * a(row, k));
                    }
                }
            } else {
                off[row] = a(row, last);
            }
            diag[row] = h;
        }
        diag[0] = off[0] = 0;
        for (int row = 0; row < n; ++row) {
            const int last = row - 1;
            if (diag[row]) {
                for (int j = 0; j <= last; ++j) {
                    double g = 0;
                    for (int k = 0; k <= last; ++k) g += a(row, k) * a(k, j);
                    for (int k = 0; k <= last; ++k) a(k, j) -= g * a(k, row);
                }
            }
            diag[row] = a(row, row);
            a(row, row) = 1;
            for (int j = 0; j <= last; ++j) a(j, row) = a(row, j) = 0;
        }
    }

    // eigen.cpp:249-318 (tqli, 30 iterations per eigenvalue); returns -1 when it does not converge
    int implicit_ql() {
        int status = 0;
        for (int i = 1; i < n; ++i) off[i - 1] = off[i];
        off[n - 1] = 0;
        for (int j = 0; j < n; ++j) {
            int iter = 0, m;
            do {
                for (m = j; m < n - 1; ++m) {
                    const double dd = std::fabs(diag[m]) + std::fabs(diag[m + 1]);
                    if (std::fabs(off[m]) + dd == dd) break;
                }
                if (m != j) {
                    if (iter++ == 30) { status = -1; break; }
                    double g = (diag[j + 1] - diag[j]) / (2 * off[j]);
                    double r;
                    {
                        const double ag = std::fabs(g);
                        r = ag > 1 ? ag * std::sqrt(1 + 1 / (g * g)) : std::sqrt(1 + g * g);
                    }
                    g = diag[m] - diag[j] + off[j] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                    double s = 1, c = 1, p = 0;
                    int i;
                    for (i = m - 1; i >= j; --i) {
                        double f = s * off[i];
                        const double b = c * off[i];
                        double af = std::fabs(f), ag = std::fabs(g);
                        if (af > ag) { ag /= af; r = af * std::sqrt(1 + ag * ag); }
                        else if (ag == 0) r = 0;
                        else { af /= ag; r = ag * std::sqrt(1 + af * af); }
                        off[i + 1] = r;
                        if (r == 0) { diag[i + 1] -= p; off[m] = 0; break; }
                        s = f / r;
                        c = g / r;
                        g = diag[i + 1] - p;
                        r = (diag[i] - g) * s + 2 * c * b;
                        diag[i + 1] = g + (p = s * r);
                        g = c * r - b;
                        for (int k = 0; k < n; ++k) {
                            f = a(k, i + 1);
                            a(k, i + 1) = s * a(k, i) + c * f;
                            a(k, i) = c * a(k, i) - s * f;
                        }
                    }
                    if (r == 0 && i >= j) continue;
                    diag[j] -= p; off[j] = g; off[m] = 0;
                }
            } while (m != j);
        }
        return status;
    }

    // eigen.cpp:152-174: selection sort, `>=` picks the LAST of equal eigenvalues
    void sort_descending() {
        for (int i = 0; i < n - 1; ++i) {
            int k = i;
            double p = diag[i];
            for (int j = i + 1; j < n; ++j) if (diag[j] >= p) p = diag[k = j];
            if (k == i) continue;
            diag[k] = diag[i]; diag[i] = p;
            for (int j = 0; j < n; ++j) std::swap(a(j, i), a(j, k));
        }
    }
};

} // namespace

// Eigen::getpi_sqrt + Eigen::eigenQREV, eigen.cpp:39-128.  Q = U diag(root) V, U V = I.
int eigen_qrev(const double *Q, const double *pi, int n, double *root, double *U, double *V) {
    std::vector<double> sq;                                   // sqrt(pi) of the states with pi != 0
    for (int j = 0; j < n; ++j) if (pi[j]) sq.push_back(std::sqrt(pi[j]));
    const int nn = (int)sq.size();
    std::vector<double> sym((size_t)nn * nn);
    int status;
    if (nn == n) {
        for (int i = 0; i < n; ++i) {
            sym[(size_t)i * n + i] = Q[i * n + i];
            for (int j = 0; j < i; ++j) sym[(size_t)i * n + j] = sym[(size_t)j * n + i] = (Q[i * n + j] * sq[i] / sq[j]);
        }
        SymEigen es(sym, n);
        es.tridiagonalise();
        status = es.implicit_ql();
        es.sort_descending();
        for (int i = 0; i < n; ++i) {
            root[i] = es.diag[i];
            for (int j = 0; j < n; ++j) { V[i * n + j] = sym[(size_t)j * n + i] * sq[j]; U[i * n + j] = sym[(size_t)i * n + j] / sq[i]; }
        }
    } else {
        // states with pi == 0 are cut out, the reduced problem solved and embedded again (eigen.cpp:83-123)
        std::vector<int> keep;
        for (int i = 0; i < n; ++i) if (pi[i]) keep.push_back(i);
        for (int a = 0; a < nn; ++a) {
            for (int b = 0; b < a; ++b)
                sym[(size_t)a * nn + b] = sym[(size_t)b * nn + a] = Q[keep[a] * n + keep[b]] * sq[a] / sq[b];
            sym[(size_t)a * nn + a] = Q[keep[a] * n + keep[a]];
        }
        SymEigen es(sym, nn);
        es.tridiagonalise();
        status = es.implicit_ql();
        es.sort_descending();
        std::vector<int> dense(n, -1);
        for (int a = 0; a < nn; ++a) dense[keep[a]] = a;
        for (int i = 0; i < n; ++i) {
            root[i] = dense[i] >= 0 ? es.diag[dense[i]] : 0;
            for (int j = 0; j < n; ++j) {
                if (dense[i] >= 0 && dense[j] >= 0) {
                    V[i * n + j] = sym[(size_t)dense[j] * nn + dense[i]] * sq[dense[j]];
                    U[i * n + j] = sym[(size_t)dense[i] * nn + dense[j]] / sq[dense[i]];
                } else {
                    V[i * n + j] = U[i * n + j] = (i == j);
                }
            }
        }
    }
    root[0] = 0;                                               // eigen.cpp:126
    return status;
}

int ModelFactory::guess_type(const std::vector<std::string> &seqs) {
    // (one table look-up per character; the counts are what the reference's two find() calls per character give)
    bool is_dna[256] = {false}, is_protein[256] = {false};
    for (const char *p = "ACGTUN"; *p; ++p) is_dna[(unsigned char)*p] = true;
    for (const char *p = protein_alphabet(); *p; ++p) is_protein[(unsigned char)*p] = true;
    long dna = 0, protein = 0;
    for (const std::string &s : seqs)
        for (char c : s) { dna += is_dna[(unsigned char)c]; protein += is_protein[(unsigned char)c]; }
    return ((float)dna) / (float)protein > 0.9 ? kDna : kProtein;
}

void ModelFactory::base_frequencies(const std::vector<std::string> &seqs, float out[4]) {
    // The reference counts in floats: a counter stops growing at 2^24 (x + 1.0f rounds back to x there), below that it is
    // the integer count.
    long n[256] = {0};
    for (const std::string &s : seqs)
        for (char ch : s) ++n[(unsigned char)ch];
    float c[4];
    const char base[4] = {'A', 'C', 'G', 'T'};
    for (int k = 0; k < 4; ++k) c[k] = (float)std::min(n[(unsigned char)base[k]], 16777216l);
    const float tot = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 4; ++k) out[k] = c[k] / tot;
}

void ModelFactory::init_dna(const float bf[4], float kappa, float rho) {
    type = kDna; S = 15; char_as = 4;
    leaf_alphabet = ancestral_alphabet = dna_full_alphabet();
    ins_rate = del_rate = 0.01f; ext_prob = 0.8f; end_ext_prob = 0.95f;                  // model_factory.cpp:1303-1306
    pi.assign(4, 0.0);
    for (int k = 0; k < 4; ++k) pi[k] = bf[k];                    // charPi, :1373-1376
    // model_factory.cpp:1378-1388, float arithmetic
    const float ka = kappa / 2.0;
    const float piR = bf[0] + bf[2], piY = bf[1] + bf[3];
    const float beta = 1 / (2 * piR * piY * (1 + ka));
    const float alfaY = (piR * piY * ka - bf[0] * bf[2] - bf[1] * bf[3]) /
                        ((2 + 2 * ka) * (piY * bf[0] * bf[2] * rho + piR * bf[1] * bf[3]));
    const float alfaR = rho * alfaY;
    double Q[16];
    auto q = [&](int i, int j) -> double & { return Q[i * 4 + j]; };
    q(0, 1) = beta * bf[1]; q(0, 2) = alfaR * bf[2] / piR + beta * bf[2]; q(0, 3) = beta * bf[3];          // :1395-1405
    q(0, 0) = 0 - q(0, 1) - q(0, 2) - q(0, 3);
    q(1, 0) = beta * bf[0]; q(1, 2) = beta * bf[2]; q(1, 3) = alfaY * bf[3] / piY + beta * bf[3];          // :1408-1418
    q(1, 1) = 0 - q(1, 0) - q(1, 2) - q(1, 3);
    q(2, 0) = alfaR * bf[0] / piR + beta * bf[0]; q(2, 1) = beta * bf[1]; q(2, 3) = beta * bf[3];          // :1421-1431
    q(2, 2) = 0 - q(2, 0) - q(2, 1) - q(2, 3);
    q(3, 0) = beta * bf[0]; q(3, 1) = alfaY * bf[1] / piY + beta * bf[1]; q(3, 2) = beta * bf[2];          // :1434-1444
    q(3, 3) = 0 - q(3, 0) - q(3, 1) - q(3, 2);
    U.assign(16, 0.0); V.assign(16, 0.0); root.assign(4, 0.0);
    eigen_qrev(Q, pi.data(), 4, root.data(), U.data(), V.data());
    // parsimony table, model_factory.cpp:147-227 (the DNA most-common table is the same table, :218-224)
    const int bits[15] = {1, 2, 4, 8, 1 | 4, 2 | 8, 1 | 2, 4 | 8, 1 | 8, 2 | 4, 2 | 4 | 8, 1 | 4 | 8, 1 | 2 | 8, 1 | 2 | 4, 15};
    int pos[16];
    for (int &p : pos) p = -1;
    for (int i = 0; i < 15; ++i) pos[bits[i]] = i;
    parsimony.assign(225, 0);
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
            const int v = bits[i] & bits[j];
            parsimony[i + j * 15] = v > 0 ? pos[v] : pos[bits[i] | bits[j]];
        }
    mostcommon = parsimony; mc_dim = 15;
    res1.clear(); res2.clear();
}

// Model_factory::define_protein_alphabet + protein_model (WAG), model_factory.cpp:304-632, 1478-1595.
void ModelFactory::init_protein() {
    type = kProtein; char_as = 20; S = 20 + 1 + 190;
    ins_rate = del_rate = 0.05f; ext_prob = 0.5f; end_ext_prob = 0.75f;                  // :1480-1497
    const std::string aa = protein_alphabet();
    pi.assign(kWagPi, kWagPi + 20);
    // symbols: 20 residues, X (all of them), then one code per unordered residue pair, first < second (:311-363)
    res1.assign(S, -1); res2.assign(S, -1);
    for (int i = 0; i < 20; ++i) res1[i] = (int16_t)i;
    res1[20] = 20;
    std::vector<int> pair_code(400, -1);
    {
        int code = 21;
        for (int i = 0; i < 19; ++i)
            for (int j = i + 1; j < 20; ++j) { res1[code] = (int16_t)i; res2[code] = (int16_t)j; pair_code[i * 20 + j] = pair_code[j * 20 + i] = code++; }
    }
    // Model_factory::get_protein_full_char_alphabet (model_factory.h:144-155): residues, X, then the pair
    // codes written as the lower-case first residue -- input is upper case, so a leaf only ever finds 0..20
    leaf_alphabet = aa + "X";
    for (int i = 0; i < 19; ++i) for (int j = i + 1; j < 20; ++j) leaf_alphabet.push_back((char)std::tolower(aa[i]));
    // ancestral_character_alphabet (:1581-1593): a pair code prints as its more frequent residue
    ancestral_alphabet = aa + "X";
    for (int i = 0; i < 19; ++i) for (int j = i + 1; j < 20; ++j) ancestral_alphabet.push_back(pi[i] > pi[j] ? aa[i] : aa[j]);
    // parsimony table (:403-541)
    parsimony.assign((size_t)S * S, 0);
    auto units = [&](int s) { return s == 20 ? 20 : (s < 20 ? 1 : 2); };
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            int v;
            if (i == j) v = i;
            else if (i == 20) v = j;
            else if (j == 20) v = i;
            else if (units(i) == 1 && units(j) == 1) v = pair_code[i * 20 + j];
            else if (units(i) == 1 && (res1[i] == res1[j] || res1[i] == res2[j])) v = res1[i];
            else if (units(j) == 1 && (res1[j] == res1[i] || res1[j] == res2[i])) v = res1[j];
            else {
                // the exchange with the largest rate among the (up to four) member pairs; the running maximum is
                // a float and only a strictly larger rate replaces it (:470-516)
                float best = -1;
                int b1 = 0, b2 = 0;
                auto consider = [&](int m, int n) { if (kWagQ[m * 20 + n] > best) { best = (float)kWagQ[m * 20 + n]; b1 = m; b2 = n; } };
                consider(res1[i], res1[j]);
                if (units(j) == 2) consider(res1[i], res2[j]);
                if (units(i) == 2) consider(res2[i], res1[j]);
                if (units(i) == 2 && units(j) == 2) consider(res2[i], res2[j]);
                v = b1 != b2 ? pair_code[b1 * 20 + b2] : 0;
            }
            parsimony[i + (size_t)j * S] = v;
        }
    // most-common table: 20 x 20 over the residues only (:621-630)
    mc_dim = 20;
    mostcommon.assign(400, 0);
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) mostcommon[i + j * 20] = kWagPi[i] > kWagPi[j] ? i : j;
    U.assign(400, 0.0); V.assign(400, 0.0); root.assign(20, 0.0);
    eigen_qrev(kWagQ, pi.data(), 20, root.data(), U.data(), V.data());
}

const char *ModelFactory::codon_alphabet() {
    return "AAAAACAAGAATACAACCACGACTAGAAGCAGGAGTATAATCATGATTCAACACCAGCATCCACCCCCGCCTCGACGCCGGCGTCTACTCCTGCTTGAAGACGAGGATGCAGCCGCG"
           "GCTGGAGGCGGGGGTGTAGTCGTGGTTTACTATTCATCCTCGTCTTGCTGGTGTTTATTCTTGTTTNNN";
}

namespace {
inline int base_code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
}

std::vector<int32_t> ModelFactory::codon_states(const std::string &nt, std::string *symbols) {
    // triplet -> state through a 64-entry table (the three stop codons stay 61)
    int8_t of_triplet[64];
    for (auto &v : of_triplet) v = 61;
    const char *names = codon_alphabet();
    for (int k = 0; k < 61; ++k)
        of_triplet[base_code(names[3 * k]) * 16 + base_code(names[3 * k + 1]) * 4 + base_code(names[3 * k + 2])] = (int8_t)k;
    std::vector<int32_t> out;
    out.reserve(nt.size() / 3 + 1);
    if (symbols) { symbols->clear(); symbols->reserve(nt.size() + 3); }
    for (size_t i = 0; i < nt.size(); i += 3) {
        int st = 61;
        if (i + 3 <= nt.size()) {
            const int a = base_code(nt[i]), b = base_code(nt[i + 1]), c = base_code(nt[i + 2]);
            if (a >= 0 && b >= 0 && c >= 0) st = of_triplet[a * 16 + b * 4 + c];
        }
        out.push_back(st);
        if (symbols) { if (st == 61) symbols->append("NNN"); else symbols->append(nt, i, 3); }
    }
    return out;
}

std::string ModelFactory::translate_codons(const std::string &cs) {
    // The reference's table as "amino acid: its codons" in the table's order; a codon listed twice keeps its FIRST amino
    // acid (std::map::insert, codon_translation.cpp:83-86): CTR is V there (and GTR is missing).  Packed into a 15^3 lookup
    // over the IUPAC letters once.
    static const char *const groups[] = {
        "M ATG", "W TGG", "F TTT TTC TTY", "Y TAT TAC TAY", "C TGT TGC TGY", "H CAT CAC CAY", "Q CAA CAG CAR", "N AAT AAC AAY",
        "K AAA AAG AAR", "D GAT GAC GAY", "E GAA GAG GAR", "I ATT ATC ATH ATA ATY ATW ATM",
        "P CCT CCC CCA CCG CCN CCY CCR CCM CCK CCS CCW CCB CCD CCH CCV",
        "T ACT ACC ACA ACG ACN ACY ACR ACM ACK ACS ACW ACB ACD ACH ACV",
        "V GTT GTC GTA GTG GTN GTY CTR GTM GTK GTS GTW GTB GTD GTH GTV",
        "A GCT GCC GCA GCG GCN GCY GCR GCM GCK GCS GCW GCB GCD GCH GCV",
        "G GGT GGC GGA GGG GGN GGY GGR GGM GGK GGS GGW GGB GGD GGH GGV",
        "S TCT TCC TCA TCG AGT AGC TCN TCY TCR TCM TCK TCS TCW TCB TCD TCH TCV AGY",
        "L TTA TTG CTT CTC CTA CTG CTN CTY CTR CTM CTK CTS CTW CTB CTD CTH CTV TTR",
        "R CGT CGC CGA CGG AGA AGG CGN CGY CGR CGM CGK CGS CGW CGB CGD CGH CGV AGR",
        "X NNN"};
    static const char letters[] = "ACGTRYMKWSBDHVN";
    struct Table {
        char aa[15 * 15 * 15];
        static int code(char c) { const char *p = c ? std::strchr(letters, c) : nullptr; return p ? (int)(p - letters) : -1; }
        Table() {
            std::memset(aa, 0, sizeof(aa));
            for (const char *g : groups)
                for (const char *p = g + 2; *p; p += (p[3] ? 4 : 3)) {
                    char &slot = aa[(code(p[0]) * 15 + code(p[1])) * 15 + code(p[2])];
                    if (!slot) slot = g[0];
                }
        }
    };
    static const Table table;
    std::string out;
    out.reserve(cs.size() / 3 + 1);
    for (size_t j = 0; j < cs.size(); j += 3) {
        char a = 'X';
        if (j + 3 <= cs.size()) {
            if (cs[j] == '-' && cs[j + 1] == '-' && cs[j + 2] == '-') a = '-';
            else {
                const int x = Table::code(cs[j]), y = Table::code(cs[j + 1]), z = Table::code(cs[j + 2]);
                if (x >= 0 && y >= 0 && z >= 0 && table.aa[(x * 15 + y) * 15 + z]) a = table.aa[(x * 15 + y) * 15 + z];
            }
        }
        out.push_back(a);
    }
    return out;
}

// Model_factory::define_codon_alphabet + codon_model, model_factory.cpp:839-1217, 1599-1805.
void ModelFactory::init_codon() {
    type = kCodon; char_as = 61; S = 61 + 1 + 1830;
    ins_rate = del_rate = 0.01f; ext_prob = 0.5f; end_ext_prob = 0.75f;                  // :1601-1618
    leaf_alphabet.clear(); ancestral_alphabet.clear();
    pi.assign(kCodonPi, kCodonPi + 61);
    // The code of the unordered pair lo < hi: pairs are numbered row by row after NNN (:879-896), which is what the
    // reference's running sum computes (:1011-1021): 61 + sum_{l<lo} (59 - l) + hi
    auto pair_code = [](int a, int b) { const int lo = std::min(a, b), hi = std::max(a, b); return 61 + lo * 59 - lo * (lo - 1) / 2 + hi; };
    res1.assign(S, -1); res2.assign(S, -1);
    for (int i = 0; i < 61; ++i) res1[i] = (int16_t)i;
    res1[61] = 61;
    for (int i = 0; i < 60; ++i)
        for (int j = i + 1; j < 61; ++j) { const int code = pair_code(i, j); res1[code] = (int16_t)i; res2[code] = (int16_t)j; }
    // names: a pair prints as the IUPAC code of its two bases, position by position (:1756-1800)
    const char *names = codon_alphabet();
    codon_names.assign(names, 62 * 3);
    codon_names.resize((size_t)S * 3, 'N');
    static const char iupac_of_mask[16] = {'N', 'A', 'C', 'M', 'G', 'R', 'S', 'N', 'T', 'W', 'Y', 'N', 'K', 'N', 'N', 'N'};
    for (int code = 62; code < S; ++code)
        for (int p = 0; p < 3; ++p)
            codon_names[(size_t)code * 3 + p] = iupac_of_mask[(1 << base_code(names[3 * res1[code] + p])) | (1 << base_code(names[3 * res2[code] + p]))];
    // parsimony table (:986-1109)
    parsimony.assign((size_t)S * S, 0);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            int v;
            if (i == j) v = i;
            else if (i == 61) v = j;
            else if (j == 61) v = i;
            else if (i < 61 && j < 61) v = pair_code(i, j);
            else if (i < 61 && (i == res1[j] || i == res2[j])) v = i;
            else if (j < 61 && (j == res1[i] || j == res2[i])) v = j;
            else {
                // the exchange with the largest rate among the member pairs: the running maximum is a float that starts
                // at the first pair's rate, and only a strictly larger rate replaces it (:1044-1088)
                int b1 = res1[i], b2 = res1[j];
                float best = (float)kCodonQ[b1 * 61 + b2];
                auto consider = [&](int m, int n) { if (kCodonQ[m * 61 + n] > best) { best = (float)kCodonQ[m * 61 + n]; b1 = m; b2 = n; } };
                if (j > 61) consider(res1[i], res2[j]);
                if (i > 61) consider(res2[i], res1[j]);
                if (i > 61 && j > 61) consider(res2[i], res2[j]);
                v = pair_code(b1, b2);
            }
            parsimony[i + (size_t)j * S] = v;
        }
    mc_dim = 61;                                                   // :1208-1217
    mostcommon.assign(61 * 61, 0);
    for (int i = 0; i < 61; ++i) for (int j = 0; j < 61; ++j) mostcommon[i + j * 61] = kCodonPi[i] > kCodonPi[j] ? i : j;
    U.assign(61 * 61, 0.0); V.assign(61 * 61, 0.0); root.assign(61, 0.0);
    eigen_qrev(kCodonQ, pi.data(), 61, root.data(), U.data(), V.data());
}

EvolModel ModelFactory::alignment_model(double distance, bool pileup_rates) const {
    EvolModel m;
    m.S = S; m.char_as = char_as;
    const int n = char_as;
    // Eigen::computePMatrix, eigen.cpp:330-358
    std::vector<double> P((size_t)n * n, 0.0);
    for (int k = 0; k < n; ++k) {
        const double e1 = std::exp(distance * root[k]);
        for (int i = 0; i < n; ++i) {
            const double e2 = U[(size_t)i * n + k] * e1;
            for (int j = 0; j < n; ++j) P[(size_t)i * n + j] += e2 * V[(size_t)k * n + j];
        }
    }
    const float ins = pileup_rates ? 0.25f : ins_rate, del = pileup_rates ? 0.25f : del_rate;   // :1901-1905
    m.log_gap_ext = std::log(ext_prob);                            // :1898 (float log)
    const double t = 1.0 - std::exp(-0.5 * (ins + del) * distance);   // :1913
    m.log_gap_open = (float)std::log(t);                           // :1915
    m.log_non_gap = (float)std::log(1.0 - 2 * t);                  // :1916
    m.log_gap_end_ext = std::log(end_ext_prob);                    // :1921
    std::vector<double> pr((size_t)S * S, 0.0), logpr((size_t)S * S, 0.0);
    auto PR = [&](int i, int j) -> double & { return pr[i + (size_t)j * S]; };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const float spf = (float)P[(size_t)i * n + j];         // :1937
            const float lo = (float)(0.5 * (pi[i] + pi[j]) * spf / (pi[i] * pi[j]));   // :1946
            PR(i, j) = lo;
            logpr[i + (size_t)j * S] = std::log(lo);               // :1948 log of a float
        }
    if (type == kDna) {
        static const char *sets[15] = {"\0", "\1", "\2", "\3", "\0\2", "\1\3", "\0\1", "\2\3", "\0\3", "\1\2",
                                       "\1\2\3", "\0\2\3", "\0\1\3", "\0\1\2", "\0\1\2\3"};
        static const int nset[15] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4};
        for (int i = 0; i < 15; ++i)                               // :1993-2016
            for (int j = 0; j < 15; ++j) {
                if (i < 4 && j < 4) continue;
                double mx = 0;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        bool in_i = false, in_j = false;
                        for (int k = 0; k < nset[i]; ++k) if (sets[i][k] == a) in_i = true;
                        for (int k = 0; k < nset[j]; ++k) if (sets[j][k] == b) in_j = true;
                        const double tt = PR(a, b) * (in_j ? 1.0 : 0.0) * (in_i ? 1.0 : 0.0);
                        if (mx < tt) mx = tt;
                    }
                PR(i, j) = mx;
                logpr[i + (size_t)j * S] = std::log(mx);
            }
    } else {
        // protein, the small-group path (:2155-2219), and codons (:2026-2090) -- the same extension over the pair
        // codes; rows in order, so X / NNN against an ambiguity code reads entries this loop has already extended
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                if (i < n && j < n) continue;
                double mx = 0;
                if (i == n) { for (int a = 0; a < n; ++a) mx = std::max(mx, PR(a, j)); }
                else if (j == n) { for (int b = 0; b < n; ++b) mx = std::max(mx, PR(i, b)); }
                else {
                    mx = PR(res1[i], res1[j]);
                    if (res2[j] >= 0) mx = std::max(mx, PR(res1[i], res2[j]));
                    if (res2[i] >= 0) mx = std::max(mx, PR(res2[i], res1[j]));
                    if (res2[i] >= 0 && res2[j] >= 0) mx = std::max(mx, PR(res2[i], res2[j]));
                }
                PR(i, j) = mx;
                logpr[i + (size_t)j * S] = std::log(mx);
            }
    }
    m.log_score.resize((size_t)S * S);
    for (size_t k = 0; k < m.log_score.size(); ++k) m.log_score[k] = (float)logpr[k];
    m.score.resize((size_t)S * S);
    for (size_t k = 0; k < m.score.size(); ++k) m.score[k] = (float)pr[k];      // Evol_model::score returns float
    m.gap_open = (float)t; m.non_gap = (float)(1.0 - 2 * t); m.gap_ext = ext_prob;          // :1917-1918, 1899
    return m;
}

} // namespace pagan
