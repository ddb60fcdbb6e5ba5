// host_model.cpp -- see host_model.h.
#include <string>
#include "host_model.h"

#include <algorithm>
#include <cmath>

namespace pagan {

namespace {

// Eigen decomposition of a real symmetric n x n matrix by cyclic Jacobi rotations.
// a (row-major) is destroyed; vec columns are the eigenvectors, val the eigenvalues,
// sorted descending like Eigen::EigenSort (src/utils/eigen.cpp:152-174).
void jacobi_sym(std::vector<double> &a, int n, std::vector<double> *vec, std::vector<double> *val) {
    std::vector<double> v(n * n, 0.0);
    for (int i = 0; i < n; ++i) v[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a[x * n + x] > a[y * n + y]; });
    vec->assign(n * n, 0.0); val->assign(n, 0.0);
    for (int k = 0; k < n; ++k) {
        (*val)[k] = a[order[k] * n + order[k]];
        for (int i = 0; i < n; ++i) (*vec)[i * n + k] = v[i * n + order[k]];
    }
}

} // namespace

void DnaModelFactory::base_frequencies(const std::vector<std::string> &seqs, float out[4]) {
    float c[4] = {0, 0, 0, 0};
    for (const std::string &s : seqs)
        for (char ch : s) {
            switch (ch) {
            case 'A': c[0]++; break;
            case 'C': c[1]++; break;
            case 'G': c[2]++; break;
            case 'T': c[3]++; break;
            default: break;
            }
        }
    const float tot = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 4; ++k) out[k] = c[k] / tot;
}

void DnaModelFactory::init(const float bf[4], float kappa, float rho) {
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
    // Eigen::eigenQREV, eigen.cpp:48-128 (all pi > 0): S = sqrt(D) Q sqrt(D)^-1 is symmetric
    double sp[4];
    for (int k = 0; k < 4; ++k) sp[k] = std::sqrt(pi[k]);
    std::vector<double> sym(16);
    for (int i = 0; i < 4; ++i) {
        sym[i * 4 + i] = q(i, i);
        for (int j = 0; j < i; ++j) sym[i * 4 + j] = sym[j * 4 + i] = q(i, j) * sp[i] / sp[j];
    }
    std::vector<double> vec, val;
    jacobi_sym(sym, 4, &vec, &val);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { V[i * 4 + j] = vec[j * 4 + i] * sp[j]; U[i * 4 + j] = vec[i * 4 + j] / sp[i]; }
    for (int k = 0; k < 4; ++k) root[k] = val[k];
    root[0] = 0;                                                   // eigen.cpp:126
    // parsimony table, model_factory.cpp:147-227
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
}

EvolModel DnaModelFactory::alignment_model(double distance) const {
    EvolModel m;
    m.S = 15; m.char_as = 4;
    // Eigen::computePMatrix, eigen.cpp:330-358
    double P[16] = {0};
    for (int k = 0; k < 4; ++k) {
        const double e1 = std::exp(distance * root[k]);
        for (int i = 0; i < 4; ++i) {
            const double e2 = U[i * 4 + k] * e1;
            for (int j = 0; j < 4; ++j) P[i * 4 + j] += e2 * V[k * 4 + j];
        }
    }
    m.log_gap_ext = std::log(ext_prob);                            // :1898 (float log)
    const double t = 1.0 - std::exp(-0.5 * (ins_rate + del_rate) * distance);   // :1913
    m.log_gap_open = (float)std::log(t);                           // :1915
    m.log_non_gap = (float)std::log(1.0 - 2 * t);                  // :1916
    m.log_gap_end_ext = std::log(end_ext_prob);                    // :1921
    double pr[225];
    double logpr[225];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const float spf = (float)P[i * 4 + j];                 // :1937
            const float lo = (float)(0.5 * (pi[i] + pi[j]) * spf / (pi[i] * pi[j]));   // :1946
            pr[i + j * 15] = lo;
            logpr[i + j * 15] = std::log(lo);                      // :1948 log of a float
        }
    static const char *sets[15] = {"\0", "\1", "\2", "\3", "\0\2", "\1\3", "\0\1", "\2\3", "\0\3", "\1\2",
                                   "\1\2\3", "\0\2\3", "\0\1\3", "\0\1\2", "\0\1\2\3"};
    static const int nset[15] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4};
    for (int i = 0; i < 15; ++i)                                   // :1993-2016
        for (int j = 0; j < 15; ++j) {
            if (i < 4 && j < 4) continue;
            double mx = 0;
            for (int n = 0; n < 4; ++n)
                for (int mm = 0; mm < 4; ++mm) {
                    bool in_i = false, in_j = false;
                    for (int k = 0; k < nset[i]; ++k) if (sets[i][k] == n) in_i = true;
                    for (int k = 0; k < nset[j]; ++k) if (sets[j][k] == mm) in_j = true;
                    const double tt = pr[n + mm * 15] * (in_j ? 1.0 : 0.0) * (in_i ? 1.0 : 0.0);
                    if (mx < tt) mx = tt;
                }
            pr[i + j * 15] = mx;
            logpr[i + j * 15] = std::log(mx);
        }
    m.log_score.resize(225);
    for (int k = 0; k < 225; ++k) m.log_score[k] = (float)logpr[k];
    return m;
}

} // namespace pagan
