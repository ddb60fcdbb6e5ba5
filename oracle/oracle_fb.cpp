// oracle_fb.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_dp.cpp header; PARITY UNPINNED).
//
// CPU restatement of the reference's forward/backward sum-product pass over the same three matrices
// (compute_full_score, src/main/basic_alignment.h:621-625):
//
//   forward()    <- the `if(compute_full_score)` halves of score_gap_ext/double/open
//                   (src/main/viterbi_alignment.cpp:2151-2155, 2182-2186, 2213-2217), of
//                   score_m/x/y_match (:2049-2054, 2078-2083, 2106-2111) with the factors set up in
//                   iterate_bwd_edges_for_match (:1376-1393), and of the end corner (:1440-1552 with
//                   score_gap_close :2249-2253), corner value initialise_array_corner (:725-736)
//   backward()   <- initialise_array_corner_bwd (:740-854), compute_bwd_full_score (:975-1026),
//                   iterate_fwd_edges_for_gap / _for_match (:1571-1662), score_*_bwd (:2259-2305)
//   posterior()  <- compute_posterior_score (:1029-1034)
//   sample()     <- sample_new_path (:1193-1322) with iterate_bwd_edges_for_sampled_gap / _match /
//                   _end_corner (:1666-2025) and add_sample_* (:2309-2446); the uniform numbers the
//                   reference takes from rand() are an input here
//
// One set of loops, two arithmetics: `Prob` multiplies and adds plain doubles exactly like the
// reference (which under/overflows on long inputs), `LogProb` keeps logarithms (product = sum,
// sum = log-sum-exp).  Tests require the two to agree to 1e-6 relative on short inputs, and the GPU
// (log space) to agree with LogProb.
//
// Reference quirks kept: the full-probability terms always use gap_ext (never the end-gap
// extension) and the plain gap-open probability (no reduced terminal penalty); edge weights enter
// matches only; the end corner's loops visit Y-close of a non-first right edge once per left edge
// (:1476-1547), so those terms are counted more than once in fwd_end.
// Deviation: cells outside the tunnel are not computed and read as probability 0 (the reference
// accumulates into the shared out-of-tunnel cell, src/utils/tunnel_matrix.h:85-98).
#include "../include/pagan_dp.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Prob {
    double v = 0.0;
    static Prob of(double p) { Prob x; x.v = p; return x; }
    static Prob from_log(double l) { return of(std::exp(l)); }
    static Prob zero() { return of(0.0); }
    Prob operator*(Prob o) const { return of(v * o.v); }
    void operator+=(Prob o) { v += o.v; }
    double log() const { return std::log(v); }
    double lin() const { return v; }
};

struct LogProb {
    double v = -HUGE_VAL;
    static LogProb of(double p) { LogProb x; x.v = std::log(p); return x; }
    static LogProb from_log(double l) { LogProb x; x.v = l; return x; }
    static LogProb zero() { return LogProb(); }
    LogProb operator*(LogProb o) const { LogProb x; x.v = v + o.v; return x; }
    void operator+=(LogProb o) {
        if (o.v == -HUGE_VAL) return;
        if (v == -HUGE_VAL) { v = o.v; return; }
        const double hi = v > o.v ? v : o.v, lo = v > o.v ? o.v : v;
        v = hi + std::log1p(std::exp(lo - hi));
    }
    double log() const { return v; }
    double lin() const { return std::exp(v); }
};

struct ProbModel {          // Evol_model's probability-space accessors (src/utils/evol_model.h:70-88)
    int S; const float *score; float gap_open, gap_ext, non_gap;
};

struct Fwd { std::vector<int> off, dst, eid; std::vector<float> w; };      // fwd lists, creation order; w = log weight

Fwd forward_lists(const pagan_graph *g) {
    struct E { int src, dst, eid; float w; };
    std::vector<E> es;
    for (int s = 0; s < g->n_sites; s++)
        for (int k = g->bwd_off[s]; k < g->bwd_off[s + 1]; k++) es.push_back({g->bwd_src[k], s, g->bwd_eid[k], g->bwd_logw[k]});
    std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.src != b.src ? a.src < b.src : a.eid < b.eid; });
    Fwd f;
    f.off.assign(g->n_sites + 1, 0);
    for (const E &e : es) f.off[e.src + 1]++;
    for (int s = 0; s < g->n_sites; s++) f.off[s + 1] += f.off[s];
    for (const E &e : es) { f.dst.push_back(e.dst); f.eid.push_back(e.eid); f.w.push_back(e.w); }
    return f;
}

template <class P> struct Pass {
    const pagan_graph *L, *R;
    ProbModel m;
    int Lx, Ly;
    std::vector<int> lo, hi;                 // band per row
    std::vector<size_t> roff;                // cells of the rows above row i: only the band's cells are stored (a 2 x 100 kb tunnel
                                             // is 9e6 cells of a 1e10-cell matrix)
    std::vector<int> cmin, cmax;             // first / last row whose band holds column j (cmax < cmin: none)
    std::vector<P> fw, bw;                   // [(roff[i] + j - lo[i])*3 + state], state 0 X, 1 Y, 2 M
    P fwd_end = P::zero();

    bool in(int i, int j) const { return i >= 0 && i < Lx && j >= lo[i] && j <= hi[i]; }
    size_t at(int s, int i, int j) const { return (roff[i] + (size_t)(j - lo[i])) * 3 + s; }
    P F(int s, int i, int j) const { return in(i, j) ? fw[at(s, i, j)] : P::zero(); }
    P B(int s, int i, int j) const { return in(i, j) ? bw[at(s, i, j)] : P::zero(); }
    P &Fw(int s, int i, int j) { return fw[at(s, i, j)]; }
    P &Bw(int s, int i, int j) { return bw[at(s, i, j)]; }
    // get_edge_weight (probability space) is the float posterior weight w; the CSR carries logf(w), so the weight
    // used here is exp((double) logf(w)) = w (1 +- 6e-8) for the few edges with w != 1 (w = 1 is exact)
    P lw(int k) const { return P::from_log((double)L->bwd_logw[k]); }
    P rw(int k) const { return P::from_log((double)R->bwd_logw[k]); }
    P emit(int i, int j) const { return P::of((double)m.score[L->state[i] + (size_t)R->state[j] * m.S]); }

    Pass(const pagan_graph *l, const pagan_graph *r, const ProbModel &pm, const pagan_band *band) : L(l), R(r), m(pm) {
        Lx = l->n_sites - 1; Ly = r->n_sites - 1;
        lo.assign(Lx, 0); hi.assign(Lx, Ly - 1);
        if (band) for (int i = 0; i < Lx; i++) { lo[i] = std::max(0, band->upper[i]); hi[i] = std::min(band->lower[i], Ly - 1); }
        roff.assign(Lx + 1, 0);
        cmin.assign(Ly, Lx); cmax.assign(Ly, -1);
        for (int i = 0; i < Lx; i++) {
            roff[i + 1] = roff[i] + (size_t)std::max(0, hi[i] - lo[i] + 1);
            for (int j = lo[i]; j <= hi[i]; j++) { cmin[j] = std::min(cmin[j], i); cmax[j] = std::max(cmax[j], i); }
        }
        fw.assign(roff[Lx] * 3, P::zero()); bw = fw;
    }

    void forward() {
        const P ext = P::of(m.gap_ext), open = P::of(m.gap_open), ng = P::of(m.non_gap), close = P::of(1.0f);
        if (in(0, 0)) Fw(2, 0, 0) = P::of(1.0);                                        // VA:730
        for (int i = 0; i < Lx; i++)
            for (int j = lo[i]; j <= hi[i]; j++) {
                if (i == 0 && j == 0) continue;
                if (i > 0) {                                                           // VA:898-915
                    P acc = P::zero();
                    for (int k = L->bwd_off[i]; k < L->bwd_off[i + 1]; k++) {
                        const int p = L->bwd_src[k];
                        acc += F(0, p, j) * ext;                                       // :2153
                        acc += F(1, p, j) * close * open;                              // :2184
                        acc += F(2, p, j) * ng * open;                                 // :2215
                    }
                    Fw(0, i, j) = acc;
                }
                if (j > 0) {
                    P acc = P::zero();
                    for (int k = R->bwd_off[j]; k < R->bwd_off[j + 1]; k++) {
                        const int q = R->bwd_src[k];
                        acc += F(1, i, q) * ext;
                        acc += F(0, i, q) * close * open;
                        acc += F(2, i, q) * ng * open;
                    }
                    Fw(1, i, j) = acc;
                }
                if (i > 0 && j > 0 && L->bwd_off[i + 1] > L->bwd_off[i] && R->bwd_off[j + 1] > R->bwd_off[j]) {
                    const P sc = emit(i, j);
                    const P mm = ng * ng * sc, xm = close * ng * sc;                   // VA:1383-1391
                    P acc = P::zero();
                    for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; k1++)
                        for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; k2++) {
                            const int p = L->bwd_src[k1], q = R->bwd_src[k2];
                            acc += F(2, p, q) * mm * lw(k1) * rw(k2);                  // :2051
                            acc += F(0, p, q) * xm * lw(k1) * rw(k2);                  // :2080
                            acc += F(1, p, q) * xm * lw(k1) * rw(k2);                  // :2108
                        }
                    Fw(2, i, j) = acc;
                }
            }
        // end corner, VA:1440-1552: M over every edge pair; X-close once per left edge; Y-close of the first
        // right edge once, of every further right edge once per left edge
        P acc = P::zero();
        const int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1], r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        if (l1 > l0 && r1 > r0) {
            auto mterm = [&](int k1, int k2) { return F(2, L->bwd_src[k1], R->bwd_src[k2]) * ng * lw(k1) * rw(k2); };
            auto xc = [&](int k1) { return F(0, L->bwd_src[k1], Ly - 1) * close; };
            auto yc = [&](int k2) { return F(1, Lx - 1, R->bwd_src[k2]) * close; };
            acc += mterm(l0, r0); acc += xc(l0); acc += yc(r0);
            for (int k2 = r0 + 1; k2 < r1; k2++) { acc += mterm(l0, k2); acc += yc(k2); }
            for (int k1 = l0 + 1; k1 < l1; k1++) {
                acc += mterm(k1, r0); acc += xc(k1);
                for (int k2 = r0 + 1; k2 < r1; k2++) { acc += mterm(k1, k2); acc += yc(k2); }
            }
        }
        fwd_end = acc;
    }

    void backward() {
        const P ext = P::of(m.gap_ext), open = P::of(m.gap_open), ng = P::of(m.non_gap), close = P::of(1.0f);
        const Fwd fl = forward_lists(L), fr = forward_lists(R);
        // initialise_array_corner_bwd, VA:740-854 (assignments, not sums)
        if (in(Lx - 1, Ly - 1)) Bw(2, Lx - 1, Ly - 1) = ng;
        const int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1], r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        if (l1 > l0 && r1 > r0)
            for (int k1 = l0; k1 < l1; k1++)
                for (int k2 = r0; k2 < r1; k2++) {
                    const int p = L->bwd_src[k1], q = R->bwd_src[k2];
                    if (in(p, q)) Bw(2, p, q) = ng * lw(k1) * rw(k2);
                }
        for (int k1 = l0; k1 < l1; k1++) if (in(L->bwd_src[k1], Ly - 1)) Bw(0, L->bwd_src[k1], Ly - 1) = close;
        for (int k2 = r0; k2 < r1; k2++) if (in(Lx - 1, R->bwd_src[k2])) Bw(1, Lx - 1, R->bwd_src[k2]) = close;
        for (int j = Ly - 1; j >= 0; j--)
            for (int i = cmax[j]; i >= cmin[j]; i--) {          // (the rows of the column's band cells, last first: the same order as over all rows)
                if (!in(i, j)) continue;
                P bx = Bw(0, i, j), by = Bw(1, i, j), bm = Bw(2, i, j);
                for (int k = fl.off[i]; k < fl.off[i + 1]; k++) {                      // iterate_fwd_edges_for_gap, left
                    const int t = fl.dst[k];
                    if (t >= Lx) continue;                                             // :1580 edge->end < slice_end
                    const P nx = B(0, t, j);
                    bx += nx * ext; by += nx * close * open; bm += nx * ng * open;     // :2281-2303
                }
                for (int k = fr.off[j]; k < fr.off[j + 1]; k++) {
                    const int t = fr.dst[k];
                    if (t >= Ly) continue;
                    const P ny = B(1, i, t);
                    by += ny * ext; bx += ny * close * open; bm += ny * ng * open;
                }
                for (int k1 = fl.off[i]; k1 < fl.off[i + 1]; k1++)                      // iterate_fwd_edges_for_match
                    for (int k2 = fr.off[j]; k2 < fr.off[j + 1]; k2++) {
                        const int t = fl.dst[k1], u = fr.dst[k2];
                        if (t >= Lx || u >= Ly) continue;
                        const P thru = B(2, t, u) * emit(t, u) * P::from_log((double)fl.w[k1]) * P::from_log((double)fr.w[k2]);   // :2269-2271
                        bx += thru * (close * ng); by += thru * (close * ng); bm += thru * (ng * ng);
                    }
                Bw(0, i, j) = bx; Bw(1, i, j) = by; Bw(2, i, j) = bm;
            }
    }
};

template <class P>
int run(const pagan_graph *l, const pagan_graph *r, const ProbModel &pm, const pagan_band *band, double *log_fwd,
        double *log_bwd, double *post /* [Lx*Ly*3] or null: full_score, probability space */,
        double *log_f /* [Lx*Ly*3] or null */) {
    Pass<P> ps(l, r, pm, band);
    ps.forward();
    ps.backward();
    *log_fwd = ps.fwd_end.log();
    *log_bwd = ps.B(2, 0, 0).log();
    // (outside the band: probability 0 -- posterior 0, log forward -inf -- as the stored zeros gave before)
    if (post || log_f)
        for (int i = 0; i < ps.Lx; i++)
            for (int j = 0; j < ps.Ly; j++)
                for (int st = 0; st < 3; st++) {
                    const size_t k = ((size_t)i * ps.Ly + j) * 3 + st;
                    const bool in = ps.in(i, j);
                    if (post) post[k] = in ? std::exp(ps.F(st, i, j).log() + ps.B(st, i, j).log() - ps.fwd_end.log()) : 0.0;   // VA:1029-1034
                    if (log_f) log_f[k] = in ? ps.F(st, i, j).log() : -HUGE_VAL;
                }
    return 0;
}

} // namespace

extern "C" {

// arithmetic: 0 = probability space (the reference's), 1 = log space
int oracle_fb(const pagan_graph *left, const pagan_graph *right, int32_t S, const float *score, float gap_open, float gap_ext,
              float non_gap, const pagan_band *band, int arithmetic, double *log_fwd, double *log_bwd, double *post, double *log_f) {
    ProbModel pm{S, score, gap_open, gap_ext, non_gap};
    return arithmetic == 0 ? run<Prob>(left, right, pm, band, log_fwd, log_bwd, post, log_f)
                           : run<LogProb>(left, right, pm, band, log_fwd, log_bwd, post, log_f);
}

// sample_new_path (VA:1193-1322) on log forward scores `log_f` ([Lx*Ly*3], -inf outside the band): at every step
// the candidates are listed in the forward pass's order with weight fwd(pred) * transition (add_sample_*,
// VA:2309-2446), and the first one whose running sum reaches u * total is taken (VA:1757-1769); u[k] in [0,1)
// replaces rand()/(RAND_MAX+1), one per step, the end corner first.  Writes the visited cells (i, j, state the cell
// was entered in) end -> start, 3 ints each, and returns their number (<= Lx+Ly), or -1.
int oracle_sample_path(const pagan_graph *L, const pagan_graph *R, int32_t S, const float *score, float gap_open, float gap_ext,
                       float non_gap, const double *log_f, const double *u, int n_u, int32_t *cells, int32_t *end) {
    const int Lx = L->n_sites - 1, Ly = R->n_sites - 1;
    auto F = [&](int s, int i, int j) { return (i < 0 || j < 0) ? -HUGE_VAL : log_f[((size_t)i * Ly + j) * 3 + s]; };
    const double ext = std::log((double)gap_ext), open = std::log((double)gap_open), ng = std::log((double)non_gap);
    struct Cand { double lw; int state, i, j; };
    auto pick = [&](std::vector<Cand> &c, double uu) -> int {
        double hi = -HUGE_VAL;
        for (auto &x : c) hi = std::max(hi, x.lw);
        if (hi == -HUGE_VAL) return -1;
        double tot = 0;
        for (auto &x : c) tot += std::exp(x.lw - hi);
        const double rv = tot * uu;
        size_t k = 0;
        double sum = std::exp(c[0].lw - hi);
        while (sum < rv && k + 1 < c.size()) { ++k; sum += std::exp(c[k].lw - hi); }
        return (int)k;
    };
    int used = 0, n = 0;
    std::vector<Cand> c;
    // end corner (iterate_bwd_edges_for_sampled_end_corner, VA:1904-2025): same order as the forward end corner
    {
        const int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1], r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        auto mt = [&](int k1, int k2) { c.push_back({F(2, L->bwd_src[k1], R->bwd_src[k2]) + ng + (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2], 2, L->bwd_src[k1], R->bwd_src[k2]}); };
        auto xc = [&](int k1) { c.push_back({F(0, L->bwd_src[k1], Ly - 1), 0, L->bwd_src[k1], Ly - 1}); };
        auto yc = [&](int k2) { c.push_back({F(1, Lx - 1, R->bwd_src[k2]), 1, Lx - 1, R->bwd_src[k2]}); };
        if (l1 > l0 && r1 > r0) {
            mt(l0, r0); xc(l0); yc(r0);
            for (int k2 = r0 + 1; k2 < r1; k2++) { mt(l0, k2); yc(k2); }
            for (int k1 = l0 + 1; k1 < l1; k1++) { mt(k1, r0); xc(k1); for (int k2 = r0 + 1; k2 < r1; k2++) { mt(k1, k2); yc(k2); } }
        }
    }
    if (used >= n_u) return -1;
    int k = pick(c, u[used++]);
    if (k < 0) return -1;
    int state = c[k].state, i = c[k].i, j = c[k].j;
    end[0] = state; end[1] = i; end[2] = j;
    while (!(i < 1 && j < 1)) {
        cells[3 * n] = i; cells[3 * n + 1] = j; cells[3 * n + 2] = state; n++;
        c.clear();
        if (state == 2) {
            const double sc = std::log((double)score[L->state[i] + (size_t)R->state[j] * S]);
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; k1++)
                for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; k2++) {
                    const int p = L->bwd_src[k1], q = R->bwd_src[k2];
                    const double w = (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2];
                    c.push_back({F(2, p, q) + ng + ng + sc + w, 2, p, q});
                    c.push_back({F(0, p, q) + ng + sc + w, 0, p, q});
                    c.push_back({F(1, p, q) + ng + sc + w, 1, p, q});
                }
        } else if (state == 0) {
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; k1++) {
                const int p = L->bwd_src[k1];
                c.push_back({F(0, p, j) + ext, 0, p, j}); c.push_back({F(1, p, j) + open, 1, p, j}); c.push_back({F(2, p, j) + ng + open, 2, p, j});
            }
        } else {
            for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; k2++) {
                const int q = R->bwd_src[k2];
                c.push_back({F(1, i, q) + ext, 1, i, q}); c.push_back({F(0, i, q) + open, 0, i, q}); c.push_back({F(2, i, q) + ng + open, 2, i, q});
            }
        }
        if (c.empty() || used >= n_u) return -1;
        k = pick(c, u[used++]);
        if (k < 0) return -1;
        state = c[k].state; i = c[k].i; j = c[k].j;
    }
    return n;
}

} // extern "C"
