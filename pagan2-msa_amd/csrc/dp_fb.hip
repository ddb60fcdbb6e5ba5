// dp_fb.hip -- forward/backward sum-product over the three alignment matrices, posteriors, path sampling.
//
// Counterpart of the reference's `compute_full_score` pass (src/main/basic_alignment.h:621-625: --full-probability,
// --sample-path, --mpost-posterior-plot-file): the forward sums accumulated beside the Viterbi maxima
// (src/main/viterbi_alignment.cpp:2049-2054, 2078-2083, 2106-2111, 2151-2155, 2182-2186, 2213-2217, 2249-2253, with the
// factors of :1376-1393 and the end corner :1440-1552), the backward pass (:329-341, 740-854, 975-1026, 1571-1662,
// 2259-2305), the posterior (:1029-1034) and sample_new_path (:1193-1322, 1666-2025, 2309-2446).
//
// The reference multiplies raw probabilities, which under- and overflows beyond a few hundred columns; here every
// quantity is a logarithm (product = sum, sum = log-sum-exp in fp64).  Same cell layout as the Viterbi kernels
// (diagonal-major, [cell][X, Y, M]), same anti-diagonal wavefront: forward sweeps d = 0 .. nd-1 reading predecessor
// cells through the bwd edge lists, backward sweeps d = nd-1 .. 0 reading successor cells through the fwd edge lists
// (built on the host from the bwd CSR: a site's fwd list is its outgoing edges in creation order).  One workgroup per
// alignment, a __syncthreads() per anti-diagonal, operands in HBM/L2: this is the plain formulation (the "next" row f3
// of the scope table), not the latency-tuned one of dp_pipe.hip.
//
// Reference quirks kept (see oracle/oracle_fb.cpp, which restates the same rules on the CPU): full-probability terms
// use gap_ext for every gap (no end-gap extension) and the plain gap-open probability; edge weights enter matches only;
// the end corner visits the Y-close term of a non-first right edge once per left edge.  Cells outside the tunnel hold
// probability 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/pagan_dp.h"
#include "dp_band.h"

struct PgFbJob {
    int Lx, Ly, nd, S;
    double l_ext, l_open, l_ng;              // logs of gap_ext, gap_open, non_gap
    const int *stL, *offL, *srcL; const float *lwL;      // bwd lists (log weights as the CSR carries them)
    const int *stR, *offR, *srcR; const float *lwR;
    const int *foffL, *fdstL; const float *flwL;         // fwd lists
    const int *foffR, *fdstR; const float *flwR;
    const double *ltab;                      // log((double) score[a + b*S])
    const int *imin, *imax; const long long *doff;
    long long cells;
    double *F, *B;                           // [cells][3] log forward / log backward
    int n_init; const long long *init_at; const double *init_val;   // initialise_array_corner_bwd
    double *totals;                          // [2]: log fwd_end, log bwd(M,0,0)
};

namespace {

#define FB_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (std::getenv("PAGAN_DP_VERBOSE")) std::fprintf(stderr, "pagan_fb: %s: %s\n", #expr, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? PAGAN_E_NOMEM : PAGAN_E_NODEVICE;                            \
        }                                                                                                    \
    } while (0)

__device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }

// log(exp(a) + exp(b))
__device__ __forceinline__ double lse(double a, double b) {
    if (b == ninf()) return a;
    if (a == ninf()) return b;
    const double hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1p(exp(lo - hi));
}

// log(exp(a) + exp(b) + exp(c)) with one log1p: the largest term is factored out, the other two cost an exp each (the chain
// lse(lse(a, b), c) costs two exp and two log1p; the sums agree to the last few ulps, the tests compare logs to 1e-9)
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double hi = fmax(a, fmax(b, c));
    if (hi == ninf()) return hi;
    double rest;
    if (a == hi) rest = exp(b - hi) + exp(c - hi);
    else if (b == hi) rest = exp(a - hi) + exp(c - hi);
    else rest = exp(a - hi) + exp(b - hi);
    return hi + log1p(rest);
}

__device__ __forceinline__ long long cell_at(const PgFbJob &J, int p, int q) {
    if (p < 0 || q < 0 || p >= J.Lx || q >= J.Ly) return -1;
    const int d = p + q;
    const int mn = J.imin[d], mx = J.imax[d];
    return (p >= mn && p <= mx) ? J.doff[d] + (p - mn) : -1;
}

__device__ __forceinline__ double rd(const double *A, long long at, int s) { return at >= 0 ? A[3 * at + s] : ninf(); }

__global__ __launch_bounds__(1024) void pg_fb_forward(const PgFbJob *jobs) {
    const PgFbJob J = jobs[blockIdx.x];
    for (int d = 0; d < J.nd; ++d) {
        const int mn = J.imin[d], mx = J.imax[d];
        const long long off = J.doff[d];
        for (int i = mn + (int)threadIdx.x; i <= mx; i += (int)blockDim.x) {
            const int j = d - i;
            double fx = ninf(), fy = ninf(), fm = ninf();
            if (i == 0 && j == 0) {
                fm = 0.0;                                                          // fwd_score = 1, VA:730
            } else {
                if (i > 0)
                    for (int k = J.offL[i]; k < J.offL[i + 1]; ++k) {
                        const long long at = cell_at(J, J.srcL[k], j);
                        fx = lse(fx, lse3(rd(J.F, at, 0) + J.l_ext,                 // VA:2153
                                          rd(J.F, at, 1) + J.l_open,                // VA:2184 (gap_close = 1)
                                          rd(J.F, at, 2) + J.l_ng + J.l_open));     // VA:2215
                    }
                if (j > 0)
                    for (int k = J.offR[j]; k < J.offR[j + 1]; ++k) {
                        const long long at = cell_at(J, i, J.srcR[k]);
                        fy = lse(fy, lse3(rd(J.F, at, 1) + J.l_ext, rd(J.F, at, 0) + J.l_open, rd(J.F, at, 2) + J.l_ng + J.l_open));
                    }
                if (i > 0 && j > 0) {
                    const double sc = J.ltab[J.stL[i] + (long long)J.stR[j] * J.S];
                    const double mm = J.l_ng + J.l_ng + sc, xm = J.l_ng + sc;      // VA:1383-1391
                    for (int k1 = J.offL[i]; k1 < J.offL[i + 1]; ++k1)
                        for (int k2 = J.offR[j]; k2 < J.offR[j + 1]; ++k2) {
                            const long long at = cell_at(J, J.srcL[k1], J.srcR[k2]);
                            const double w = (double)J.lwL[k1] + (double)J.lwR[k2];
                            fm = lse(fm, lse3(rd(J.F, at, 2) + mm + w,              // VA:2051
                                              rd(J.F, at, 0) + xm + w,              // VA:2080
                                              rd(J.F, at, 1) + xm + w));            // VA:2108
                        }
                }
            }
            double *o = J.F + 3 * (off + (i - mn));
            o[0] = fx; o[1] = fy; o[2] = fm;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // end corner, VA:1440-1552
        double acc = ninf();
        const int l0 = J.offL[J.Lx], l1 = J.offL[J.Lx + 1], r0 = J.offR[J.Ly], r1 = J.offR[J.Ly + 1];
        auto mt = [&](int k1, int k2) { return rd(J.F, cell_at(J, J.srcL[k1], J.srcR[k2]), 2) + J.l_ng + (double)J.lwL[k1] + (double)J.lwR[k2]; };
        auto xc = [&](int k1) { return rd(J.F, cell_at(J, J.srcL[k1], J.Ly - 1), 0); };
        auto yc = [&](int k2) { return rd(J.F, cell_at(J, J.Lx - 1, J.srcR[k2]), 1); };
        if (l1 > l0 && r1 > r0) {
            acc = lse(acc, mt(l0, r0)); acc = lse(acc, xc(l0)); acc = lse(acc, yc(r0));
            for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(l0, k2)); acc = lse(acc, yc(k2)); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) {
                acc = lse(acc, mt(k1, r0)); acc = lse(acc, xc(k1));
                for (int k2 = r0 + 1; k2 < r1; ++k2) { acc = lse(acc, mt(k1, k2)); acc = lse(acc, yc(k2)); }
            }
        }
        J.totals[0] = acc;
    }
}

__global__ __launch_bounds__(1024) void pg_fb_backward(const PgFbJob *jobs) {
    const PgFbJob J = jobs[blockIdx.x];
    for (long long k = threadIdx.x; k < 3 * J.cells; k += blockDim.x) J.B[k] = ninf();
    __syncthreads();
    for (int k = threadIdx.x; k < J.n_init; k += blockDim.x) J.B[J.init_at[k]] = J.init_val[k];   // VA:740-854
    __syncthreads();
    for (int d = J.nd - 1; d >= 0; --d) {
        const int mn = J.imin[d], mx = J.imax[d];
        const long long off = J.doff[d];
        for (int i = mn + (int)threadIdx.x; i <= mx; i += (int)blockDim.x) {
            const int j = d - i;
            double *o = J.B + 3 * (off + (i - mn));
            double bx = o[0], by = o[1], bm = o[2];
            for (int k = J.foffL[i]; k < J.foffL[i + 1]; ++k) {                    // iterate_fwd_edges_for_gap, left site
                const int t = J.fdstL[k];
                if (t >= J.Lx) continue;                                           // VA:1580
                const double nx = rd(J.B, cell_at(J, t, j), 0);
                bx = lse(bx, nx + J.l_ext); by = lse(by, nx + J.l_open); bm = lse(bm, nx + J.l_ng + J.l_open);   // VA:2281-2303
            }
            for (int k = J.foffR[j]; k < J.foffR[j + 1]; ++k) {
                const int t = J.fdstR[k];
                if (t >= J.Ly) continue;
                const double ny = rd(J.B, cell_at(J, i, t), 1);
                by = lse(by, ny + J.l_ext); bx = lse(bx, ny + J.l_open); bm = lse(bm, ny + J.l_ng + J.l_open);
            }
            for (int k1 = J.foffL[i]; k1 < J.foffL[i + 1]; ++k1)                    // iterate_fwd_edges_for_match
                for (int k2 = J.foffR[j]; k2 < J.foffR[j + 1]; ++k2) {
                    const int t = J.fdstL[k1], u = J.fdstR[k2];
                    if (t >= J.Lx || u >= J.Ly) continue;
                    const double thru = rd(J.B, cell_at(J, t, u), 2) + J.ltab[J.stL[t] + (long long)J.stR[u] * J.S] +
                                        (double)J.flwL[k1] + (double)J.flwR[k2];   // VA:2269-2271
                    bx = lse(bx, thru + J.l_ng); by = lse(by, thru + J.l_ng); bm = lse(bm, thru + J.l_ng + J.l_ng);
                }
            o[0] = bx; o[1] = by; o[2] = bm;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) J.totals[1] = rd(J.B, cell_at(J, 0, 0), 2);
}

struct FwdLists { std::vector<int> off, dst, slot; std::vector<float> lw; };

// A site's fwd list = its outgoing edges in creation order (Edge::index); the log weight is the bwd CSR's.
FwdLists forward_lists(const pagan_graph *g) {
    struct E { int src, dst, eid; float lw; };
    std::vector<E> es;
    es.reserve(g->bwd_off[g->n_sites]);
    for (int s = 0; s < g->n_sites; ++s)
        for (int k = g->bwd_off[s]; k < g->bwd_off[s + 1]; ++k) es.push_back({g->bwd_src[k], s, g->bwd_eid[k], g->bwd_logw[k]});
    std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.src != b.src ? a.src < b.src : a.eid < b.eid; });
    FwdLists f;
    f.off.assign(g->n_sites + 1, 0);
    for (const E &e : es) f.off[e.src + 1]++;
    for (int s = 0; s < g->n_sites; ++s) f.off[s + 1] += f.off[s];
    for (const E &e : es) { f.dst.push_back(e.dst); f.lw.push_back(e.lw); }
    return f;
}

} // namespace

int pagan_internal_replay(const pagan_graph *L, const pagan_graph *R, int64_t cells, const int *endcell, double endscore,
                          const int *trace, pagan_result *out);     // dp_abi.hip

struct pagan_fb {
    int device = 0;
    int Lx = 0, Ly = 0, S = 0;
    const pagan_graph *L = nullptr, *R = nullptr;       // borrowed: must outlive the handle for sample_path
    std::vector<float> score;                           // copy of the probability table
    float gap_open = 0, gap_ext = 0, non_gap = 0;
    DiagIndex dx;
    RowBand rb;
    char *arena = nullptr;
    double *dF = nullptr, *dB = nullptr;
    double totals[2] = {0, 0};
    float kernel_ms[2] = {0, 0};                        // pg_fb_forward, pg_fb_backward (HIP events)
    std::vector<double> hF;                             // downloaded lazily
    long long at(int i, int j) const {
        if (i < 0 || j < 0 || i >= Lx || j >= Ly) return -1;
        const int d = i + j;
        return (i >= dx.imin[d] && i <= dx.imax[d]) ? dx.doff[d] + (i - dx.imin[d]) : -1;
    }
};

extern "C" {

int pagan_fb_run(const pagan_graph *left, const pagan_graph *right, const pagan_model_prob *model, const pagan_band *band,
                 const pagan_opts *opts, pagan_fb **out) {
    if (!left || !right || !model || !out || !model->score || model->n_states < 1) return PAGAN_E_ARG;
    int rc = check_graph(left);
    if (rc == PAGAN_OK) rc = check_graph(right);
    if (rc != PAGAN_OK) return rc;
    const int Lx = left->n_sites - 1, Ly = right->n_sites - 1, S = model->n_states;
    for (int s = 1; s < Lx; ++s) if (left->state[s] < 0 || left->state[s] >= S) return PAGAN_E_MODEL;
    for (int s = 1; s < Ly; ++s) if (right->state[s] < 0 || right->state[s] >= S) return PAGAN_E_MODEL;
    if (!(model->gap_open > 0) || !(model->gap_ext > 0) || !(model->non_gap > 0)) return PAGAN_E_MODEL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAGAN_E_NODEVICE;
    pagan_fb *fb = new pagan_fb();
    std::unique_ptr<pagan_fb, void (*)(pagan_fb *)> guard(fb, [](pagan_fb *p) { pagan_fb_destroy(p); });
    if (opts && opts->device >= 0) { FB_TRY(hipSetDevice(opts->device)); fb->device = opts->device; }
    else FB_TRY(hipGetDevice(&fb->device));
    fb->Lx = Lx; fb->Ly = Ly; fb->S = S; fb->L = left; fb->R = right;
    fb->score.assign(model->score, model->score + (size_t)S * S);
    fb->gap_open = model->gap_open; fb->gap_ext = model->gap_ext; fb->non_gap = model->non_gap;
    rc = fb->rb.build(Lx, Ly, band);
    if (rc != PAGAN_OK) return rc;
    fb->dx.build(Lx, Ly, fb->rb);
    const long long cells = fb->dx.cells;
    const int nd = Lx + Ly - 1;
    const FwdLists fl = forward_lists(left), fr = forward_lists(right);
    std::vector<double> ltab((size_t)S * S);
    for (size_t k = 0; k < ltab.size(); ++k) ltab[k] = std::log((double)model->score[k]);
    // initialise_array_corner_bwd (VA:740-854): assignments onto the cells the end corner reads
    std::vector<long long> init_at;
    std::vector<double> init_val;
    const double l_ng = std::log((double)model->non_gap);
    auto put_init = [&](int i, int j, int s, double v) {
        const long long a = fb->at(i, j);
        if (a < 0) return;
        for (size_t k = 0; k < init_at.size(); ++k) if (init_at[k] == 3 * a + s) { init_val[k] = v; return; }   // later assignment wins
        init_at.push_back(3 * a + s); init_val.push_back(v);
    };
    put_init(Lx - 1, Ly - 1, PAGAN_M_MAT, l_ng);
    {
        const int l0 = left->bwd_off[Lx], l1 = left->bwd_off[Lx + 1], r0 = right->bwd_off[Ly], r1 = right->bwd_off[Ly + 1];
        if (l1 > l0 && r1 > r0)
            for (int k1 = l0; k1 < l1; ++k1)
                for (int k2 = r0; k2 < r1; ++k2)
                    put_init(left->bwd_src[k1], right->bwd_src[k2], PAGAN_M_MAT, l_ng + (double)left->bwd_logw[k1] + (double)right->bwd_logw[k2]);
        for (int k1 = l0; k1 < l1; ++k1) put_init(left->bwd_src[k1], Ly - 1, PAGAN_X_MAT, 0.0);
        for (int k2 = r0; k2 < r1; ++k2) put_init(Lx - 1, right->bwd_src[k2], PAGAN_Y_MAT, 0.0);
    }
    // one arena: inputs, then F and B
    size_t cur = 0;
    auto take = [&](size_t bytes) { const size_t o = cur; cur = (cur + bytes + 255) / 256 * 256; return o; };
    const int nbL = left->bwd_off[left->n_sites], nbR = right->bwd_off[right->n_sites];
    const size_t o_job = take(sizeof(PgFbJob));
    const size_t o_stL = take(4 * (size_t)left->n_sites), o_offL = take(4 * ((size_t)left->n_sites + 1)), o_srcL = take(4 * (size_t)nbL), o_lwL = take(4 * (size_t)nbL);
    const size_t o_stR = take(4 * (size_t)right->n_sites), o_offR = take(4 * ((size_t)right->n_sites + 1)), o_srcR = take(4 * (size_t)nbR), o_lwR = take(4 * (size_t)nbR);
    const size_t o_foL = take(4 * fl.off.size()), o_fdL = take(4 * fl.dst.size()), o_fwL = take(4 * fl.lw.size());
    const size_t o_foR = take(4 * fr.off.size()), o_fdR = take(4 * fr.dst.size()), o_fwR = take(4 * fr.lw.size());
    const size_t o_tab = take(8 * ltab.size());
    const size_t o_imin = take(4 * (size_t)nd), o_imax = take(4 * (size_t)nd), o_doff = take(8 * (size_t)nd);
    const size_t o_iat = take(8 * init_at.size()), o_ival = take(8 * init_val.size());
    const size_t o_tot = take(16);
    const size_t in_bytes = cur;
    const size_t o_F = take(24 * (size_t)cells), o_B = take(24 * (size_t)cells);
    FB_TRY(hipMalloc((void **)&fb->arena, cur));
    std::vector<char> stage(in_bytes, 0);
    auto put = [&](size_t off, const void *src, size_t bytes) { if (bytes) std::memcpy(stage.data() + off, src, bytes); };
    put(o_stL, left->state, 4 * (size_t)left->n_sites); put(o_offL, left->bwd_off, 4 * ((size_t)left->n_sites + 1));
    put(o_srcL, left->bwd_src, 4 * (size_t)nbL); put(o_lwL, left->bwd_logw, 4 * (size_t)nbL);
    put(o_stR, right->state, 4 * (size_t)right->n_sites); put(o_offR, right->bwd_off, 4 * ((size_t)right->n_sites + 1));
    put(o_srcR, right->bwd_src, 4 * (size_t)nbR); put(o_lwR, right->bwd_logw, 4 * (size_t)nbR);
    put(o_foL, fl.off.data(), 4 * fl.off.size()); put(o_fdL, fl.dst.data(), 4 * fl.dst.size()); put(o_fwL, fl.lw.data(), 4 * fl.lw.size());
    put(o_foR, fr.off.data(), 4 * fr.off.size()); put(o_fdR, fr.dst.data(), 4 * fr.dst.size()); put(o_fwR, fr.lw.data(), 4 * fr.lw.size());
    put(o_tab, ltab.data(), 8 * ltab.size());
    put(o_imin, fb->dx.imin.data(), 4 * (size_t)nd); put(o_imax, fb->dx.imax.data(), 4 * (size_t)nd); put(o_doff, fb->dx.doff.data(), 8 * (size_t)nd);
    put(o_iat, init_at.data(), 8 * init_at.size()); put(o_ival, init_val.data(), 8 * init_val.size());
    char *b = fb->arena;
    PgFbJob J;
    J.Lx = Lx; J.Ly = Ly; J.nd = nd; J.S = S;
    J.l_ext = std::log((double)model->gap_ext); J.l_open = std::log((double)model->gap_open); J.l_ng = l_ng;
    J.stL = (const int *)(b + o_stL); J.offL = (const int *)(b + o_offL); J.srcL = (const int *)(b + o_srcL); J.lwL = (const float *)(b + o_lwL);
    J.stR = (const int *)(b + o_stR); J.offR = (const int *)(b + o_offR); J.srcR = (const int *)(b + o_srcR); J.lwR = (const float *)(b + o_lwR);
    J.foffL = (const int *)(b + o_foL); J.fdstL = (const int *)(b + o_fdL); J.flwL = (const float *)(b + o_fwL);
    J.foffR = (const int *)(b + o_foR); J.fdstR = (const int *)(b + o_fdR); J.flwR = (const float *)(b + o_fwR);
    J.ltab = (const double *)(b + o_tab);
    J.imin = (const int *)(b + o_imin); J.imax = (const int *)(b + o_imax); J.doff = (const long long *)(b + o_doff);
    J.cells = cells;
    J.F = (double *)(b + o_F); J.B = (double *)(b + o_B);
    J.n_init = (int)init_at.size(); J.init_at = (const long long *)(b + o_iat); J.init_val = (const double *)(b + o_ival);
    J.totals = (double *)(b + o_tot);
    std::memcpy(stage.data() + o_job, &J, sizeof(J));
    fb->dF = J.F; fb->dB = J.B;
    FB_TRY(hipMemcpy(fb->arena, stage.data(), in_bytes, hipMemcpyHostToDevice));
    // a thread per cell of the widest diagonal, up to the 1024 of a workgroup (a cell is ~20 exp / log1p calls: a thread with
    // eight cells of a 2,000-cell diagonal was the whole sweep's pace)
    const int mw = fb->dx.max_width;
    const int block = mw >= 768 ? 1024 : mw >= 384 ? 512 : mw >= 192 ? 256 : (mw >= 96 ? 128 : 64);
    // the two sweeps are independent of each other: side by side on two streams (a workgroup each)
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    FB_TRY(hipStreamCreate(&s1)); FB_TRY(hipStreamCreate(&s2));
    FB_TRY(hipEventCreate(&e0)); FB_TRY(hipEventCreate(&e1)); FB_TRY(hipEventCreate(&e2)); FB_TRY(hipEventCreate(&e3));
    FB_TRY(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(pg_fb_forward, dim3(1), dim3(block), 0, s1, (const PgFbJob *)(b + o_job));
    FB_TRY(hipEventRecord(e1, s1));
    FB_TRY(hipEventRecord(e2, s2));
    hipLaunchKernelGGL(pg_fb_backward, dim3(1), dim3(block), 0, s2, (const PgFbJob *)(b + o_job));
    FB_TRY(hipEventRecord(e3, s2));
    FB_TRY(hipGetLastError());
    FB_TRY(hipStreamSynchronize(s1)); FB_TRY(hipStreamSynchronize(s2));
    (void)hipEventElapsedTime(&fb->kernel_ms[0], e0, e1);
    (void)hipEventElapsedTime(&fb->kernel_ms[1], e2, e3);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
    FB_TRY(hipMemcpy(fb->totals, b + o_tot, 16, hipMemcpyDeviceToHost));
    *out = guard.release();
    return PAGAN_OK;
}

int pagan_fb_kernel_ms(const pagan_fb *fb, double ms[2]) {
    if (!fb || !ms) return PAGAN_E_ARG;
    ms[0] = fb->kernel_ms[0]; ms[1] = fb->kernel_ms[1];
    return PAGAN_OK;
}

int pagan_fb_totals(const pagan_fb *fb, double *log_fwd, double *log_bwd, int64_t *cells) {
    if (!fb) return PAGAN_E_ARG;
    if (log_fwd) *log_fwd = fb->totals[0];
    if (log_bwd) *log_bwd = fb->totals[1];
    if (cells) *cells = fb->dx.cells;
    return PAGAN_OK;
}

// which: 0 = log forward, 1 = log backward, 2 = posterior (compute_posterior_score, VA:1029-1034).
// dst [Lx][Ly][3] row-major, state order X, Y, M; outside the tunnel -inf (logs) / 0 (posterior).
int pagan_fb_dump(pagan_fb *fb, int32_t which, double *dst) {
    if (!fb || !dst || which < 0 || which > 2) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    const size_t n = 3 * (size_t)fb->dx.cells;
    std::vector<double> a(n), bb;
    FB_TRY(hipMemcpy(a.data(), which == 1 ? fb->dB : fb->dF, 8 * n, hipMemcpyDeviceToHost));
    if (which == 2) { bb.resize(n); FB_TRY(hipMemcpy(bb.data(), fb->dB, 8 * n, hipMemcpyDeviceToHost)); }
    const double outside = which == 2 ? 0.0 : -HUGE_VAL;
    for (int i = 0; i < fb->Lx; ++i)
        for (int j = 0; j < fb->Ly; ++j) {
            const long long at = fb->at(i, j);
            double *o = dst + ((size_t)i * fb->Ly + j) * 3;
            for (int s = 0; s < 3; ++s)
                o[s] = at < 0 ? outside : which == 2 ? std::exp(a[3 * at + s] + bb[3 * at + s] - fb->totals[0]) : a[3 * at + s];
        }
    return PAGAN_OK;
}

// Posterior of n cells given as (state, i, j) triples.
int pagan_fb_posterior_cells(pagan_fb *fb, int32_t n, const int32_t *cells, double *post) {
    if (!fb || n < 0 || (n > 0 && (!cells || !post))) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    for (int k = 0; k < n; ++k) {
        const int s = cells[3 * k];
        const long long at = fb->at(cells[3 * k + 1], cells[3 * k + 2]);
        if (s < 0 || s > 2) return PAGAN_E_ARG;
        if (at < 0) { post[k] = 0.0; continue; }
        double f = 0, b = 0;
        FB_TRY(hipMemcpy(&f, fb->dF + 3 * at + s, 8, hipMemcpyDeviceToHost));
        FB_TRY(hipMemcpy(&b, fb->dB + 3 * at + s, 8, hipMemcpyDeviceToHost));
        post[k] = std::exp(f + b - fb->totals[0]);
    }
    return PAGAN_OK;
}

// sample_new_path (VA:1193-1322): a path drawn from the posterior by walking back from the end corner; at every step
// the predecessors are listed in the forward pass's candidate order with weight fwd(pred) * transition (add_sample_*,
// VA:2309-2446) and the first one whose running sum reaches u * total is taken (VA:1757-1769).  u[k] in [0, 1) stands
// for rand()/(RAND_MAX+1): one number per step, the end corner first (at most Lx + Ly + 1 are used).  The result has
// the shape of a Viterbi result (columns, used edges; score = log of the full probability).  visited (optional,
// 3 * (Lx + Ly) ints): the cells of the path end -> start as (i, j, state).
int pagan_fb_sample_path(pagan_fb *fb, const double *u, int32_t n_u, pagan_result *out, int32_t *visited, int32_t *n_visited) {
    if (!fb || !u || !out) return PAGAN_E_ARG;
    FB_TRY(hipSetDevice(fb->device));
    if (fb->hF.empty()) {
        fb->hF.resize(3 * (size_t)fb->dx.cells);
        FB_TRY(hipMemcpy(fb->hF.data(), fb->dF, 8 * fb->hF.size(), hipMemcpyDeviceToHost));
    }
    const pagan_graph *L = fb->L, *R = fb->R;
    const int Lx = fb->Lx, Ly = fb->Ly, S = fb->S;
    auto F = [&](int s, int i, int j) { const long long a = fb->at(i, j); return a < 0 ? -HUGE_VAL : fb->hF[3 * a + s]; };
    const double ext = std::log((double)fb->gap_ext), open = std::log((double)fb->gap_open), ng = std::log((double)fb->non_gap);
    struct Cand { double lw; int state, i, j, k1, k2; };
    std::vector<Cand> c;
    auto pick = [&](double uu) -> int {
        double hi = -HUGE_VAL;
        for (const Cand &x : c) hi = std::max(hi, x.lw);
        if (c.empty() || hi == -HUGE_VAL) return -1;
        double tot = 0;
        for (const Cand &x : c) tot += std::exp(x.lw - hi);
        const double rv = tot * uu;
        size_t k = 0;
        double sum = std::exp(c[0].lw - hi);
        while (sum < rv && k + 1 < c.size()) { ++k; sum += std::exp(c[k].lw - hi); }
        return (int)k;
    };
    int used = 0;
    {   // iterate_bwd_edges_for_sampled_end_corner, VA:1904-2025
        const int l0 = L->bwd_off[Lx], l1 = L->bwd_off[Lx + 1], r0 = R->bwd_off[Ly], r1 = R->bwd_off[Ly + 1];
        auto mt = [&](int k1, int k2) { c.push_back({F(2, L->bwd_src[k1], R->bwd_src[k2]) + ng + (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2], 2, L->bwd_src[k1], R->bwd_src[k2], k1 - l0, k2 - r0}); };
        auto xc = [&](int k1) { c.push_back({F(0, L->bwd_src[k1], Ly - 1), 0, L->bwd_src[k1], Ly - 1, k1 - l0, -1}); };
        auto yc = [&](int k2) { c.push_back({F(1, Lx - 1, R->bwd_src[k2]), 1, Lx - 1, R->bwd_src[k2], -1, k2 - r0}); };
        if (l1 > l0 && r1 > r0) {
            mt(l0, r0); xc(l0); yc(r0);
            for (int k2 = r0 + 1; k2 < r1; ++k2) { mt(l0, k2); yc(k2); }
            for (int k1 = l0 + 1; k1 < l1; ++k1) { mt(k1, r0); xc(k1); for (int k2 = r0 + 1; k2 < r1; ++k2) { mt(k1, k2); yc(k2); } }
        }
    }
    if (used >= n_u) return PAGAN_E_ARG;
    int k = pick(u[used++]);
    int endcell[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < 0) {                                     // full probability 0: nothing to sample
        endcell[0] = 1; endcell[4] = endcell[5] = -1;
        return pagan_internal_replay(L, R, fb->dx.cells, endcell, fb->totals[0], nullptr, out);
    }
    int state = c[k].state, i = c[k].i, j = c[k].j;
    endcell[1] = state; endcell[2] = i; endcell[3] = j; endcell[4] = c[k].k1; endcell[5] = c[k].k2;
    std::vector<int> trace;
    trace.reserve(3 * ((size_t)Lx + Ly));
    while (!(i < 1 && j < 1)) {
        c.clear();
        if (state == 2) {
            const double sc = std::log((double)fb->score[L->state[i] + (size_t)R->state[j] * S]);
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; ++k1)
                for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; ++k2) {
                    const int p = L->bwd_src[k1], q = R->bwd_src[k2];
                    const double w = (double)L->bwd_logw[k1] + (double)R->bwd_logw[k2];
                    const int a = k1 - L->bwd_off[i], b = k2 - R->bwd_off[j];
                    c.push_back({F(2, p, q) + ng + ng + sc + w, 2, p, q, a, b});
                    c.push_back({F(0, p, q) + ng + sc + w, 0, p, q, a, b});
                    c.push_back({F(1, p, q) + ng + sc + w, 1, p, q, a, b});
                }
        } else if (state == 0) {
            for (int k1 = L->bwd_off[i]; k1 < L->bwd_off[i + 1]; ++k1) {
                const int p = L->bwd_src[k1], a = k1 - L->bwd_off[i];
                c.push_back({F(0, p, j) + ext, 0, p, j, a, 0}); c.push_back({F(1, p, j) + open, 1, p, j, a, 0});
                c.push_back({F(2, p, j) + ng + open, 2, p, j, a, 0});
            }
        } else {
            for (int k2 = R->bwd_off[j]; k2 < R->bwd_off[j + 1]; ++k2) {
                const int q = R->bwd_src[k2], b = k2 - R->bwd_off[j];
                c.push_back({F(1, i, q) + ext, 1, i, q, 0, b}); c.push_back({F(0, i, q) + open, 0, i, q, 0, b});
                c.push_back({F(2, i, q) + ng + open, 2, i, q, 0, b});
            }
        }
        if (used >= n_u) return PAGAN_E_ARG;
        k = pick(u[used++]);
        if (k < 0 || (size_t)trace.size() >= 3 * ((size_t)Lx + Ly)) return PAGAN_E_INTERNAL;
        trace.push_back(i); trace.push_back(j);
        trace.push_back((int)((unsigned)state | ((unsigned)c[k].k1 << 4) | ((unsigned)c[k].k2 << 18)));
        state = c[k].state; i = c[k].i; j = c[k].j;
    }
    endcell[6] = (int)(trace.size() / 3);
    if (visited) for (size_t t = 0; t < trace.size() / 3; ++t) { visited[3 * t] = trace[3 * t]; visited[3 * t + 1] = trace[3 * t + 1]; visited[3 * t + 2] = trace[3 * t + 2] & 3; }
    if (n_visited) *n_visited = endcell[6];
    trace.resize(trace.size() + 3, 0);
    return pagan_internal_replay(L, R, fb->dx.cells, endcell, fb->totals[0], trace.data(), out);
}

void pagan_fb_destroy(pagan_fb *fb) {
    if (!fb) return;
    if (fb->arena) { (void)hipSetDevice(fb->device); (void)hipFree(fb->arena); }
    delete fb;
}

} // extern "C"
