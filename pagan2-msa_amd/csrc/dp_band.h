// dp_band.h -- host-side helpers shared by the C-ABI layers (dp_abi.hip, dp_fb.hip): graph validation,
// the tunnel as a clamped row band, and its per-anti-diagonal form (the device's cell index).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pagan_dp.h"
#include "dp_device.h"

namespace {

int check_graph(const pagan_graph *g) {
    if (!g || g->n_sites < 2 || g->n_edges < 0 || !g->state || !g->bwd_off) return PAGAN_E_GRAPH;
    if (g->bwd_off[0] != 0) return PAGAN_E_GRAPH;
    for (int s = 0; s < g->n_sites; ++s) {
        const int a = g->bwd_off[s], b = g->bwd_off[s + 1];
        if (b < a || b - a > PG_MAX_SLOT) return PAGAN_E_GRAPH;
        if (b > a && (!g->bwd_src || !g->bwd_logw || !g->bwd_eid)) return PAGAN_E_GRAPH;
        for (int k = a; k < b; ++k) {
            if (g->bwd_src[k] < 0 || g->bwd_src[k] >= s) return PAGAN_E_GRAPH;      // edges point forward
            if (g->bwd_eid[k] < 0 || g->bwd_eid[k] >= g->n_edges) return PAGAN_E_GRAPH;
        }
    }
    return PAGAN_OK;
}

// Row band clamped as Tunnel_matrix does (src/utils/tunnel_matrix.h:194).
struct RowBand {
    std::vector<int> lo, hi;
    int build(int Lx, int Ly, const pagan_band *band) {
        lo.assign(Lx, 0);
        hi.assign(Lx, Ly - 1);
        if (band) {
            if (band->n < Lx || !band->upper || !band->lower) return PAGAN_E_BAND;
            for (int i = 0; i < Lx; ++i) {
                lo[i] = band->upper[i] > 0 ? band->upper[i] : 0;
                hi[i] = band->lower[i] < Ly - 1 ? band->lower[i] : Ly - 1;
            }
            // The tunnel must be monotone (tunnel_matrix.h:162-164) and must hold the start
            // corner, where the reference writes M[0][0] = 0 (VA:725-736).
            if (lo[0] > 0 || hi[0] < 0) return PAGAN_E_BAND;
            for (int i = 1; i < Lx; ++i)
                if (lo[i] < lo[i - 1] || hi[i] < hi[i - 1]) return PAGAN_E_BAND;
        }
        return PAGAN_OK;
    }
    int64_t cells() const {
        int64_t c = 0;
        for (size_t i = 0; i < lo.size(); ++i) if (hi[i] >= lo[i]) c += hi[i] - lo[i] + 1;
        return c;
    }
};

// Per anti-diagonal d = i+j: the in-band rows form one interval [imin,imax] because
// lo[i]+i and hi[i]+i are strictly increasing for a monotone band.
struct DiagIndex {
    std::vector<int> imin, imax;
    std::vector<long long> doff;
    long long cells = 0;
    int max_width = 0;
    void build(int Lx, int Ly, const RowBand &rb) {
        const int nd = Lx + Ly - 1;
        imin.resize(nd); imax.resize(nd); doff.resize(nd);
        int a = -1, b = 0;       // a = max{i: lo[i]+i <= d}, b = min{i: hi[i]+i >= d}
        cells = 0; max_width = 0;
        for (int d = 0; d < nd; ++d) {
            while (a + 1 < Lx && rb.lo[a + 1] + (a + 1) <= d) ++a;
            while (b < Lx && rb.hi[b] + b < d) ++b;
            imin[d] = b; imax[d] = a; doff[d] = cells;
            const int w = a - b + 1;
            if (w > 0) { cells += w; if (w > max_width) max_width = w; }
        }
    }
};

} // namespace
