"""Host logic of the row strips (dp_abi.hip: plan_strips), no device: the strips of a wide job cover every cell of the job
exactly once and put every score where the job's diagonal index has it; a strip's rows never fall from one diagonal to the
next; strip k > 0 is fed by the compute wave above its first row; the general steps are the few the design names; the bound on
a diagonal's multi-edge sites refuses dense jobs."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, synth


def band_of(Lx, Ly, half, seed):
    rng = np.random.default_rng(seed)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    h = rng.integers(half // 2, half, Lx)
    upper = np.maximum.accumulate(np.maximum(centre - h, 0))
    lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    return abi.Band(upper, lower), upper, lower


def diag_index(Lx, Ly, lo, hi):
    nd = Lx + Ly - 1
    imin, imax, doff = np.zeros(nd, int), np.zeros(nd, int), np.zeros(nd + 1, int)
    a, b = -1, 0
    for d in range(nd):
        while a + 1 < Lx and lo[a + 1] + a + 1 <= d:
            a += 1
        while b < Lx and hi[b] + b < d:
            b += 1
        imin[d], imax[d] = b, a
        doff[d + 1] = doff[d] + max(0, a - b + 1)
    return imin, imax, doff


@pytest.mark.parametrize("case", ["full", "full_tall", "band"])
def test_strips_cover_the_job_once(pg, case):
    if case == "band":
        left = synth.random_graph(1500, 15, 1, p_extra=0.05, max_deg=3, max_span=20)
        right = synth.random_graph(1400, 15, 2, p_extra=0.05, max_deg=3, max_span=20)
        Lx, Ly = left.n_sites - 1, right.n_sites - 1
        band, lo, hi = band_of(Lx, Ly, 420, 3)
    else:
        nl, nr = (700, 330) if case == "full" else (1000, 90)
        left = synth.random_graph(nl, 15, 4, p_extra=0.05, max_deg=3, max_span=9)
        right = synth.random_graph(nr, 15, 5, p_extra=0.05, max_deg=3, max_span=9)
        Lx, Ly = left.n_sites - 1, right.n_sites - 1
        band, lo, hi = None, np.zeros(Lx, int), np.full(Lx, Ly - 1)
    imin, imax, doff = diag_index(Lx, Ly, lo, hi)
    strips = pg.debug_strips(left, right, band)
    assert len(strips) == (Lx + 191) // 192
    seen = np.zeros(doff[-1], np.int32)
    for k, (r0, r1, d0, d1, feed, c0, desc) in enumerate(strips):
        assert r0 == 192 * k and r1 == min(r0 + 191, Lx - 1)
        assert feed == (-1 if k == 0 else (r0 // 64 + 3) % 4), "the feeder is the wave above the strip's first row"
        assert 0 <= d0 < d1 <= Lx + Ly - 1 and len(desc) == d1 - d0
        assert c0 % 64 == 0
        first = desc[:, 0]
        assert np.all(np.diff(first) >= 0), "a strip's first row never falls"
        for t in range(d1 - d0):
            a, b, cell, cls = (int(v) for v in desc[t])
            d = d0 + t
            if b < a:
                continue
            assert r0 <= a and b <= r1 and imin[d] <= a and b <= imax[d]
            assert cell == doff[d] + (a - imin[d]), "scores go where the job's diagonal index has them"
            seen[cell: cell + (b - a + 1)] += 1
            if d <= 1:
                assert cls == 3
        # the first column staged lies at or before every column the strip touches
        cols = [d0 + t - int(desc[t, 1]) for t in range(d1 - d0) if desc[t, 1] >= desc[t, 0]]
        assert min(cols) >= c0
    assert np.all(seen == 1), "every cell of the job belongs to exactly one strip"


def test_general_steps_are_few(pg):
    """no multi-edge site: the general steps are the diagonals 0 and 1 (edges from site 0 meet M(0,0) there)"""
    left = synth.random_graph(500, 15, 6, p_extra=0.0)
    right = synth.random_graph(450, 15, 7, p_extra=0.0)
    strips = pg.debug_strips(left, right)
    general = sum(int(np.sum((desc[:, 3] == 3) & (desc[:, 1] >= desc[:, 0]))) for *_, desc in strips)
    assert general == 2
    # (random_graph gives half of its edges a weight: such a site is an "easy" multi-edge site, evaluated by the lanes -- class 1)
    lanes = sum(int(np.sum((desc[:, 3] <= 1) & (desc[:, 1] >= desc[:, 0]))) for *_, desc in strips)
    staged = sum(int(np.sum((desc[:, 3] == 2) & (desc[:, 1] >= desc[:, 0]))) for *_, desc in strips)
    assert lanes > 1800 and staged == 0, "a strip's other diagonals are the lanes' own (no assist wave, no general step)"


def test_the_site_bound_refuses_dense_jobs(pg):
    left = synth.random_graph(600, 15, 8, p_extra=0.4, max_deg=4, max_span=8)
    right = synth.random_graph(600, 15, 9, p_extra=0.4, max_deg=4, max_span=8)
    assert pg.debug_strips(left, right, max_sites=56) == []
    assert len(pg.debug_strips(left, right, max_sites=0)) == 4
    model = synth.random_model(15, 1)
    assert pg.debug_route(left, right, model)[0] == "pg_fill_tiles_flow"

    def unweighted(g):       # (random_graph gives half of its edges a weight, and a weighted edge makes a site a multi-edge one)
        return abi.Graph(g.state, g.bwd_off, g.bwd_src, np.zeros_like(g.bwd_logw), g.bwd_eid, n_edges=g.n_edges)

    sparse_l = unweighted(synth.random_graph(600, 15, 8, p_extra=0.02, max_deg=3, max_span=8))
    sparse_r = unweighted(synth.random_graph(600, 15, 9, p_extra=0.02, max_deg=3, max_span=8))
    assert pg.debug_route(sparse_l, sparse_r, model)[0] == "pg_fill_pipe (row strips)"
