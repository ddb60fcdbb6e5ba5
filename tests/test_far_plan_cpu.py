"""Far histories and three-edge sites in the banded kernel's plan (dp_abi.hip: plan_far_hist, classify_diagonals; round 5).
No GPU: pagan_dp_debug_far is host code.  Invariants the kernel's hist_tail / third_pass rest on:

  * a served far site is an easy two-edge site whose other edge reaches REACH - 1 .. PG_HIST_MAX_SPAN sites back; its start site
    carries the writer flag with the SAME line; two pairs whose intervals overlap never share a line unless they share the
    start site;
  * the flagged diagonals (bit 0 of hbit) cover every diagonal from the start site's first cell to the far site's last, and all
    of them run where the writers' cells are appended to the line: the hand-scheduled loop (class <= 2), a wide run (class 4:
    wide_run7 / wide_run) or one of the two general steps behind one (hist_append) -- never a class 5 diagonal;
  * a cell where a served far site meets a site with an other edge of its own lies on a class 2 diagonal (the pair of the two
    other edges is nobody's in the lanes);
  * the third-pass bit is only set on class 1 diagonals that hold a three-edge site with one edge from the previous site and
    both others inside the ring's reach; a cell where two such sites meet lies on a class 2 diagonal;
  * PAGAN_DP_HIST=0 / PAGAN_DP_THREE=0 give the plan of pagan_dp_debug_plan (the planner without either)."""
import numpy as np
import pytest

import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth
from test_plan_cpu import REACH

MAX_SPAN = 44          # dp_device.h: PG_HIST_MAX_SPAN


def job(seed, n=2500, p_extra=0.03, max_deg=3, max_span=30, half=(6, 40)):
    rng = np.random.default_rng(seed)
    left = synth.random_graph(n, 15, 100 + seed, p_extra=p_extra, max_deg=max_deg, max_span=max_span)
    right = synth.random_graph(n + int(rng.integers(-60, 60)), 15, 200 + seed, p_extra=p_extra, max_deg=max_deg, max_span=max_span)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    hw = rng.integers(half[0], half[1], Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - hw, 0)); lower = np.maximum.accumulate(np.minimum(centre + hw, Ly - 1))
    upper[0] = 0; lower[-1] = Ly - 1
    return left, right, abi.Band(upper, lower)


def dists(g, s):
    return [int(s - p) for p in g.bwd_src[g.bwd_off[s]:g.bwd_off[s + 1]]]


def band_index(left, right, band):
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    lo = np.maximum(band.upper[:Lx].astype(np.int64), 0); hi = np.minimum(band.lower[:Lx].astype(np.int64), Ly - 1)
    return Lx, Ly, lo, hi


@pytest.mark.parametrize("seed", range(6))
def test_far_history_plan_invariants(seed):
    left, right, band = job(seed)
    n, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    Lx, Ly, lo, hi = band_index(left, right, band)
    assert n > 0, "the job is meant to have far sites"
    readers = [(True, int(i)) for i in np.nonzero(hfl & 0x80)[0]] + [(False, int(j)) for j in np.nonzero(hfr & 0x80)[0]]
    assert len(readers) == n
    intervals = []
    for is_left, s in readers:
        g, hf = (left, hfl) if is_left else (right, hfr)
        ds = dists(g, s)
        assert len(ds) == 2 and sorted(ds)[0] == 1, "a served site is an easy two-edge site"
        k = max(ds)
        assert REACH - 1 <= k <= MAX_SPAN
        src = s - k
        assert src >= 1 and (hf[src] & 0x40), "its start site is a writer"
        assert (hf[src] >> 4) & 3 == hf[s] & 3, "... of the same line"
        if is_left:
            d0, d1 = src + lo[src], s + hi[s]
        else:
            rows_src = np.nonzero((lo <= src) & (src <= hi))[0]; rows_s = np.nonzero((lo <= s) & (s <= hi))[0]
            d0, d1 = int(rows_src[0]) + src, int(rows_s[-1]) + s
        assert (hb[d0:d1 + 1] & 1).all(), "every diagonal of the pair's life is flagged"
        assert (cls[d0:d1 + 1] != 5).all(), "... and none of them is a class 5 diagonal (nobody appends there)"
        for d in np.nonzero(cls[d0:d1 + 1] == 3)[0] + d0:
            assert (cls[max(d - 2, 0):d] == 4).any(), "a general step inside an interval is one of the two behind a wide run"
        intervals.append((d0, d1, int(hf[s] & 3), is_left, src))
        # crossings: where the far site meets a site with an other edge of its own
        if is_left:
            for j in range(lo[s], hi[s] + 1):
                if max(dists(right, j) or [0]) >= 2:
                    assert cls[s + j] >= 2
        else:
            for i in np.nonzero((lo <= s) & (s <= hi))[0]:
                if max(dists(left, int(i)) or [0]) >= 2:
                    assert cls[int(i) + s] >= 2
    for a in range(len(intervals)):
        for b in range(a + 1, len(intervals)):
            x, y = intervals[a], intervals[b]
            if x[2] == y[2] and not (x[1] < y[0] or y[1] < x[0]):
                assert x[3] == y[3] and x[4] == y[4], "two pairs share a line at the same time only through their start site"


@pytest.mark.parametrize("seed", range(3))
def test_an_interval_may_cross_a_wide_run(seed, monkeypatch):
    """A box wider than the lanes in the middle of the band: far sites whose life crosses its wide run keep their line (the wide
    runs append the writers' cells); with PAGAN_DP_HIST=narrow, as before round 5's item 11, they have none."""
    rng = np.random.default_rng(50 + seed)
    n = 1800
    left = synth.random_graph(n, 15, 300 + seed, p_extra=0.03, max_deg=3, max_span=30)
    right = synth.random_graph(n, 15, 400 + seed, p_extra=0.03, max_deg=3, max_span=30)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum(centre - 25, 0); lower = np.minimum(centre + 25, Ly - 1)
    upper[600:900] = upper[600]; lower[600:900] = lower[899] + 30
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    upper[0] = 0; lower[-1] = Ly - 1
    band = abi.Band(upper, lower)
    n_all, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    cls = cls & 15
    assert (cls == 4).sum() > 100 and not (cls == 5).any()
    crossing = int(((hb & 1) != 0)[cls == 4].sum())
    assert crossing > 0, "some served interval is meant to cross the wide run"
    monkeypatch.setenv("PAGAN_DP_HIST", "narrow")
    n_narrow, _, _, hb2, cls2 = pg.debug_far(left, right, band)
    assert n_narrow < n_all and not ((hb2 & 1) != 0)[(cls2 & 15) == 4].any()


@pytest.mark.parametrize("seed", range(4))
def test_third_pass_bit(seed):
    left, right, band = job(40 + seed, p_extra=0.05, max_deg=4, max_span=9)
    n, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    Lx, Ly, lo, hi = band_index(left, right, band)
    assert (hb & 2).any(), "the job is meant to have three-edge sites on class 1 diagonals"

    def three(g, s):
        ds = dists(g, s)
        return s >= 1 and len(ds) == 3 and ds.count(1) == 1 and max(ds) <= REACH - 2

    tl = np.array([three(left, i) for i in range(Lx)]); tr = np.array([three(right, j) for j in range(Ly)])
    for d in np.nonzero(hb & 2)[0]:
        assert cls[d] == 1
        rows = [i for i in range(max(0, d - Ly + 1), min(Lx, d + 1)) if lo[i] <= d - i <= hi[i]]
        assert any(tl[i] or tr[d - i] for i in rows)
    for i in np.nonzero(tl)[0]:
        for j in range(lo[i], hi[i] + 1):
            if tr[j]:
                assert cls[i + j] >= 2, "a cell where two three-edge sites meet stays with the assist waves"


def test_switches_give_the_plan_without_either(monkeypatch):
    left, right, band = job(3)
    base, _ = pg.debug_plan(left, right, band)
    monkeypatch.setenv("PAGAN_DP_HIST", "0")
    monkeypatch.setenv("PAGAN_DP_THREE", "0")
    n, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    assert n == 0 and not hb.any() and np.array_equal(cls, base)
