"""Host-side planner of the banded fill kernel (dp_abi.hip: classify_diagonals, schedule_waves) against a
brute-force restatement of its rules.  No GPU: pagan_dp_debug_plan is host code."""
import numpy as np
import pytest

import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth



def _geometry():
    """PG_PIPE_* of pagan2-msa_amd/csrc/dp_device.h (the kernel's geometry the planner has to agree with)"""
    import os, re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pagan2-msa_amd", "csrc", "dp_device.h")).read()
    return tuple(int(re.search(r"#define\s+PG_PIPE_%s\s+(\d+)" % k, text).group(1)) for k in ("REACH", "WIDTH", "WINDOW", "RING", "WAKE"))


REACH, WIDTH, WINDOW, RING, WAKE = _geometry()
assert WIDTH == 256 - REACH and RING >= REACH


def site_features(g, n):
    span = np.zeros(n, np.int64); simple = np.zeros(n, bool); nopred = np.zeros(n, bool); easy = np.zeros(n, bool)
    for s in range(n):
        a, b = g.bwd_off[s], g.bwd_off[s + 1]
        if b > a:
            span[s] = s - g.bwd_src[a:b].min()
        nopred[s] = b == a
        simple[s] = s > 0 and b - a == 1 and g.bwd_src[a] == s - 1 and g.bwd_logw[a] == 0.0
        # what the compute waves evaluate themselves: one edge from the previous site, alone or beside ONE other edge
        easy[s] = s > 0 and b - a in (1, 2) and int((g.bwd_src[a:b] == s - 1).sum()) == 1
    return span, simple, nopred, easy


def brute_plan(left, right, band):
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    lo = np.zeros(Lx, np.int64); hi = np.full(Lx, Ly - 1, np.int64)
    if band is not None:
        lo = np.maximum(band.upper[:Lx].astype(np.int64), 0); hi = np.minimum(band.lower[:Lx].astype(np.int64), Ly - 1)
    nd = Lx + Ly - 1
    sl, simL, npL, easyL = site_features(left, Lx)
    sr, simR, npR, easyR = site_features(right, Ly)
    distL = [[s - p for p in left.bwd_src[left.bwd_off[s]:left.bwd_off[s + 1]]] or [1] for s in range(Lx)]
    distR = [[s - p for p in right.bwd_src[right.bwd_off[s]:right.bwd_off[s + 1]]] or [1] for s in range(Ly)]
    cls = np.zeros(nd, np.uint8)
    need = np.zeros(nd, np.int64)
    rows_of = [[] for _ in range(nd)]
    for i in range(Lx):
        for j in range(lo[i], hi[i] + 1):
            rows_of[i + j].append(i)
    last_wide = -1000
    for d in range(nd):
        rows = rows_of[d]
        imin, imax = (rows[0], rows[-1]) if rows else (0, -1)
        cols = [d - i for i in rows]
        if len(rows) > WIDTH:
            c = 5 if len(rows) > WINDOW else 4; last_wide = d
        elif d - last_wide < REACH:
            c = 3
        elif not (imin >= 2 and imax <= Lx - 2 and d - imax >= 2 and d - imin <= Ly - 2):
            c = 3
        elif any(max(sl[i], 1) + max(sr[d - i], 1) >= REACH for i in rows):      # a site without bwd edges counts as span 1
            c = 2
        elif (imin < REACH or d - imax < REACH) and (any(not simL[i] for i in rows) or any(not simR[j] for j in cols)):
            c = 2
        elif any(not simL[i] for i in rows) or any(not simR[j] for j in cols):
            # class 1 (the compute waves' own) only if every multi-edge site on the diagonal is an easy one; the plan is
            # that of a job whose model table fits LDS
            c = 1 if all(easyL[i] for i in rows) and all(easyR[j] for j in cols) else 2
        else:
            c = 0
        cls[d] = c
        # how far back the cells of d read in the rows of the wave above (what bounds the reuse of a ring row)
        if c <= 2:
            ages = [2]
            for i in rows:
                for dl in distL[i]:
                    if dl > i % 64 and dl < REACH:        # this edge leaves the lane's block of 64 rows
                        ages.append(dl)
                        ages += [dl + dr for dr in distR[d - i] if dl + dr < REACH]
            need[d] = max(ages)
        else:
            need[d] = REACH - 1
    active = np.zeros((4, nd), bool)
    for d in range(nd):
        for w in range(4):
            active[w, d] = cls[d] >= 4 or any((i % 256) // 64 == w for i in rows_of[d])
    lead = np.full(nd, -1, np.int64)
    for D in range(RING, nd):
        for t in range(D - RING + 1, min(D - RING + REACH, nd)):
            if t - need[t] <= D - RING:
                lead[D] = t
    return cls, active, lead


def check(left, right, band):
    cls, waves, lead = pg.debug_plan(left, right, band, with_lead=True)
    want, active, lead_exact = brute_plan(left, right, band)
    assert np.array_equal(cls, want), "classes differ at %s" % np.nonzero(cls != want)[0][:10]
    nd = cls.size
    # the downstream wave must have completed at least what the exact rule asks (the planner bounds a multi-edge
    # diagonal's reach from above), never the diagonal being computed, and in simple stretches
    # exactly RING - 2 diagonals back
    assert (lead >= lead_exact).all() and (lead <= np.maximum(np.arange(nd) - 1, -1)).all()
    simple_run = np.convolve((cls == 0).astype(int), np.ones(RING, int), 'full')[:nd] == RING
    d_idx = np.nonzero(simple_run)[0]
    assert (lead[d_idx] == d_idx - RING + 2).all()
    for w in range(4):
        awake = np.zeros(nd, bool)
        last = -1
        for a, b in waves[w]:
            assert last < a < b <= nd
            awake[a:b] = True
            last = b
        # awake from WAKE steps before a row of the wave is in the band until RING steps after the last one left
        need = np.zeros(nd, bool)
        for d in np.nonzero(active[w])[0]:
            need[max(0, d - WAKE): min(nd, d + RING + 1)] = True
        assert np.array_equal(awake, need), "wave %d schedule differs" % w


@pytest.mark.parametrize("seed", range(4))
def test_plan_random_graphs_full_matrix(seed):
    left = synth.random_graph(60 + 40 * seed, 15, seed, p_extra=0.3, p_dead=0.02 * (seed % 2))
    right = synth.random_graph(90 + 25 * seed, 15, 50 + seed, p_extra=0.3, p_dead=0.0)
    check(left, right, None)


@pytest.mark.parametrize("seed,p_dead", [(0, 0.05), (1, 0.3)])
def test_plan_banded_with_sites_without_bwd_edges(seed, p_dead):
    rng = np.random.default_rng(seed)
    left = synth.random_graph(400, 15, 70 + seed, p_extra=0.08, p_dead=p_dead, max_span=25)
    right = synth.random_graph(420, 15, 80 + seed, p_extra=0.08, p_dead=p_dead, max_span=25)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(8, 40, Lx)
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0)); lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    upper[0] = 0; lower[-1] = Ly - 1
    check(left, right, abi.Band(upper, lower))


@pytest.mark.parametrize("seed", range(4))
def test_plan_banded_with_long_edges(seed):
    rng = np.random.default_rng(seed)
    n = 700
    span = 40 if seed < 2 else 8
    left = synth.random_graph(n, 15, 10 + seed, p_extra=0.08, p_dead=0.0, max_span=span)
    right = synth.random_graph(n + 30, 15, 20 + seed, p_extra=0.08, p_dead=0.0, max_span=span)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(8, 40, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    # one box wider than the lanes in both directions, to get class 4 and the class-3 steps after it
    # seed 1: wider than 352 cells (the wide ring of 9 rows x 512 positions); seed 3: wider than the record windows too (class 5)
    at, rows, jump = {1: (300, 390, 420), 3: (200, 480, 60)}.get(seed, (300, 260, 300))
    upper[at:at + rows] = upper[at]
    lower[at:at + rows] = np.minimum(lower[at - 1 + rows] + jump, Ly - 1)
    lower = np.maximum.accumulate(lower); upper = np.maximum.accumulate(upper)
    check(left, right, abi.Band(upper, lower))
    cls, _ = pg.debug_plan(left, right, abi.Band(upper, lower))
    assert set(np.unique(cls)) >= ({2, 3, 4} if seed < 2 else {1, 3, 4})
    assert (5 in cls) == (seed == 3)
    # the far plan (histories, third pass) the batch path uses has the same general and wide steps
    cls_far = pg.debug_far(left, right, abi.Band(upper, lower))[4] & 15
    assert np.array_equal(cls_far == 3, cls == 3) and np.array_equal(cls_far >= 4, cls >= 4)


@pytest.mark.parametrize("p_extra", [0.0, 0.3])
def test_two_general_steps_behind_a_wide_run(p_extra):
    """Behind a wide run the loop restarts after TWO general steps (the lane's cell and its shifted predecessor are back in
    registers); until the ring holds REACH - 1 diagonals again a diagonal of simple sites is class 0 and one with any other site class 2
    (its residency mask sends the older operands to L2), never class 1.  A box in the middle of the matrix, away from its borders."""
    n = 1300
    left = synth.random_graph(n, 15, 31, p_extra=p_extra, p_dead=0.0, max_span=6)
    right = synth.random_graph(n, 15, 32, p_extra=p_extra, p_dead=0.0, max_span=6)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum(centre - 20, 0); lower = np.minimum(centre + 20, Ly - 1)
    upper[400:700] = upper[400]; lower[400:700] = lower[699] + 40
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    upper[0] = 0; lower[-1] = Ly - 1
    band = abi.Band(upper, lower)
    for cls in (pg.debug_plan(left, right, band)[0], pg.debug_far(left, right, band)[4] & 15):
        wide = np.nonzero(cls >= 4)[0]
        assert len(wide) > 100
        last = int(wide[-1])
        assert cls[last + 1] == 3 and cls[last + 2] == 3
        near = cls[last + 3: last + REACH]
        assert np.all((near == 0) | (near == 2)) and np.any(near == 2), near       # (these graphs' edges carry weights: no site is "simple")
        if p_extra == 0.0: assert np.any(cls[last + REACH: last + REACH + 40] == 1)      # ... and class 1 again once the ring is full


def test_tile_list_covers_exactly_the_tiles_the_band_touches():
    """dp_tiles.hip is launched over the tiles listed by the host: every in-band cell must lie in a listed tile,
    every listed tile's row block must reach its column block, and a tile row lists its columns contiguously."""
    import pagan2_msa_amd as pg
    rng = np.random.default_rng(5)
    left = synth.random_graph(700, 15, 1, p_extra=0.05)
    right = synth.random_graph(640, 15, 2, p_extra=0.05)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    for case in range(4):
        if case == 0:
            band, lo, hi = None, np.zeros(Lx, np.int64), np.full(Lx, Ly - 1)
        else:
            centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
            h = rng.integers(10, 60 * case, Lx)
            lo = np.maximum.accumulate(np.maximum(centre - h, 0)); hi = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
            lo[0] = 0; hi[-1] = Ly - 1
            band = abi.Band(lo, hi)
        side, tiles = pg.debug_tiles(left, right, band)
        assert side == 64
        want = set()
        for i in range(Lx):
            for b in range(lo[i] // side, hi[i] // side + 1):
                want.add((i // side, int(b)))
        assert want <= set(tiles)
        rows = {}
        for a, b in tiles:
            rows.setdefault(a, []).append(b)
        for a, bs in rows.items():
            assert bs == list(range(bs[0], bs[-1] + 1))
            assert bs[0] == lo[a * side: (a + 1) * side].min() // side and bs[-1] == hi[a * side: (a + 1) * side].max() // side
        assert len(tiles) == len(set(tiles))


def test_a_tile_with_too_many_edges_is_not_tiled():
    import pagan2_msa_amd as pg
    left = synth.random_graph(200, 15, 3, p_extra=1.0, max_deg=40, max_span=60)      # ~20 extra edges per site
    right = synth.random_graph(200, 15, 4, p_extra=0.0)
    side, tiles = pg.debug_tiles(left, right)
    assert tiles == []


@pytest.mark.parametrize("seed", range(4))
def test_plan_over_several_threads_is_the_plan(monkeypatch, seed):
    """round 5: the plan of one alignment runs over ranges of diagonals on several host threads (dp_abi.hip, par_ranges; the
    sliding windows restart at a range's first diagonal).  Classes, wave schedules and ring-row reuse are those of one thread.
    Long jobs only: a range has at least 4096 diagonals."""
    rng = np.random.default_rng(700 + seed)
    n = int(rng.integers(9000, 14000))
    left = synth.random_graph(n, 15, 7000 + seed, p_extra=0.06, max_deg=4, max_span=int(rng.choice([6, 20, 40])), p_dead=0.002)
    right = synth.random_graph(n + int(rng.integers(-300, 300)), 15, 7100 + seed, p_extra=0.06, max_deg=4, max_span=int(rng.choice([6, 20, 40])))
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(5, 90, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0)); lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    for _ in range(3):                                             # boxes wider than the lanes / the record windows
        a = int(rng.integers(10, Lx - 500)); b = a + int(rng.integers(40, 420))
        upper[a:b] = upper[a]; lower[a:b] = min(lower[b - 1] + int(rng.integers(100, 450)), Ly - 1)
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower); upper[0] = 0; lower[-1] = Ly - 1
    band = abi.Band(upper, lower)
    monkeypatch.setenv("PAGAN_DP_PLAN_THREADS", "1")
    one = pg.debug_plan(left, right, band, with_lead=True)
    for t in ("3", "8"):
        monkeypatch.setenv("PAGAN_DP_PLAN_THREADS", t)
        many = pg.debug_plan(left, right, band, with_lead=True)
        assert np.array_equal(one[0], many[0])                     # classes
        assert len(one[1]) == len(many[1]) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(one[1], many[1]))   # wave schedules
        assert np.array_equal(one[2], many[2])                     # ring-row reuse
