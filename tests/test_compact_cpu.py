"""Host logic of the dead-site compaction (dp_abi.hip: CompactSide, compact_band) through the host-only
diagnostic entry point: which sites go (no bwd edge, or bwd edges from such sites only -- the cascade), which always stay
(start, end, the last site before the end), which edges are dropped and what position the kept ones had in the caller's
lists, and how a band is re-indexed.  Checked against a brute-force restatement in Python."""
import numpy as np
import pytest

import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth


def brute(g):
    n = g.n_sites
    off = g.bwd_off.astype(np.int64)
    alive = np.ones(n, bool)
    for s in range(1, n - 2):
        src = g.bwd_src[off[s]:off[s + 1]]
        alive[s] = bool(np.any(alive[src])) if len(src) else False
    keep = np.nonzero(alive)[0]
    slots = []
    for s in keep:
        for k, e in enumerate(range(off[s], off[s + 1])):
            if alive[g.bwd_src[e]]:
                slots.append(k)
    return alive, keep, np.array(slots, np.int64)


@pytest.mark.parametrize("seed", range(8))
def test_compaction_of_random_graphs(seed):
    p_dead = [0.0, 0.05, 0.2, 0.5][seed % 4]
    left = synth.random_graph(300 + 17 * seed, 15, 1000 + seed, p_extra=0.2, max_deg=4, max_span=30, p_dead=p_dead)
    right = synth.random_graph(260 + 13 * seed, 15, 2000 + seed, p_extra=0.2, max_deg=4, max_span=30, p_dead=p_dead)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    rng = np.random.default_rng(seed)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    h = rng.integers(5, 60, Lx)
    upper = np.maximum.accumulate(np.maximum(centre - h, 0)); lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
    upper[0] = 0; lower[-1] = Ly - 1
    band = abi.Band(upper, lower)
    got = pg.debug_compact(left, right, band)
    al, kl, sl = brute(left)
    ar, kr, sr = brute(right)
    assert np.array_equal(got["keep_left"], kl) and np.array_equal(got["keep_right"], kr)
    assert np.array_equal(got["slot_left"], sl) and np.array_equal(got["slot_right"], sr)
    # the start site, the end site and the last site before it stay whatever they are
    for keep, g in ((kl, left), (kr, right)):
        assert keep[0] == 0 and keep[-1] == g.n_sites - 1 and keep[-2] == g.n_sites - 2
    # the band: per kept row the kept columns inside the caller's interval, in compacted numbers
    col_new = -np.ones(right.n_sites, np.int64)
    col_new[kr] = np.arange(len(kr))
    for t, i in enumerate(kl[:-1]):
        inside = [col_new[c] for c in range(int(upper[i]), int(lower[i]) + 1) if col_new[c] >= 0 and c < Ly]
        if inside:
            assert (got["upper"][t], got["lower"][t]) == (min(inside), max(inside)), (t, i)
        else:
            assert got["upper"][t] > got["lower"][t], (t, i)
    assert np.all(np.diff(got["upper"]) >= 0) and np.all(np.diff(got["lower"]) >= 0)


def test_cascade_and_edges_from_dead_sites():
    """0 <- 1 <- 2 ... a chain; site 3 loses its edges, site 4 hangs on 3 alone (dead by cascade), site 5 has one edge from 4
    (dead) and one from 2 (alive): it stays, with the second edge only, which was at position 1 of its list."""
    n = 9
    off = [0]; src = []; w = []; eid = []
    def site(edges):
        for s_ in edges:
            src.append(s_); w.append(-0.1); eid.append(len(eid))
        off.append(len(src))
    site([])            # 0: start
    site([0]); site([1])
    site([])            # 3: dead
    site([3])           # 4: dead by cascade
    site([4, 2])        # 5: keeps (2 -> 5), position 1
    site([5]); site([6])
    site([7])           # 8: end
    g = abi.Graph(np.r_[-1, np.zeros(n - 2, np.int32), -1], off, src, np.array(w, np.float32), eid)
    got = pg.debug_compact(g, g)
    assert got["keep_left"].tolist() == [0, 1, 2, 5, 6, 7, 8]
    assert got["slot_left"].tolist() == [0, 0, 1, 0, 0, 0]


def test_staircase_predicate():
    """When the tiled kernel's dataflow launch may order a tile behind its three neighbours only (dp_abi.hip: tiles_staircase)."""
    full = [(a, b) for a in range(5) for b in range(6)]
    assert pg.debug_tiles_staircase(full)
    band = [(a, b) for a in range(8) for b in range(max(0, a - 1), min(8, a + 3))]
    assert pg.debug_tiles_staircase(band)
    assert pg.debug_tiles_staircase([(3, 4)])                                   # one tile; rows before the first one do not count
    assert not pg.debug_tiles_staircase([(0, 0), (0, 1), (0, 3)])               # a hole in a row
    assert not pg.debug_tiles_staircase([(0, 0), (0, 1), (2, 1), (2, 2)])       # an empty row between two rows
    assert not pg.debug_tiles_staircase([(0, 0), (0, 1), (1, 3), (1, 4)])       # rows that do not touch
    assert pg.debug_tiles_staircase([(0, 0), (0, 1), (1, 2), (1, 3)])           # touching through the corner
    assert not pg.debug_tiles_staircase([(0, 2), (0, 3), (1, 1), (1, 2), (1, 3)])   # first column falls
    assert not pg.debug_tiles_staircase([(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)])   # last column falls


def test_real_bands_are_staircases():
    left = synth.random_graph(700, 15, 5, p_extra=0.1, max_deg=3, max_span=20)
    right = synth.random_graph(900, 15, 6, p_extra=0.1, max_deg=3, max_span=20)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    upper = np.maximum.accumulate(np.maximum(centre - 300, 0)); lower = np.maximum.accumulate(np.minimum(centre + 300, Ly - 1))
    upper[0] = 0; lower[-1] = Ly - 1
    side, tiles = pg.debug_tiles(left, right, abi.Band(upper, lower))
    assert len(tiles) > 20 and pg.debug_tiles_staircase(tiles)
    side, tiles = pg.debug_tiles(left, right)
    assert pg.debug_tiles_staircase(tiles)
    # a jump of three tile columns between two rows: inside a tile row the tile list bridges it (a tile row is listed from
    # its first row's first column to its last row's last column), on a tile-row boundary it does not -- the band of
    # tests/test_tiles_gpu.py::test_band_whose_tiles_are_no_staircase
    for jump_row, stair in ((350, True), (320, False)):
        upper = np.zeros(Lx, np.int64); lower = np.zeros(Lx, np.int64)
        upper[:jump_row] = 0; lower[:jump_row] = 330
        upper[jump_row:] = 520; lower[jump_row:] = Ly - 1
        side, tiles = pg.debug_tiles(left, right, abi.Band(upper, lower))
        assert len(tiles) > 20 and pg.debug_tiles_staircase(tiles) == stair, jump_row
