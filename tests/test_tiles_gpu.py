"""GPU parity of the tiled wide-matrix fill kernel (dp_tiles.hip): full matrices and wide bands spanning
several tiles, with the graph shapes each of its code paths is for -- plain sequences (simple steps), skip
edges inside the halo (straight-line two-edge cells), sites with three and more bwd edges, edges that start
before the halo or several tiles back (operands from HBM), predecessor-less sites -- plus the switch back to
the one-workgroup wavefront, a protein-sized table (not cached in LDS) and negative-zero parameters (which the
comparing wavefront kernel takes)."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def tiled_route(monkeypatch):
    """wide jobs whose model table fits LDS run as row strips on the banded kernel by default (tests/test_strips_gpu.py):
    these tests are about the tiled kernel"""
    monkeypatch.setenv("PAGAN_DP_WIDE", "tiles")


def same(a, b, what=""):
    assert a.status == b.status, what
    assert np.float64(a.score).tobytes() == np.float64(b.score).tobytes(), what + " score %r != %r" % (a.score, b.score)
    assert a.end == b.end, what
    assert np.array_equal(a.cols, b.cols), what + " columns differ"
    assert np.array_equal(a.left_used, b.left_used) and np.array_equal(a.right_used, b.right_used), what


def wide_band(Lx, Ly, half, seed):
    """A monotone band a few hundred cells wide: too wide for the banded kernel, so it is tiled."""
    rng = np.random.default_rng(seed)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    h = rng.integers(half // 2, half, Lx)
    upper = np.maximum.accumulate(np.maximum(centre - h, 0))
    lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    return abi.Band(upper, lower)


CASES = {
    # name: (left sites, right sites, p_extra, max_deg, max_span, p_dead, states)
    "plain": (300, 330, 0.0, 2, 2, 0.0, 15),
    "skip_edges_in_halo": (280, 260, 0.10, 2, 6, 0.0, 15),
    "three_and_more_edges": (260, 300, 0.15, 5, 7, 0.0, 15),
    "edges_from_before_the_halo": (270, 250, 0.08, 3, 40, 0.0, 15),
    "edges_across_tiles": (330, 300, 0.05, 4, 200, 0.0, 15),
    "dead_sites": (250, 270, 0.10, 3, 12, 0.03, 15),
    "protein_table": (200, 230, 0.08, 3, 10, 0.0, 211),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_matrix_over_several_tiles(pg, oracle, name):
    nl, nr, p_extra, max_deg, max_span, p_dead, states = CASES[name]
    left = synth.random_graph(nl, states, 101, p_extra=p_extra, max_deg=max_deg, max_span=max_span, p_dead=p_dead)
    right = synth.random_graph(nr, states, 202, p_extra=p_extra, max_deg=max_deg, max_span=max_span, p_dead=p_dead)
    model = synth.random_model(states, 7)
    side, tiles = pg.debug_tiles(left, right)
    assert len(tiles) >= 16, "the job is meant to span several tiles in both directions"
    same(pg.align(left, right, model), oracle.dp_align(left, right, model), name)


@pytest.mark.parametrize("seed", range(3))
def test_wide_band_is_tiled(pg, oracle, seed):
    left = synth.random_graph(900, 15, 10 + seed, p_extra=0.08, max_deg=4, max_span=30)
    right = synth.random_graph(860, 15, 20 + seed, p_extra=0.08, max_deg=4, max_span=30)
    band = wide_band(left.n_sites - 1, right.n_sites - 1, 400, seed)
    side, tiles = pg.debug_tiles(left, right, band)
    full = ((left.n_sites + side - 2) // side) * ((right.n_sites + side - 2) // side)
    assert 0 < len(tiles) < full, "tiles outside the band are not launched"
    model = synth.random_model(15, seed)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


@pytest.mark.parametrize("flags", [abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN])
def test_option_bits(pg, oracle, flags):
    left = synth.random_graph(200, 15, 5, p_extra=0.1, max_deg=3, max_span=9)
    right = synth.random_graph(170, 15, 6, p_extra=0.1, max_deg=3, max_span=9)
    model = synth.random_model(15, 3)
    same(pg.align(left, right, model, flags=flags), oracle.dp_align(left, right, model, flags=flags))


def test_tiles_and_wavefront_store_identical_scores(pg, monkeypatch):
    left = synth.random_graph(260, 15, 31, p_extra=0.12, max_deg=4, max_span=50)
    right = synth.random_graph(240, 15, 32, p_extra=0.12, max_deg=4, max_span=50)
    job = (left, right, synth.random_model(15, 9), None)
    scores = {}
    for kernel in ("tiles", "wavefront"):
        monkeypatch.setenv("PAGAN_DP_WIDE", kernel)
        b = pg.Batch([job])
        b.run(); b.sync()
        scores[kernel] = b.debug_scores(0)
        b.close()
    assert np.array_equal(scores["tiles"].view(np.int64), scores["wavefront"].view(np.int64))


def test_negative_zero_parameters_keep_the_comparing_kernel(pg, oracle):
    left = synth.random_graph(150, 15, 41, p_extra=0.1, max_deg=3, max_span=9)
    right = synth.random_graph(140, 15, 42, p_extra=0.1, max_deg=3, max_span=9)
    model = synth.random_model(15, 4)
    t = model.log_score.copy()
    t[2, 5] = t[5, 2] = np.float32(-0.0)
    m2 = abi.Model(t, *model.params)
    same(pg.align(left, right, m2), oracle.dp_align(left, right, m2))


def test_banded_and_tiled_jobs_share_a_batch(pg, oracle):
    """One launch sequence: the banded kernel on the batch's stream, the tile launches beside it."""
    a = synth.random_graph(500, 15, 51, p_extra=0.05, max_deg=3, max_span=8)
    b = synth.random_graph(520, 15, 52, p_extra=0.05, max_deg=3, max_span=8)
    Lx, Ly = a.n_sites - 1, b.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    upper = np.maximum.accumulate(np.maximum(centre - 20, 0)); lower = np.maximum.accumulate(np.minimum(centre + 20, Ly - 1))
    upper[0] = 0; lower[-1] = Ly - 1
    narrow = abi.Band(upper, lower)
    model = synth.random_model(15, 5)
    jobs = [(a, b, model, narrow), (a, b, model, None), (b, a, model, None)]
    got = pg.align_batch(jobs)
    for k, (l, r, m, band) in enumerate(jobs):
        same(got[k], oracle.dp_align(l, r, m, band), "job %d" % k)


@pytest.mark.parametrize("nl,nr", [(1, 700), (700, 1), (3, 500), (62, 62), (63, 64), (64, 65), (65, 127), (128, 129), (2, 2)])
def test_shapes_around_the_tile_size(pg, oracle, nl, nr):
    """Matrices of 1 x many tiles, and sides just below / at / above multiples of the tile side (the matrix has
    nl + 1 by nr + 1 cells: start site included)."""
    left = synth.random_graph(nl, 15, 300 + nl, p_extra=0.2, max_deg=4, max_span=12, p_dead=0.02)
    right = synth.random_graph(nr, 15, 400 + nr, p_extra=0.2, max_deg=4, max_span=12, p_dead=0.02)
    model = synth.random_model(15, nl + nr)
    same(pg.align(left, right, model), oracle.dp_align(left, right, model), "%d x %d" % (nl, nr))


def test_band_with_boxes_on_the_matrix_edges(pg, oracle):
    left = synth.random_graph(600, 15, 71, p_extra=0.1, max_deg=3, max_span=20)
    right = synth.random_graph(640, 15, 72, p_extra=0.1, max_deg=3, max_span=20)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    upper = np.maximum(centre - 150, 0); lower = np.minimum(centre + 150, Ly - 1)
    upper[:200] = 0; lower[:200] = np.maximum(lower[:200], 400)          # a box in the start corner
    upper[-150:] = upper[-150]; lower[-150:] = Ly - 1                      # and one in the end corner
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    band = abi.Band(upper, lower)
    side, tiles = pg.debug_tiles(left, right, band)
    assert len(tiles) > 30
    model = synth.random_model(15, 11)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band))


@pytest.mark.parametrize("schedule", ["flow", "nolag", "watermark", "launches"])
def test_tile_schedules_agree(pg, oracle, monkeypatch, schedule):
    """The dataflow launch (neighbour flags only / plus the diagonal watermark) and the launch per tile anti-diagonal
    compute the same alignment -- on a matrix deep enough that many tiles are in flight at once, with bwd edges that reach
    several tiles back (operands another wave wrote during the same launch)."""
    if schedule != "flow":
        monkeypatch.setenv("PAGAN_DP_TILES", schedule)
    left = synth.random_graph(1500, 15, 91, p_extra=0.08, max_deg=4, max_span=300, p_dead=0.01)
    right = synth.random_graph(1400, 15, 92, p_extra=0.08, max_deg=4, max_span=300, p_dead=0.01)
    model = synth.random_model(15, 13)
    band = wide_band(left.n_sites - 1, right.n_sites - 1, 700, 5)
    jobs = [(left, right, model, None), (left, right, model, band), (right, left, model, None)]
    got = pg.align_batch(jobs)
    for k, (l, r, m, bd) in enumerate(jobs):
        same(got[k], oracle.dp_align(l, r, m, bd), "%s job %d" % (schedule, k))


def test_band_whose_tiles_are_no_staircase(pg, oracle):
    """Rows whose column ranges do not touch (the band is monotone, so it is accepted; cells behind the jump are reachable
    through skip edges only): the tile rows do not touch either, the neighbour flags do not order a tile behind everything
    it may read, and the host falls back to the diagonal watermark."""
    left = synth.random_graph(700, 15, 95, p_extra=0.15, max_deg=4, max_span=400)
    right = synth.random_graph(900, 15, 96, p_extra=0.15, max_deg=4, max_span=400)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    upper = np.zeros(Lx, np.int64); lower = np.zeros(Lx, np.int64)
    upper[:320] = 0; lower[:320] = 330
    upper[320:] = 520; lower[320:] = Ly - 1                                # a jump of three tile columns, on a tile-row boundary
    assert not pg.debug_tiles_staircase(pg.debug_tiles(left, right, abi.Band(upper, lower))[1])
    band = abi.Band(upper, lower)
    side, tiles = pg.debug_tiles(left, right, band)
    assert len(tiles) > 30
    model = synth.random_model(15, 17)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band))


@pytest.mark.parametrize("compact", ["on", "off"])
@pytest.mark.parametrize("p_dead", [0.06, 0.3, 0.6])
def test_many_dead_sites(pg, oracle, monkeypatch, compact, p_dead):
    """Sites without bwd edges make whole rows / columns -inf; from 5 % on the library aligns the compacted graphs and maps
    the path back (dp_abi.hip: CompactJob).  Full matrix, wide band and a narrow band (the banded kernel), edges that start
    at dead sites included; with PAGAN_DP_COMPACT=0 the same inputs go through uncompacted."""
    if compact == "off":
        monkeypatch.setenv("PAGAN_DP_COMPACT", "0")
    left = synth.random_graph(900, 15, 501, p_extra=0.15, max_deg=4, max_span=40, p_dead=p_dead)
    right = synth.random_graph(840, 15, 502, p_extra=0.15, max_deg=4, max_span=40, p_dead=p_dead)
    model = synth.random_model(15, 21)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    narrow_u = np.maximum.accumulate(np.maximum(centre - 60, 0)); narrow_l = np.maximum.accumulate(np.minimum(centre + 60, Ly - 1))
    narrow_u[0] = 0; narrow_l[-1] = Ly - 1
    jobs = [(left, right, model, None), (left, right, model, wide_band(Lx, Ly, 400, 3)),
            (left, right, model, abi.Band(narrow_u, narrow_l)), (right, left, model, None)]
    got = pg.align_batch(jobs)
    for k, (l, r, m, bd) in enumerate(jobs):
        want = oracle.dp_align(l, r, m, bd)
        same(got[k], want, "p_dead %.2f compact %s job %d" % (p_dead, compact, k))
        assert got[k].cells == want.cells


@pytest.mark.parametrize("seed", range(6))
def test_dead_sites_at_the_matrix_edges(pg, oracle, seed):
    """The terminal-gap rules name the first and the last row / column: with the last real site of a sequence dead (and
    the first ones), the compacted matrices must still apply them to the same sites."""
    rng = np.random.default_rng(900 + seed)
    left = synth.random_graph(500, 15, 600 + seed, p_extra=0.15, max_deg=4, max_span=40, p_dead=0.3)
    right = synth.random_graph(460, 15, 700 + seed, p_extra=0.15, max_deg=4, max_span=40, p_dead=0.3)

    def kill(g, sites):
        """the given sites lose their bwd edges"""
        off = g.bwd_off.astype(np.int64)
        keep = np.ones(int(off[-1]), bool)
        for s in sites:
            keep[off[s]:off[s + 1]] = False
        cnt = np.diff(off)
        cnt[list(sites)] = 0
        new_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        return abi.Graph(g.state, new_off, g.bwd_src[keep], g.bwd_logw[keep], g.bwd_eid[keep], n_edges=g.n_edges)

    n_l, n_r = left.n_sites, right.n_sites
    left2 = kill(left, [n_l - 2] + ([1, 2] if seed % 2 else []))
    right2 = kill(right, [n_r - 2] if seed % 3 else [1])
    model = synth.random_model(15, 31 + seed)
    for flags in (0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN):
        same(pg.align(left2, right2, model, flags=flags), oracle.dp_align(left2, right2, model, flags=flags), "seed %d flags %d" % (seed, flags))


def test_a_codon_sized_table_on_the_tiled_kernel(pg, oracle):
    """1892 states (the codon alphabet's size, model_factory.cpp:839-897): a tile stages the 64 x 64 scores of its state pairs
    out of a 14 MB table."""
    S = 1892
    left, right = synth.random_graph(300, S, 5, p_extra=0.2, max_span=12), synth.random_graph(280, S, 6, p_extra=0.2, max_span=12)
    model = synth.random_model(S, 3)
    assert pg.debug_route(left, right, model, None)[0] == "pg_fill_tiles_flow"
    same(pg.align(left, right, model), oracle.dp_align(left, right, model), "codon-sized table")
