"""GPU parity of the row strips (dp_pipe.hip, strip_feeder): a wide job -- a full matrix, a band wider than the banded kernel's
lanes -- cut into strips of 192 rows that the banded kernel fills one behind the other, each strip's fourth wave feeding the
rows above it into the ring.  Same graph shapes as the tiled kernel's tests (plain sequences, skip edges, three and more bwd
edges, edges reaching past the ring and into other strips, sites without predecessors), the option bits, a wide band, several
jobs in one batch, and matrices of one, two and many strips."""
import numpy as np
import pytest

import pagan2_msa_amd as pgm
from pagan2_msa_amd import abi, synth

pytestmark = pytest.mark.gpu


def same(a, b, what=""):
    assert a.status == b.status, what
    assert np.float64(a.score).tobytes() == np.float64(b.score).tobytes(), what + " score %r != %r" % (a.score, b.score)
    assert a.end == b.end, what
    assert np.array_equal(a.cols, b.cols), what + " columns differ"
    assert np.array_equal(a.left_used, b.left_used) and np.array_equal(a.right_used, b.right_used), what


@pytest.fixture(params=["any_xcd", "one_xcd"])
def strips(monkeypatch, request):
    """any_xcd (the default): a job's strips wherever the dispatcher puts them, scores written through to memory;
    one_xcd (PAGAN_DP_STRIP_SPREAD=0, round 4's placement): a job's strips at workgroup indices of one residue mod 8"""
    monkeypatch.setenv("PAGAN_DP_STRIP_SPREAD", "1" if request.param == "any_xcd" else "0")
    monkeypatch.setenv("PAGAN_DP_WIDE", "strips")
    monkeypatch.setenv("PAGAN_DP_STRIP_SITES", "100000")      # (every wide job here runs as strips, however many multi-edge sites its diagonals hold)
    monkeypatch.setenv("PAGAN_DP_STRIP_STATES", "100000")     # (... and however large its model table: by default only tables that fit LDS do)


CASES = {
    # name: (left sites, right sites, p_extra, max_deg, max_span, p_dead)
    "plain": (300, 330, 0.0, 2, 2, 0.0),
    "two_strips_wide": (250, 700, 0.05, 3, 6, 0.0),
    "skip_edges_in_the_ring": (480, 260, 0.10, 2, 6, 0.0),
    "three_and_more_edges": (460, 300, 0.15, 5, 7, 0.0),
    "edges_past_the_ring": (470, 250, 0.08, 3, 40, 0.0),
    "edges_across_strips": (730, 300, 0.05, 4, 300, 0.0),
    "dead_sites": (450, 270, 0.10, 3, 12, 0.03),
    "five_strips": (900, 500, 0.06, 3, 10, 0.0),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_matrix_as_strips(pg, oracle, strips, name):
    nl, nr, p_extra, max_deg, max_span, p_dead = CASES[name]
    left = synth.random_graph(nl, 15, 101, p_extra=p_extra, max_deg=max_deg, max_span=max_span, p_dead=p_dead)
    right = synth.random_graph(nr, 15, 202, p_extra=p_extra, max_deg=max_deg, max_span=max_span, p_dead=p_dead)
    model = synth.random_model(15, 7)
    assert pg.debug_route(left, right, model)[0] == "pg_fill_pipe (row strips)"
    same(pg.align(left, right, model), oracle.dp_align(left, right, model), name)


PROTEIN_CASES = {
    # a 211-state table (not cached in LDS: the compute waves' C++ step, every model score gathered by the assist waves)
    "plain_weighted": (400, 380, 0.0, 2, 2),
    "multi_edge": (460, 430, 0.12, 4, 20),
    "one_row_in_the_last_strip": (385, 300, 0.08, 3, 9),
    "far_edges": (600, 250, 0.06, 4, 200),
}


@pytest.mark.parametrize("name", sorted(PROTEIN_CASES))
@pytest.mark.parametrize("flags", [0, abi.OPT_NO_TERMINAL_EDGES])
def test_protein_table_as_strips(pg, oracle, strips, name, flags):
    nl, nr, p_extra, max_deg, max_span = PROTEIN_CASES[name]
    left = synth.random_graph(nl, 211, 11, p_extra=p_extra, max_deg=max_deg, max_span=max_span)
    right = synth.random_graph(nr, 211, 12, p_extra=p_extra, max_deg=max_deg, max_span=max_span)
    model = synth.random_model(211, 3)
    assert pg.debug_route(left, right, model)[0] == "pg_fill_pipe (row strips)"
    same(pg.align(left, right, model, flags=flags), oracle.dp_align(left, right, model, flags=flags), name)


@pytest.mark.parametrize("flags", [abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN])
def test_option_bits(pg, oracle, strips, flags):
    left = synth.random_graph(400, 15, 5, p_extra=0.1, max_deg=3, max_span=9)
    right = synth.random_graph(370, 15, 6, p_extra=0.1, max_deg=3, max_span=9)
    model = synth.random_model(15, 3)
    same(pg.align(left, right, model, flags=flags), oracle.dp_align(left, right, model, flags=flags))


def wide_band(Lx, Ly, half, seed):
    rng = np.random.default_rng(seed)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    h = rng.integers(half // 2, half, Lx)
    upper = np.maximum.accumulate(np.maximum(centre - h, 0))
    lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    return abi.Band(upper, lower)


@pytest.mark.parametrize("seed", range(3))
def test_wide_band_as_strips(pg, oracle, strips, seed):
    left = synth.random_graph(900, 15, 10 + seed, p_extra=0.08, max_deg=4, max_span=30)
    right = synth.random_graph(860, 15, 20 + seed, p_extra=0.08, max_deg=4, max_span=30)
    band = wide_band(left.n_sites - 1, right.n_sites - 1, 400, seed)
    model = synth.random_model(15, seed)
    assert pg.debug_route(left, right, model, band)[0] == "pg_fill_pipe (row strips)"
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


def test_batch_of_striped_jobs(pg, oracle, strips):
    """more jobs than XCDs: two jobs share an XCD's workgroup indices"""
    jobs = []
    for k in range(11):
        left = synth.random_graph(250 + 37 * k, 15, 300 + k, p_extra=0.07, max_deg=3, max_span=12)
        right = synth.random_graph(280 + 23 * k, 15, 400 + k, p_extra=0.07, max_deg=3, max_span=12)
        jobs.append((left, right, synth.random_model(15, k), None))
    got = pg.align_batch(jobs)
    for k, (left, right, model, band) in enumerate(jobs):
        same(got[k], oracle.dp_align(left, right, model), "job %d" % k)


def test_a_strip_on_another_xcd_runs_the_batch_again_alone(pg, oracle, strips, monkeypatch):
    """debug flag 0x800: every feeder reports the strip above on another XCD; the fetch launches the strips once more with
    nothing dispatched beside them (the host clears the bit for that launch) and returns that result"""
    monkeypatch.setenv("PAGAN_DP_DEBUG_FLAGS", "0x800")
    monkeypatch.setenv("PAGAN_DP_STRIP_SPREAD", "0")          # (only this placement asks where the strip above runs)
    left = synth.random_graph(500, 15, 71, p_extra=0.08, max_deg=3, max_span=12)
    right = synth.random_graph(450, 15, 72, p_extra=0.08, max_deg=3, max_span=12)
    model = synth.random_model(15, 9)
    b = pg.Batch([(left, right, model, None)])
    b.run()
    got = b.fetch()[0]
    assert b.debug_reruns() == 1
    b.close()
    same(got, oracle.dp_align(left, right, model))


def test_strips_that_fail_alone_too_go_to_the_tiled_kernel(pg, oracle, strips, monkeypatch):
    """debug flags 0x800 | 0x400: the launch of the strips alone reports another XCD as well.  A resident batch hands the
    error to its caller; pagan_dp_align_batch plans the batch a third time with every wide job on the tiled kernel and
    returns that result (round 4's advisor: the 'alone' launch is alone only within its own batch)."""
    monkeypatch.setenv("PAGAN_DP_DEBUG_FLAGS", "0xc00")
    monkeypatch.setenv("PAGAN_DP_STRIP_SPREAD", "0")
    left = synth.random_graph(500, 15, 73, p_extra=0.08, max_deg=3, max_span=12)
    right = synth.random_graph(450, 15, 74, p_extra=0.08, max_deg=3, max_span=12)
    model = synth.random_model(15, 10)
    b = pg.Batch([(left, right, model, None)])
    b.run()
    with pytest.raises(pgm.PaganError) as e:
        b.fetch()
    assert e.value.code == abi.PAGAN_E_INTERNAL and b.debug_reruns() == 1
    b.close()
    same(pg.align(left, right, model), oracle.dp_align(left, right, model))


@pytest.mark.parametrize("seed", range(6))
def test_random_batches_as_strips(pg, oracle, strips, seed):
    """random wide jobs (full matrices and wide bands, dead sites, long edges, several jobs per XCD) against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    jobs = []
    for k in range(int(rng.integers(2, 10))):
        nl, nr = int(rng.integers(200, 1300)), int(rng.integers(200, 1300))
        span = int(rng.choice([6, 30, 300]))
        left = synth.random_graph(nl, 15, int(rng.integers(1 << 30)), p_extra=float(rng.choice([0.03, 0.1, 0.25])), max_deg=int(rng.integers(2, 6)),
                                  max_span=span, p_dead=float(rng.choice([0.0, 0.02])))
        right = synth.random_graph(nr, 15, int(rng.integers(1 << 30)), p_extra=float(rng.choice([0.03, 0.1, 0.25])), max_deg=int(rng.integers(2, 6)),
                                   max_span=span, p_dead=float(rng.choice([0.0, 0.02])))
        band = wide_band(left.n_sites - 1, right.n_sites - 1, int(rng.integers(300, 600)), seed) if rng.random() < 0.5 else None
        jobs.append((left, right, synth.random_model(15, int(rng.integers(1 << 30))), band))
    flags = int(rng.choice([0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN]))
    got = pg.align_batch(jobs, flags=flags)
    for k, (left, right, model, band) in enumerate(jobs):
        same(got[k], oracle.dp_align(left, right, model, band, flags=flags), "seed %d job %d" % (seed, k))
