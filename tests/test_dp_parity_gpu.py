"""GPU parity: HIP fill + traceback (through the C ABI) against the CPU oracle, bit-exact on
score bits, end cell, alignment columns and used-edge sets."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, synth

pytestmark = pytest.mark.gpu


def assert_same(a, b, what=""):
    assert a.status == b.status, what
    assert np.float64(a.score).tobytes() == np.float64(b.score).tobytes(), "%s score %r vs %r" % (what, a.score, b.score)
    assert a.end == b.end, "%s end %r vs %r" % (what, a.end, b.end)
    assert np.array_equal(a.cols, b.cols), what + " columns differ"
    assert np.array_equal(a.left_used, b.left_used), what + " left used edges differ"
    assert np.array_equal(a.right_used, b.right_used), what + " right used edges differ"


def pair_of_leaves(length, seed, **kw):
    _, seqs, _ = synth.evolve_balanced(2, length, seed=seed, **kw)
    return synth.chain_graph(seqs[0]), synth.chain_graph(seqs[1]), seqs


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("length", [1, 7, 64, 300])
def test_plain_leaves_full_matrix(pg, oracle, seed, length):
    left, right, _ = pair_of_leaves(length, seed, sub=0.1, indel_start=0.03)
    model = synth.jc_like_dna_model(0.1)
    assert_same(pg.align(left, right, model), oracle.dp_align(left, right, model))


@pytest.mark.parametrize("flags", [0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN,
                                   abi.OPT_NO_TERMINAL_EDGES | abi.OPT_NO_REDUCED_TERMINAL_PEN])
def test_option_bits(pg, oracle, flags):
    left, right, _ = pair_of_leaves(200, 11, sub=0.08, indel_start=0.05, mean_len=6)
    model = synth.jc_like_dna_model(0.2)
    assert_same(pg.align(left, right, model, flags=flags), oracle.dp_align(left, right, model, flags=flags))


@pytest.mark.parametrize("seed", range(8))
def test_random_graphs(pg, oracle, seed):
    """Multi-edge sites in shuffled list order, long edges, weighted edges, dead sites, tie-rich table."""
    S = 15 if seed % 2 == 0 else 23
    left = synth.random_graph(40 + 17 * seed, S, seed, p_extra=0.4, p_dead=0.03 * (seed % 3))
    right = synth.random_graph(55 + 11 * seed, S, 100 + seed, p_extra=0.4, p_dead=0.02 * (seed % 2))
    model = synth.random_model(S, seed)
    got, want = pg.align(left, right, model), oracle.dp_align(left, right, model)
    assert_same(got, want, "seed %d" % seed)


def test_wide_diagonals_use_multiwave_block(pg, oracle):
    """Diagonals wider than one wave (256/1024-thread workgroups with barriers)."""
    left, right, _ = pair_of_leaves(700, 5)
    model = synth.jc_like_dna_model(0.1)
    assert_same(pg.align(left, right, model), oracle.dp_align(left, right, model))
    left = synth.random_graph(600, 15, 77, p_extra=0.2)
    right = synth.random_graph(650, 15, 78, p_extra=0.2)
    model = synth.random_model(15, 5)
    assert_same(pg.align(left, right, model), oracle.dp_align(left, right, model))


def test_banded_prefix_anchor_tunnel(pg, oracle):
    _, seqs, _ = synth.evolve_balanced(2, 3000, branch=0.01, sub=0.01, indel_start=0.002, seed=9)
    ol, orr = oracle.OGraph.leaf(seqs[0]), oracle.OGraph.leaf(seqs[1])
    band, nhits = oracle.define_tunnel(ol, orr)
    assert nhits > 5
    left, right = ol.flatten(), orr.flatten()
    model = synth.jc_like_dna_model(0.02)
    got, want = pg.align(left, right, model, band), oracle.dp_align(left, right, model, band)
    assert_same(got, want)
    assert got.cells < 0.3 * (left.n_sites - 1) * (right.n_sites - 1)
    # a full-width band is the full matrix
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    full = abi.Band(np.zeros(Lx, np.int32), np.full(Lx, Ly, np.int32))
    assert_same(pg.align(left, right, model, full), pg.align(left, right, model))


def test_random_monotone_bands_on_graphs(pg, oracle):
    rng = np.random.default_rng(4)
    for seed in range(4):
        left = synth.random_graph(150, 15, 200 + seed, p_extra=0.3)
        right = synth.random_graph(170, 15, 300 + seed, p_extra=0.3)
        Lx, Ly = left.n_sites - 1, right.n_sites - 1
        centre = np.linspace(0, Ly - 1, Lx)
        up = np.maximum.accumulate(np.clip(centre - rng.integers(3, 25, Lx), 0, None)).astype(np.int32)
        lo = np.maximum.accumulate(np.clip(centre + rng.integers(3, 25, Lx), 0, Ly + 5)).astype(np.int32)
        up[0] = 0
        band = abi.Band(up, lo)
        model = synth.random_model(15, seed)
        assert_same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


def test_homopolymer_and_454_leaves(pg, oracle):
    s1 = "ACGTTTTTTACGGGGACCCCCCCCATTTAGGA" * 6
    s2 = "ACGTTTTTACGGGGGACCCCCCATTAGGA" * 6
    model = synth.jc_like_dna_model(0.1)
    for flags in (1, 2):
        left, right = oracle.OGraph.leaf(s1, flags=flags).flatten(), oracle.OGraph.leaf(s2, flags=flags).flatten()
        assert int(np.diff(left.bwd_off).max()) > 1
        assert_same(pg.align(left, right, model), oracle.dp_align(left, right, model), "flags %d" % flags)


def test_unreachable_end_corner_is_a_status(pg, oracle):
    left, right, _ = pair_of_leaves(60, 3)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    up = np.zeros(Lx, np.int32)
    lo = np.full(Lx, 3, np.int32)           # tunnel never reaches the last columns
    band = abi.Band(up, lo)
    if Ly - 1 <= 3:
        pytest.skip("sequence too short")
    model = synth.jc_like_dna_model(0.1)
    got, want = pg.align(left, right, model, band), oracle.dp_align(left, right, model, band)
    assert got.status == abi.PAGAN_DP_UNREACHABLE == want.status
    assert got.score == -np.inf and got.cols.shape[0] == 0


def test_batch_matches_single_calls(pg, oracle):
    jobs = []
    for seed in range(5):
        left = synth.random_graph(30 + 40 * seed, 15, 500 + seed)
        right = synth.random_graph(90 - 10 * seed, 15, 600 + seed)
        jobs.append((left, right, synth.random_model(15, seed), None))
    res = pg.align_batch(jobs)
    for (l, r, m, b), got in zip(jobs, res):
        assert_same(got, oracle.dp_align(l, r, m, b))
    batch = pg.Batch(jobs)
    batch.run()
    batch.run()                                   # resident inputs can be replayed
    for (l, r, m, b), got in zip(jobs, batch.fetch()):
        assert_same(got, oracle.dp_align(l, r, m, b))
    assert batch.cells == sum(r.cells for r in res)
    ms = batch.last_ms()
    assert ms[0] > 0 and ms[1] > 0
    batch.close()


def test_invalid_inputs_are_error_codes(pg):
    left, right, _ = pair_of_leaves(20, 1)
    model = synth.jc_like_dna_model(0.1)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    bad = abi.Band(np.array([0] + [5] * (Lx - 2) + [2], np.int32), np.full(Lx, Ly, np.int32))
    with pytest.raises(pg.PaganError) as e:
        pg.align(left, right, model, bad)
    assert e.value.code == abi.PAGAN_E_BAND
    short = abi.Band(np.zeros(3, np.int32), np.zeros(3, np.int32))
    with pytest.raises(pg.PaganError) as e:
        pg.align(left, right, model, short)
    assert e.value.code == abi.PAGAN_E_BAND
    small = synth.random_model(2, 0)
    with pytest.raises(pg.PaganError) as e:
        pg.align(left, right, small)
    assert e.value.code == abi.PAGAN_E_MODEL
    st = left.state.copy()
    broken = abi.Graph(st, left.bwd_off, left.bwd_src + 1, left.bwd_logw, left.bwd_eid, n_edges=left.n_edges)
    with pytest.raises(pg.PaganError) as e:
        pg.align(broken, right, model)
    assert e.value.code == abi.PAGAN_E_GRAPH


def banded_around_diagonal(left, right, rng, lo_w=4, hi_w=30):
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.linspace(0, Ly - 1, Lx)
    up = np.maximum.accumulate(np.clip(centre - rng.integers(lo_w, hi_w, Lx), 0, None)).astype(np.int32)
    lo = np.maximum.accumulate(np.clip(centre + rng.integers(lo_w, hi_w, Lx), 0, Ly + 5)).astype(np.int32)
    up[0] = 0
    return abi.Band(up, lo)


def test_ring_kernel_long_edges_around_ring_depth(pg, oracle):
    """Banded jobs (LDS ring kernel) whose graph edges reach 1..40 diagonals back: predecessors in
    the ring, exactly at its depth, and beyond it (read back from HBM).  Output buffers are
    poisoned between runs so a read of a not-yet-written cell cannot pass by luck; repeated
    runs catch wave-timing races."""
    rng = np.random.default_rng(11)
    jobs = []
    for seed in range(6):
        left = synth.random_graph(700 + 50 * seed, 15, 900 + seed, p_extra=0.25, max_deg=3, max_span=12 + 5 * seed, p_dead=0.01)
        right = synth.random_graph(720 + 40 * seed, 15, 950 + seed, p_extra=0.25, max_deg=3, max_span=30 - 4 * seed)
        jobs.append((left, right, synth.random_model(15, seed), banded_around_diagonal(left, right, rng)))
    want = [oracle.dp_align(*j) for j in jobs]
    batch = pg.Batch(jobs)
    for rep in range(6):
        assert pg.lib().pagan_batch_debug_poison(batch._h) == 0
        batch.run()
        for k, got in enumerate(batch.fetch()):
            assert_same(got, want[k], "rep %d job %d" % (rep, k))
    batch.close()


def test_ring_kernel_wide_boxes_inside_a_band(pg, oracle):
    """A band that is narrow except for boxes wider than the ring (256 cells): the kernel leaves
    the ring for those diagonals (HBM operands, full drains) and re-enters it afterwards."""
    rng = np.random.default_rng(5)
    left = synth.random_graph(1500, 15, 41, p_extra=0.1, max_span=20)
    right = synth.random_graph(1500, 15, 42, p_extra=0.1, max_span=20)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    up = np.clip(np.arange(Lx) - 12, 0, None).astype(np.int32)
    lo = np.clip(np.arange(Lx) + 12, 0, Ly).astype(np.int32)
    for a in (300, 900):                      # two 340 x 340 boxes
        up[a:a + 340] = up[a]
        lo[a:a + 340] = lo[a + 339]
    band = abi.Band(up, lo)
    model = synth.random_model(15, 9)
    want = oracle.dp_align(left, right, model, band)
    batch = pg.Batch([(left, right, model, band)])
    for rep in range(3):
        pg.lib().pagan_batch_debug_poison(batch._h)
        batch.run()
        assert_same(batch.fetch()[0], want, "rep %d" % rep)
    batch.close()


def test_idle_arenas_are_kept_for_the_next_batch_and_can_be_released(pg, oracle):
    left, right, _ = pair_of_leaves(300, 5, sub=0.1, indel_start=0.03)
    model = synth.jc_like_dna_model(0.1)
    L = pg.lib()
    L.pagan_dp_release_cache()
    assert L.pagan_dp_cached_device_bytes(0) == 0
    want = oracle.dp_align(left, right, model)
    assert_same(pg.align(left, right, model), want)
    held = L.pagan_dp_cached_device_bytes(0)
    assert held > 0
    assert_same(pg.align(left, right, model), want)            # runs in the arena the first call left behind
    assert L.pagan_dp_cached_device_bytes(0) == held
    L.pagan_dp_release_cache()
    assert L.pagan_dp_cached_device_bytes(0) == 0
