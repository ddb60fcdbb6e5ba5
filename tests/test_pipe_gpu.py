"""GPU parity of the register-wavefront fill kernel (dp_pipe.hip) on the code paths the tree workloads
reach rarely: far edges (class 2), general steps (3), wide boxes (4), waves waking and sleeping; plus the
older LDS ring kernel behind its switch and the negative-zero route."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, synth

pytestmark = pytest.mark.gpu


def same(a, b, what=""):
    assert a.status == b.status, what
    assert np.float64(a.score).tobytes() == np.float64(b.score).tobytes(), what + " score %r != %r" % (a.score, b.score)
    assert a.end == b.end, what
    assert np.array_equal(a.cols, b.cols), what + " columns differ"
    assert np.array_equal(a.left_used, b.left_used) and np.array_equal(a.right_used, b.right_used), what


def banded_job(seed, n=700, max_span=40, box=True, p_dead=0.0):
    rng = np.random.default_rng(seed)
    if box and seed == 3: n = max(n, 2000)                        # (its box of 480 rows must not make the job a wide one)
    left = synth.random_graph(n, 15, 10 + seed, p_extra=0.08, p_dead=p_dead, max_span=max_span)
    right = synth.random_graph(n + 30, 15, 20 + seed, p_extra=0.08, p_dead=p_dead, max_span=max_span)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(8, 40, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    if box:
        # seed 1: wider than 352 cells (the wide ring of 9 rows x 512 positions); seed 3: wider than the record windows too (class 5)
        at, rows, jump = {1: (300, 390, 420), 3: (200, 480, 60)}.get(seed, (300, 260, 300))
        upper[at:at + rows] = upper[at]
        lower[at:at + rows] = np.minimum(lower[at - 1 + rows] + jump, Ly - 1)
        lower = np.maximum.accumulate(lower)
        upper = np.maximum.accumulate(upper)
    return left, right, synth.random_model(15, seed), abi.Band(upper, lower)


@pytest.mark.parametrize("seed", range(4))
def test_far_edges_general_steps_and_a_wide_box(pg, oracle, seed):
    left, right, model, band = banded_job(seed, max_span=40 if seed < 2 else 8)
    cls, waves = pg.debug_plan(left, right, band)
    assert set(np.unique(cls)) >= ({2, 3, 4} if seed < 2 else {1, 3, 4}), "the job is meant to reach these classes"
    assert sum(len(w) for w in waves) >= 4, "waves are meant to sleep and wake"
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


def sweep_case(seed, case):
    """the job tests/diagnostics/sweep_parity.py makes for (PG_SWEEP_SEED, case)"""
    rng = np.random.default_rng(seed + case)
    n = int(rng.integers(150, 1400))
    span = int(rng.choice([4, 8, 17, 19, 25, 40, 80]))
    p_extra = float(rng.choice([0.02, 0.08, 0.3]))
    left = synth.random_graph(n, 15, 3000 + case, p_extra=p_extra, max_deg=int(rng.integers(2, 5)), max_span=span, p_dead=float(rng.choice([0, 0, 0.01])))
    right = synth.random_graph(n + int(rng.integers(-40, 60)), 15, 4000 + case, p_extra=p_extra, max_deg=int(rng.integers(2, 5)), max_span=span)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(3, 60, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    for _ in range(int(rng.integers(0, 3))):
        a = int(rng.integers(10, max(11, Lx - 450))); rows = int(rng.integers(30, 440)); jump = int(rng.integers(30, 460))
        b = min(a + rows, Lx - 1)
        upper[a:b] = upper[a]; lower[a:b] = min(lower[b - 1] + jump, Ly - 1)
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    upper[0] = 0; lower[-1] = Ly - 1
    flags = int(rng.choice([0, 0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN]))
    return left, right, synth.random_model(15, case), abi.Band(upper, lower), flags


def test_a_wide_run_whose_first_row_moved_on_its_first_diagonal(pg, oracle):
    """Found by a fresh seed of the sweep (5000 / 739) at the end of round 5: a run of class 4 diagonals directly behind a run of
    class 5 ones, the band's first row one further than on the diagonal before, and in that row a cell of the general rules (the
    last column).  "A row above the previous diagonal's band reads -inf" compared with the run's own first row on the run's
    first diagonal and dropped the cell's operand (i-1, j)."""
    left, right, model, band, flags = sweep_case(5000, 739)
    cls = pg.debug_far(left, right, band)[4] & 15
    first4 = [d for d in range(1, len(cls)) if cls[d] == 4 and cls[d - 1] == 5]
    assert first4, "the case is meant to have a class 4 run directly behind a class 5 run"
    same(pg.align(left, right, model, band, flags=flags), oracle.dp_align(left, right, model, band, flags=flags))


@pytest.mark.parametrize("env", [{"PAGAN_DP_WIDE7": "0"}, {"PAGAN_DP_AFTER_WIDE": "reach"}, {"PAGAN_DP_HIST": "0", "PAGAN_DP_THREE": "0"}])
@pytest.mark.parametrize("seed", [0, 1])
def test_wide_boxes_with_the_round_4_paths(pg, oracle, monkeypatch, env, seed):
    """The A/B switches keep round 4's paths alive: wide runs on the four compute waves only (what short runs still take),
    REACH - 1 general steps behind a wide run, no far histories / third pass -- each against the oracle on the box jobs."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    left, right, model, band = banded_job(seed, max_span=40)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d %s" % (seed, env))


@pytest.mark.parametrize("seed,p_dead,max_span", [(11, 0.05, 6), (12, 0.2, 6), (13, 0.4, 30), (14, 0.1, 40)])
def test_sites_without_bwd_edges_stay_on_the_multi_edge_paths(pg, oracle, seed, p_dead, max_span):
    """Upper levels of deep trees: 18-40 % of the sites have no bwd edge (tools/probe_plan.py on cfg5); their
    cells take the item loop of the multi-edge steps instead of sending the whole diagonal to the general path."""
    left, right, model, band = banded_job(seed, n=500, max_span=max_span, box=False, p_dead=p_dead)
    cls, _ = pg.debug_plan(left, right, band)
    assert (cls == 3).sum() < 100 and ((cls == 1) | (cls == 2)).sum() > 800
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


@pytest.mark.parametrize("flags", [abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN])
def test_option_bits_on_a_banded_job(pg, oracle, flags):
    left, right, model, band = banded_job(7, n=400, box=False)
    same(pg.align(left, right, model, band, flags=flags), oracle.dp_align(left, right, model, band, flags=flags))


def test_ring_kernel_behind_its_switch(pg, oracle, monkeypatch):
    left, right, model, band = banded_job(3)
    monkeypatch.setenv("PAGAN_DP_FILL", "ring")
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band))


def test_pipe_and_ring_store_identical_scores(pg, monkeypatch):
    job = banded_job(5)
    scores = {}
    for kernel in ("ring", "pipe"):
        monkeypatch.setenv("PAGAN_DP_FILL", kernel)
        b = pg.Batch([job])
        b.run(); b.sync()
        scores[kernel] = b.debug_scores(0)
        b.close()
    assert np.array_equal(scores["ring"].view(np.int64), scores["pipe"].view(np.int64))


def test_negative_zero_parameter_keeps_the_comparing_kernel(pg, oracle):
    left, right, model, band = banded_job(2, n=300, box=False)
    t = model.log_score.copy()
    t[3, 3] = np.float32(-0.0)
    m2 = abi.Model(t, *model.params)
    cls, _ = pg.debug_plan(left, right, band)        # the planner itself does not look at the model
    assert cls.size == left.n_sites + right.n_sites - 3
    same(pg.align(left, right, m2, band), oracle.dp_align(left, right, m2, band))


def poisoned_run(pg, job, flags=0):
    b = pg.Batch([job], flags=flags)
    pg.lib().pagan_batch_debug_poison(b._h)
    b.run(); b.sync()
    return b


@pytest.mark.parametrize("seed,kw", [(0, {}), (1, {}), (2, dict(max_span=8)), (12, dict(n=500, max_span=6, box=False, p_dead=0.2)),
                                      (13, dict(n=500, max_span=30, box=False, p_dead=0.4))])
def test_backpointer_pass_writes_what_the_comparing_kernels_write(pg, monkeypatch, seed, kw):
    """pg_backptr re-derives every cell's back-pointers from the stored scores (the banded fill's hot loop stores scores
    only).  Reference for the words: the older ring kernel, which evaluates every candidate with compare-and-replace
    in the reference's order and writes its own back-pointers; every word of every cell must agree -- classes 0-5, dead
    sites, far edges."""
    job = banded_job(seed, **kw)
    monkeypatch.setenv("PAGAN_DP_COMPACT", "0")      # the device arrays are read back as the caller's matrices
    words = {}
    for mode, kernel in (("fill", "ring"), ("pass", "pipe")):
        monkeypatch.setenv("PAGAN_DP_FILL", kernel)
        monkeypatch.setenv("PAGAN_DP_BP", mode)
        b = poisoned_run(pg, job)
        words[mode] = (b.debug_backptrs(0), b.debug_scores(0))
        b.close()
    assert np.array_equal(words["fill"][1].view(np.int64), words["pass"][1].view(np.int64)), "scores differ"
    bad = np.argwhere(words["fill"][0] != words["pass"][0])
    assert bad.size == 0, "first differing back-pointers (cell, state): %s" % bad[:5].tolist()


@pytest.mark.parametrize("flags", [0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN])
def test_backpointer_pass_under_the_option_bits(pg, oracle, monkeypatch, flags):
    job = banded_job(7, n=400, box=False)
    words = {}
    for mode, kernel in (("fill", "ring"), ("pass", "pipe")):
        monkeypatch.setenv("PAGAN_DP_FILL", kernel)
        monkeypatch.setenv("PAGAN_DP_BP", mode)
        b = poisoned_run(pg, job, flags)
        words[mode] = b.debug_backptrs(0)
        if mode == "pass":
            same(b.fetch()[0], oracle.dp_align(*job, flags=flags))
        b.close()
    assert np.array_equal(words["fill"], words["pass"])


@pytest.mark.parametrize("seed", [0, 2])
def test_a_codon_sized_table_on_the_banded_kernel(pg, oracle, seed):
    """1892 states -- the size of the reference's codon alphabet (61 codons, NNN, 1830 codon pairs;
    model_factory.cpp:839-897) -- i.e. a 14 MB score table that stays in HBM / L2: the banded kernel's large-table path
    (model scores gathered per diagonal by the assist waves), class 0..4 diagonals included."""
    S = 1892
    left, right, _, band = banded_job(seed, max_span=40 if seed < 2 else 8)
    rng = np.random.default_rng(100 + seed)
    left.state[1:-1] = rng.integers(0, S, left.n_sites - 2)
    right.state[1:-1] = rng.integers(0, S, right.n_sites - 2)
    model = synth.random_model(S, seed)
    assert pg.debug_route(left, right, model, band)[0] == "pg_fill_pipe (large table)"
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


def test_back_pointers_are_written_behind_the_fill_and_equal_the_pass_afterwards(pg, oracle, monkeypatch):
    """The follower workgroups of pg_fill_pipe write the back-pointers of the diagonals whose scores have landed while the
    fill is still running (poisoned arena: a chunk read too early would see NaN scores); pg_backptr afterwards only writes
    what they left.  Same words as with the followers switched off (PAGAN_DP_FOLLOW=0: pg_backptr writes everything), and
    most chunks must have come from the followers."""
    jobs = [banded_job(s) for s in (0, 1, 3)] + [banded_job(2, max_span=8)]
    words, followed = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PAGAN_DP_FOLLOW", mode)
        b = pg.Batch(jobs)
        for rep in range(2):                             # the second run finds the first one's flags and words in the arena
            pg.lib().pagan_batch_debug_poison(b._h)
            b.run(); b.sync()
        words[mode] = [b.debug_backptrs(k) for k in range(len(jobs))]
        followed[mode] = [b.debug_followed(k) for k in range(len(jobs))]
        res = b.fetch()
        for k, (left, right, model, band) in enumerate(jobs):
            same(res[k], oracle.dp_align(left, right, model, band), "job %d, followers %s" % (k, mode))
        b.close()
    for k in range(len(jobs)):
        assert np.array_equal(words["0"][k], words["1"][k]), "job %d" % k
        assert followed["0"][k][0] == 0 and followed["0"][k][1] > 50
        done, total = followed["1"][k]
        assert total == followed["0"][k][1] and done >= 0.9 * total, "job %d: the followers wrote %d of %d chunks" % (k, done, total)


def _poke_on_path(pg, b, t):
    """Asks for the back-pointer of the t-th visited cell of job 0's last traceback to name another predecessor state."""
    tr = b.debug_trace(0, t + 2)
    i, j, w = (int(x) for x in tr[t])
    vit, frm = w & 3, int(tr[t + 1][2]) & 3              # a cell's `from` label is the next visited cell's matrix
    word = ((w & 0xffffffff) & ~3) | ((frm + 1) % 3)
    b.debug_poke_bp(0, i, j, vit, word)
    return i, j, vit


@pytest.mark.parametrize("kind", ["banded", "tiled"])
def test_a_wrong_back_pointer_on_the_path_is_seen_and_the_batch_runs_again(pg, oracle, monkeypatch, kind):
    """Every cell the traceback visited is re-evaluated from the stored scores (pg_trace_check): a back-pointer that names
    another predecessor than the scores give -- what a follower workgroup reading a score too early would write -- must not
    become an alignment.  fetch() runs the batch once more with every pointer written after the fill; the hook's word is
    gone by then, so the result is the oracle's and one re-run is counted."""
    job = banded_job(3, box=False) if kind == "banded" else banded_job(3, box=False)[:3] + (None,)
    want = oracle.dp_align(*job)
    b = pg.Batch([job])
    b.run(); b.sync()
    same(b.fetch()[0], want)
    assert b.debug_reruns() == 0
    n_real = int(np.isin(want.cols[:, 2], (2, 3, 4)).sum())       # the traceback visits one cell per real column
    for t in (5, 200, n_real - 40):
        before = b.debug_reruns()
        _poke_on_path(pg, b, t)
        b.run(); b.sync()
        same(b.fetch()[0], want, "cell %d of the path" % t)
        assert b.debug_reruns() == before + 1, "the corrupted pointer of visited cell %d went unnoticed" % t
    b.close()


def test_a_wrong_back_pointer_is_an_error_when_the_rerun_is_switched_off(pg, monkeypatch):
    job = banded_job(3, box=False)
    b = pg.Batch([job])
    b.run(); b.sync(); b.fetch()
    _poke_on_path(pg, b, 300)
    monkeypatch.setenv("PAGAN_DP_RERUN", "0")
    b.run(); b.sync()
    with pytest.raises(pg.PaganError) as e:
        b.fetch()
    assert e.value.code == abi.PAGAN_E_INTERNAL
    b.close()
