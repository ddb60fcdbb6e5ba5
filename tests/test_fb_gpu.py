"""GPU: forward/backward full probability, posteriors and path sampling (dp_fb.hip) against the oracle's log-space
restatement (oracle/oracle_fb.cpp).  Sums are log-sum-exp in fp64 on both sides, the device's exp/log1p differ from
glibc's in the last bits, so the comparison is a tolerance: 1e-9 on logs (north_star asks 1e-6 for log-probability
scores), 1e-7 relative on posteriors."""
import numpy as np
import pytest

import pagan2_msa_amd as pgm
from pagan2_msa_amd import abi, host, synth

pytestmark = pytest.mark.gpu
LOG_TOL = 1e-9


def close_logs(a, b):
    fa, fb = np.isfinite(a), np.isfinite(b)
    return np.array_equal(fa, fb) and np.allclose(a[fa], b[fb], rtol=LOG_TOL, atol=LOG_TOL)


def check_pair(oracle, gl, gr, mp, band=None):
    fb = pgm.FullProbability(gl, gr, mp, band)
    lf, lb, post, logf = oracle.fb(gl, gr, mp, band=band)
    assert abs(fb.log_fwd - lf) <= LOG_TOL * max(1, abs(lf)), (fb.log_fwd, lf)
    assert abs(fb.log_bwd - lb) <= LOG_TOL * max(1, abs(lb)), (fb.log_bwd, lb)
    assert close_logs(fb.log_forward(), logf)
    assert np.allclose(fb.posterior(), post, rtol=1e-7, atol=1e-12)
    return fb, post, logf


def test_leaf_pairs_dna_and_protein(pg, oracle):
    _, seqs, _ = synth.evolve_balanced(2, 300, branch=0.05, sub=0.06, indel_start=0.01, mean_len=3, seed=41)
    gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
    mp = host.model_prob(1, 0.1, base_freq=[0.3, 0.2, 0.2, 0.3])
    fb, post, _ = check_pair(oracle, gl, gr, mp)
    assert abs(np.exp(fb.log_fwd - fb.log_bwd) - 1) < 1e-9              # the reference's own check, VA:351-355
    cells = np.array([[2, 0, 0], [2, 10, 11], [0, 5, 4], [1, 299, 300], [2, 1000, 3]], np.int32)
    want = [post[0, 0, 2], post[10, 11, 2], post[5, 4, 0], post[299, 300, 1] if post.shape[1] > 300 else 0.0, 0.0]
    assert np.allclose(fb.posterior_cells(cells), want, rtol=1e-7, atol=1e-12)
    aa = "ARNDCQEGHILKMFPSTWYV"
    _, ps, _ = synth.evolve_balanced(2, 200, branch=0.05, sub=0.08, indel_start=0.01, mean_len=3, seed=42, alphabet=aa)
    leaf_alpha, _ = host.alphabets(2)
    pl, pr = (host.HGraph.leaf(s, leaf_alpha).flatten() for s in ps)
    check_pair(oracle, pl, pr, host.model_prob(2, 0.2))


def test_long_input_stays_in_range(pg, oracle):
    """2 x 3 kb: the reference's probability-space products (log-odds "probabilities", mostly > 1) leave the range of a
    double here -- exp(2900) -- where the log-space pass does not care."""
    _, seqs, _ = synth.evolve_balanced(2, 3000, branch=0.02, sub=0.02, indel_start=0.004, mean_len=4, seed=43)
    gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
    band, _ = host.define_tunnel(seqs[0], seqs[1], seqs[0], seqs[1])
    mp = host.model_prob(1, 0.04, base_freq=[0.25] * 4)
    fb = pgm.FullProbability(gl, gr, mp, band)
    lf, lb, _, _ = oracle.fb(gl, gr, mp, band=band, matrices=False)
    assert abs(fb.log_fwd) > 710 and abs(fb.log_fwd - lf) <= LOG_TOL * abs(lf) and abs(fb.log_bwd - lb) <= LOG_TOL * abs(lb)
    assert abs(fb.log_fwd - fb.log_bwd) < 1e-7


def test_graph_vs_graph_with_tunnel_and_sampling(pg, oracle):
    """Internal nodes of a small tree (multi-edge sites, skipped sites), inside their tunnels; sampled paths."""
    names, seqs, nwk = synth.evolve_balanced(8, 250, branch=0.03, sub=0.03, indel_start=0.01, mean_len=4, seed=44)
    msa = host.Msa(names, seqs, nwk, use_anchors=1, prefix_hit_length=15).align()
    bf = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    bf /= bf.sum()
    rng = np.random.default_rng(5)
    multi = 0
    for k in (4, 5, 6):                                   # the two level-2 nodes and the root
        left, right, model, band = msa.node_job(k)
        multi += int((np.diff(left.bwd_off) > 1).sum() + (np.diff(right.bwd_off) > 1).sum())
        mp = host.model_prob(1, msa.node_info(k).dist, base_freq=bf)
        fb, post, logf = check_pair(oracle, left, right, mp, band)
        for _ in range(3):
            u = rng.random(left.n_sites + right.n_sites)
            res, visited = fb.sample_path(u)
            want, end = oracle.sample_path(left, right, mp, logf, u)
            assert np.array_equal(visited, want)
            assert res.status == 0 and res.score == fb.log_fwd
            # the columns consume every site of both children once, in order
            assert [c for c in res.cols[:, 0] if c >= 0] == list(range(1, left.n_sites - 1))
            assert [c for c in res.cols[:, 1] if c >= 0] == list(range(1, right.n_sites - 1))
            # matched / gapped columns are exactly the visited cells
            real = res.cols[res.cols[:, 2] <= 4]
            assert real.shape[0] == visited.shape[0]
        # a parent graph can be built from a sampled path like from a Viterbi path
        info = msa.node_info(k)
        hp = host.HGraph.parent(msa.node_graph(info.left), msa.node_graph(info.right), res, info.dist / 2, info.dist / 2,
                                oracle.dna_parsimony(), 4)
        assert hp.flatten().n_sites == res.cols.shape[0] + 2
    assert multi > 0


def test_errors(pg):
    g = host.HGraph.leaf("ACGT").flatten()
    bad = abi.ModelProb(np.ones((15, 15), np.float32), 0.0, 0.5, 0.9)
    with pytest.raises(pgm.PaganError) as e:
        pgm.FullProbability(g, g, bad)
    assert e.value.code == abi.PAGAN_E_MODEL


def test_wide_pairs_spread_a_diagonal_over_several_workgroups(pg, oracle, monkeypatch):
    """Full matrices of internal nodes (multi-edge sites, ~900-cell diagonals): the sweeps cut the matrix into 64 x 64 blocks that
    a grid of one-wave workgroups works through block anti-diagonal by block anti-diagonal (dp_fb.hip, pg_fb_forward_tiled /
    pg_fb_backward_tiled: a counter per block diagonal, no barrier per cell diagonal); same logs as the oracle and as the
    one-workgroup sweeps (PAGAN_FB_GROUPS=1) to the comparison's tolerance, forward total = backward total."""
    names, seqs, nwk = synth.evolve_balanced(4, 900, branch=0.04, sub=0.04, indel_start=0.01, mean_len=4, seed=45)
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
    bf = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    bf /= bf.sum()
    left, right, _model, band = msa.node_job(2)           # the root: graph against graph
    assert band is None and left.n_sites > 850 and (np.diff(left.bwd_off) > 1).sum() > 0
    mp = host.model_prob(1, msa.node_info(2).dist, base_freq=bf)
    fb, _post, logf = check_pair(oracle, left, right, mp)
    assert abs(fb.log_fwd - fb.log_bwd) <= 1e-7 * abs(fb.log_fwd)
    wide_f, wide_b = fb.log_forward().copy(), fb.log_backward().copy()
    monkeypatch.setenv("PAGAN_FB_GROUPS", "1")
    one = pgm.FullProbability(left, right, mp)
    assert close_logs(one.log_forward(), wide_f) and close_logs(one.log_backward(), wide_b)


def test_a_batch_of_pairs_in_one_launch_per_sweep(pg, oracle):
    """pagan_fb_run_batch: the forward sweeps of all wide pairs of a tree in ONE launch, the backward sweeps in another (a pair's
    workgroup count cut to its share of the device), the narrow pairs beside them on their own kernels -- every pair's logs
    equal the one-pair call's to the comparison's tolerance, its totals the oracle's."""
    names, seqs, nwk = synth.evolve_balanced(8, 700, branch=0.04, sub=0.04, indel_start=0.01, mean_len=4, seed=46)
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
    bf = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    bf /= bf.sum()
    pairs = []
    for k in range(msa.n_internal):
        left, right, _model, band = msa.node_job(k)
        pairs.append((left, right, host.model_prob(1, msa.node_info(k).dist, base_freq=bf), band))
    # a narrow (banded) pair among them: the first leaf pair again behind a band of 40 columns
    l0, r0, mp0, _ = pairs[0]
    Lx, Ly = l0.n_sites - 1, r0.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum(centre - 20, 0); lower = np.minimum(centre + 20, Ly - 1)
    upper[0] = 0; lower[-1] = Ly - 1
    pairs.append((l0, r0, mp0, abi.Band(upper, lower)))
    batch = pgm.full_probability_batch(pairs)
    assert len(batch) == len(pairs)
    for k, (left, right, mp, band) in enumerate(pairs):
        one = pgm.FullProbability(left, right, mp, band)
        assert close_logs(batch[k].log_forward(), one.log_forward()) and close_logs(batch[k].log_backward(), one.log_backward()), k
        assert abs(batch[k].log_fwd - one.log_fwd) <= 1e-9 * abs(one.log_fwd) and abs(batch[k].log_bwd - one.log_bwd) <= 1e-9 * abs(one.log_bwd)
        lf, lb, _post, _logf = oracle.fb(left, right, mp, band=band)
        assert abs(batch[k].log_fwd - lf) <= LOG_TOL * max(1, abs(lf)) and abs(batch[k].log_bwd - lb) <= LOG_TOL * max(1, abs(lb)), k
        one.close()
    for fb in batch:
        fb.close()


def test_a_batch_larger_than_one_launch_holds(pg, oracle):
    """more wide pairs than one launch takes at eight workgroups a sweep (the batch goes in chunks): every pair's totals the oracle's"""
    mp = host.model_prob(1, 0.1, base_freq=[0.3, 0.2, 0.2, 0.3])
    pairs = []
    for k in range(52):
        _, seqs, _ = synth.evolve_balanced(2, 270 + 3 * k, branch=0.05, sub=0.06, indel_start=0.01, mean_len=3, seed=500 + k)
        gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
        pairs.append((gl, gr, mp, None))
    batch = pgm.full_probability_batch(pairs)
    for k, (gl, gr, mp_, band) in enumerate(pairs):
        lf, lb, _post, _logf = oracle.fb(gl, gr, mp_, band=band, matrices=False)
        assert abs(batch[k].log_fwd - lf) <= LOG_TOL * max(1, abs(lf)) and abs(batch[k].log_bwd - lb) <= LOG_TOL * max(1, abs(lb)), k
        batch[k].close()


def _random_tunnel(rng, Lx, Ly, lo_half, hi_half):
    half = rng.integers(lo_half, hi_half, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    return abi.Band(upper.astype(np.int32), lower.astype(np.int32))


def test_tunnels_on_the_block_schedule(pg, oracle, monkeypatch):
    """A long narrow tunnel runs as 64 x 64 blocks too (round 5: fb_live_rows picks the few blocks the band has on a block
    anti-diagonal).  PAGAN_FB_BAND_MIN_ND=0 sends pairs of any length there: leaf pairs and graph pairs with multi-edge sites
    behind random tunnels (5 to 70 columns either side: narrower and wider than a block, edges that leave it), every cell of both
    matrices against the oracle; the same pairs on the one-workgroup kernels ("off") give the same sums."""
    rng = np.random.default_rng(77)
    names, seqs, nwk = synth.evolve_balanced(4, 700, branch=0.04, sub=0.04, indel_start=0.01, mean_len=4, seed=46)
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
    bf = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    bf /= bf.sum()
    cases = []
    for k in range(msa.n_internal):
        left, right, _model, _band = msa.node_job(k)
        mp = host.model_prob(1, msa.node_info(k).dist, base_freq=bf)
        for lo_half, hi_half in ((5, 12), (20, 70)):
            cases.append((left, right, mp, _random_tunnel(rng, left.n_sites - 1, right.n_sites - 1, lo_half, hi_half)))
    assert sum(int((np.diff(c[0].bwd_off) > 1).sum() + (np.diff(c[1].bwd_off) > 1).sum()) for c in cases) > 0
    monkeypatch.setenv("PAGAN_FB_BAND_MIN_ND", "0")
    blocked = []
    for left, right, mp, band in cases:
        fb = pgm.FullProbability(left, right, mp, band)
        lf, lb, post, logf = oracle.fb(left, right, mp, band=band)
        assert abs(fb.log_fwd - lf) <= LOG_TOL * max(1, abs(lf)) and abs(fb.log_bwd - lb) <= LOG_TOL * max(1, abs(lb))
        assert close_logs(fb.log_forward(), logf)
        assert np.allclose(fb.posterior(), post, rtol=1e-7, atol=1e-12)
        blocked.append((fb.log_fwd, fb.log_bwd))
        fb.close()
    # ... and through the batch call (all pairs' sweeps in one launch per direction)
    fbs = pgm.full_probability_batch(cases)
    for fb, (f, b) in zip(fbs, blocked):
        assert fb.log_fwd == f and fb.log_bwd == b
        fb.close()
    monkeypatch.setenv("PAGAN_FB_BAND_MIN_ND", "off")
    for (left, right, mp, band), (f, b) in zip(cases, blocked):
        fb = pgm.FullProbability(left, right, mp, band)
        assert abs(fb.log_fwd - f) <= LOG_TOL * max(1, abs(f)) and abs(fb.log_bwd - b) <= LOG_TOL * max(1, abs(b))
        fb.close()


def test_long_tunnel_between_leaves_by_default(pg, oracle, monkeypatch):
    """2 x 6 kb inside define_tunnel's band (12 k cell diagonals: beyond the default thresholds): plain sequences take the
    LDS-ring sweeps (groups 0), with PAGAN_FB_RING=0 the block schedule (groups > 1); totals against the oracle, both."""
    _, seqs, _ = synth.evolve_balanced(2, 6000, branch=0.02, sub=0.02, indel_start=0.003, mean_len=4, seed=47)
    gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
    band, _ = host.define_tunnel(seqs[0], seqs[1], seqs[0], seqs[1])
    mp = host.model_prob(1, 0.04, base_freq=[0.25] * 4)
    lf, lb, _, _ = oracle.fb(gl, gr, mp, band=band, matrices=False)
    for ring, want_groups in (("1", 0), ("0", None)):
        monkeypatch.setenv("PAGAN_FB_RING", ring)
        fb = pgm.FullProbability(gl, gr, mp, band)
        assert fb.groups == want_groups if want_groups is not None else fb.groups > 1
        assert abs(fb.log_fwd - lf) <= LOG_TOL * abs(lf) and abs(fb.log_bwd - lb) <= LOG_TOL * abs(lb)
        assert abs(fb.log_fwd - fb.log_bwd) < 1e-7
        fb.close()


def test_ring_sweeps_cell_by_cell(pg, oracle, monkeypatch):
    """The LDS-ring sweeps (PAGAN_FB_RING_MIN_ND=0: pairs of any length) on leaf pairs behind random tunnels -- 5 to 12, 20 to 70
    100 to 180 and 270 to 330 columns either side: 64 to 1,024 rows a workgroup, rows re-used by their threads many times over --, DNA and
    protein (a score table that does not fit LDS), no tunnel at all on a short pair; every cell of both matrices and the
    posteriors against the oracle, one call per pair and all pairs in one batch."""
    rng = np.random.default_rng(78)
    monkeypatch.setenv("PAGAN_FB_RING_MIN_ND", "0")
    cases = []
    _, seqs, _ = synth.evolve_balanced(2, 900, branch=0.04, sub=0.05, indel_start=0.01, mean_len=4, seed=48)
    gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
    mp = host.model_prob(1, 0.08, base_freq=[0.3, 0.2, 0.2, 0.3])
    for lo_half, hi_half in ((5, 12), (20, 70), (100, 180)):
        cases.append((gl, gr, mp, _random_tunnel(rng, gl.n_sites - 1, gr.n_sites - 1, lo_half, hi_half)))
    # a tunnel wider than 512 cells on its widest diagonal: 1,024 threads, one a row, 113 KB of LDS
    _, wide, _ = synth.evolve_balanced(2, 1400, branch=0.04, sub=0.05, indel_start=0.01, mean_len=4, seed=51)
    wl, wr = (host.HGraph.leaf(s).flatten() for s in wide)
    cases.append((wl, wr, mp, _random_tunnel(rng, wl.n_sites - 1, wr.n_sites - 1, 270, 330)))
    _, short, _ = synth.evolve_balanced(2, 150, branch=0.04, sub=0.05, indel_start=0.01, mean_len=3, seed=49)
    sl, sr = (host.HGraph.leaf(s).flatten() for s in short)
    cases.append((sl, sr, mp, None))
    aa = "ARNDCQEGHILKMFPSTWYV"
    _, ps, _ = synth.evolve_balanced(2, 400, branch=0.05, sub=0.08, indel_start=0.01, mean_len=3, seed=50, alphabet=aa)
    leaf_alpha, _ = host.alphabets(2)
    pl, pr = (host.HGraph.leaf(s, leaf_alpha).flatten() for s in ps)
    cases.append((pl, pr, host.model_prob(2, 0.2), _random_tunnel(rng, pl.n_sites - 1, pr.n_sites - 1, 10, 40)))
    singles = []
    for left, right, m, band in cases:
        fb = pgm.FullProbability(left, right, m, band)
        assert fb.groups == 0
        lf, lb, post, logf = oracle.fb(left, right, m, band=band)
        assert abs(fb.log_fwd - lf) <= LOG_TOL * max(1, abs(lf)) and abs(fb.log_bwd - lb) <= LOG_TOL * max(1, abs(lb))
        assert close_logs(fb.log_forward(), logf)
        assert np.allclose(fb.posterior(), post, rtol=1e-7, atol=1e-12)
        singles.append((fb.log_fwd, fb.log_bwd))
        fb.close()
    fbs = pgm.full_probability_batch(cases)
    for fb, (f, b), (left, right, m, band) in zip(fbs, singles, cases):
        assert fb.groups == 0 and fb.log_fwd == f and fb.log_bwd == b
        _lf, _lb, post, _logf = oracle.fb(left, right, m, band=band)
        assert np.allclose(fb.posterior(), post, rtol=1e-7, atol=1e-12)
        fb.close()
