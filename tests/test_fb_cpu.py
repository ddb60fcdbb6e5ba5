"""CPU: the forward/backward restatement (oracle/oracle_fb.cpp).  The reference computes in probability space;
the product computes in log space.  Here the oracle's two arithmetics -- the same loops instantiated with plain
products/sums and with sums/log-sum-exp -- must agree to 1e-6 relative on inputs short enough for the
probability-space version not to underflow, the reference's own consistency check (forward total = backward total,
viterbi_alignment.cpp:351-355) must hold, and the model's probability-space view must match the product's."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, host, synth

REL = 1e-6          # north_star: per-node log-probability scores within 1e-6


def leaf_pair(oracle, n, seed, alphabet="ACGT", flags=0):
    _, seqs, _ = synth.evolve_balanced(2, n, branch=0.05, sub=0.06, indel_start=0.01, mean_len=3, seed=seed, alphabet=alphabet)
    alpha = oracle.protein_leaf_alphabet() if len(alphabet) == 20 else oracle.DNA_ALPHABET
    return [oracle.OGraph.leaf(s, alpha, flags).flatten() for s in seqs], seqs


def test_prob_model_view_matches_oracle_bits(oracle, pg):
    bf = np.array([0.3, 0.2, 0.2, 0.3], np.float32)
    for dt, kw in ((1, {"base_freq": bf}), (2, {})):
        a, b = host.model_prob(dt, 0.1, **kw), oracle.model_prob(dt, 0.1, **kw)
        assert a.table.tobytes() == b.table.tobytes()
        assert (a.gap_open, a.gap_ext, a.non_gap) == (b.gap_open, b.gap_ext, b.non_gap)
        # the log table the Viterbi pass uses is the log of this table (core entries logf of the float)
        lm = host.dna_model(bf, 0.1)[0] if dt == 1 else host.protein_model(0.1)[0]
        assert np.allclose(np.log(a.score.astype(np.float64)), lm.log_score, rtol=3e-7, atol=3e-7)


@pytest.mark.parametrize("n,seed", [(12, 1), (60, 2), (150, 3)])
def test_log_space_agrees_with_probability_space(oracle, pg, n, seed):
    (gl, gr), _ = leaf_pair(oracle, n, seed)
    mp = oracle.model_prob(1, 0.1, base_freq=[0.25] * 4)
    lf0, lb0, post0, f0 = oracle.fb(gl, gr, mp, log_space=False)
    lf1, lb1, post1, f1 = oracle.fb(gl, gr, mp, log_space=True)
    assert abs(lf0 - lf1) <= REL * abs(lf0) and abs(lb0 - lb1) <= REL * abs(lb0)
    assert np.allclose(post0, post1, rtol=1e-6, atol=1e-12)
    fin = np.isfinite(f0)
    assert np.array_equal(fin, np.isfinite(f1)) and np.allclose(f0[fin], f1[fin], rtol=1e-9, atol=1e-9)
    # VA:351-355: forward and backward totals agree
    assert abs(np.exp(lf1 - lb1) - 1) < 1e-9
    # every alignment passes through exactly one cell per ... at least: posteriors are probabilities
    assert post1.min() >= 0 and post1.max() <= 1 + 1e-9
    # all paths enter through (0,0): the start corner's posterior is 1
    assert abs(post1[0, 0, 2] - 1) < 1e-9


def test_probability_space_leaves_the_double_range_where_log_space_does_not(oracle, pg):
    (gl, gr), _ = leaf_pair(oracle, 1400, 4)
    mp = oracle.model_prob(1, 0.1, base_freq=[0.25] * 4)
    lf0, _, _, _ = oracle.fb(gl, gr, mp, log_space=False, matrices=False)
    lf1, lb1, _, _ = oracle.fb(gl, gr, mp, log_space=True, matrices=False)
    assert abs(lf1) > 710 and not np.isfinite(lf0)              # exp(lf1) is not a double: the reference's arithmetic overflows
    assert np.isfinite(lf1) and abs(lf1 - lb1) < 1e-8 * abs(lf1)


def test_band_graphs_and_protein(oracle, pg):
    # multi-edge graphs (an internal node over homopolymer leaves) and a tunnel
    (gl, gr), seqs = leaf_pair(oracle, 90, 5, flags=2)
    mp = oracle.model_prob(1, 0.1, base_freq=[0.25] * 4)
    up = np.maximum(np.arange(gl.n_sites - 1) - 25, 0).astype(np.int32)
    lo = np.minimum(np.arange(gl.n_sites - 1) + 25, gr.n_sites - 2).astype(np.int32)
    band = abi.Band(up, lo)
    for b in (None, band):
        lf0, lb0, p0, _ = oracle.fb(gl, gr, mp, band=b, log_space=False)
        lf1, lb1, p1, _ = oracle.fb(gl, gr, mp, band=b, log_space=True)
        assert abs(lf0 - lf1) <= REL * abs(lf0) and np.allclose(p0, p1, rtol=1e-6, atol=1e-12)
    assert oracle.fb(gl, gr, mp, band=band)[0] < oracle.fb(gl, gr, mp)[0]          # fewer paths inside the tunnel
    (pl, pr), _ = leaf_pair(oracle, 70, 6, alphabet="ARNDCQEGHILKMFPSTWYV")
    mpp = oracle.model_prob(2, 0.2)
    lf0, lb0, _, _ = oracle.fb(pl, pr, mpp, log_space=False)
    lf1, lb1, _, _ = oracle.fb(pl, pr, mpp, log_space=True)
    assert abs(lf0 - lf1) <= REL * abs(lf0) and abs(np.exp(lf1 - lb1) - 1) < 1e-9


def test_sampled_paths_follow_the_posterior(oracle, pg):
    (gl, gr), _ = leaf_pair(oracle, 25, 7)
    mp = oracle.model_prob(1, 0.1, base_freq=[0.25] * 4)
    _, _, post, logf = oracle.fb(gl, gr, mp)
    rng = np.random.default_rng(0)
    n = 4000
    hits = np.zeros_like(post)
    for _ in range(n):
        cells, end = oracle.sample_path(gl, gr, mp, logf, rng.random(gl.n_sites + gr.n_sites))
        assert tuple(cells[0]) == (end[1], end[2], end[0])
        for i, j, s in cells:
            hits[i, j, s] += 1
        # a path steps back through the lattice
        assert np.all(np.diff(cells[:, 0] + cells[:, 1]) < 0)
    freq = hits / n
    big = post > 0.05
    big[0, 0, :] = False                                        # the walk stops on reaching the start corner
    assert np.abs(freq[big] - post[big]).max() < 0.04          # Monte-Carlo error at n = 4000


def test_band_storage_long_tunnel_and_ragged_band(oracle, pg):
    """The restatement stores the band's cells only (a tunnel of 2 x 100 kb is 9e6 of 1e10 cells: bench.py's forward/backward
    baseline on cfg4's leaf pairs) and walks a column's band rows in the backward pass.  (i) 2 x 20 kb inside a tunnel of 31
    columns -- 4e8 cells as a full matrix, 6e5 in the band --: forward total = backward total (the reference's own check,
    VA:351-355).  (ii) A ragged band with a box, a short pair: the matrices inside the band equal those of the same pair with
    the cells outside the band ... computed as a full matrix restricted by a band that holds everything (the band argument is
    then only a different storage), and outside the band the outputs are -inf / 0."""
    (gl, gr), _ = leaf_pair(oracle, 20000, 11)
    mp = oracle.model_prob(1, 0.1, base_freq=[0.25] * 4)
    Lx, Ly = gl.n_sites - 1, gr.n_sites - 1
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    up = np.maximum.accumulate(np.maximum(centre - 15, 0)).astype(np.int32)
    lo = np.maximum.accumulate(np.minimum(centre + 15, Ly - 1)).astype(np.int32)
    up[0] = 0
    lo[-1] = Ly - 1
    lf, lb, _, _ = oracle.fb(gl, gr, mp, band=abi.Band(up, lo), matrices=False)
    assert np.isfinite(lf) and abs(lf - lb) <= 1e-9 * abs(lf)
    (sl, sr), _ = leaf_pair(oracle, 120, 12)
    Lx, Ly = sl.n_sites - 1, sr.n_sites - 1
    everything = abi.Band(np.zeros(Lx, np.int32), np.full(Lx, Ly - 1, np.int32))
    full = oracle.fb(sl, sr, mp)
    same = oracle.fb(sl, sr, mp, band=everything)
    assert full[0] == same[0] and full[1] == same[1] and np.array_equal(full[2], same[2]) and np.array_equal(full[3], same[3])
    centre = np.arange(Lx) * (Ly - 1) // (Lx - 1)
    up = np.maximum(centre - 9, 0)
    lo = np.minimum(centre + 9, Ly - 1)
    up[40:70] = up[40]
    lo[40:70] = min(lo[69] + 25, Ly - 1)
    up = np.maximum.accumulate(up).astype(np.int32)
    lo = np.maximum.accumulate(lo).astype(np.int32)
    up[0] = 0
    lo[-1] = Ly - 1
    lf, lb, post, logf = oracle.fb(sl, sr, mp, band=abi.Band(up, lo))
    assert abs(lf - lb) <= 1e-9 * abs(lf) and lf < full[0]
    outside = np.ones((Lx, Ly), bool)
    for i in range(Lx):
        outside[i, up[i]:lo[i] + 1] = False
    assert np.all(post[outside] == 0) and np.all(np.isneginf(logf[outside]))
    assert np.isfinite(logf[~outside][:, 2]).sum() > 0
