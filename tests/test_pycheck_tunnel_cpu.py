"""CPU: the second reading of the reference's tunnel builders (tests/pycheck_tunnel.py) against the oracle's restatement
(oracle/oracle_host.cpp) and the product (pagan2-msa_amd/csrc/host_anchors.cpp) on random hit lists: gapped child strings,
overlapping and crossing hits, both anchor widths the tests elsewhere use.  Three texts of the same reference functions
(find_anchors.cpp:225-317, 320-447, 633-861), compared entry by entry."""
import numpy as np
import pytest

import pycheck_tunnel as pt
from pagan2_msa_amd import host
from test_tunnel_cpu import gapped, random_hits


def distinct_scores(hits):
    """the reference's sorts leave ties undefined: give every hit its own score and its own start site"""
    h = hits.copy()
    h[:, 3] = h[:, 3] * 64 + np.arange(h.shape[0])
    _, first = np.unique(h[:, 0], return_index=True)
    return h[np.sort(first)]


@pytest.mark.parametrize("seed", range(16))
def test_order_conflicts_and_define_tunnel(oracle, pg, seed):
    rng = np.random.default_rng(100 + seed)
    g1, g2 = gapped(rng, 400 + 50 * seed, 0.02 * (seed % 3)), gapped(rng, 420 + 45 * seed, 0.02 * ((seed + 1) % 3))
    s1, s2 = g1.replace("-", ""), g2.replace("-", "")
    hits = distinct_scores(random_hits(rng, len(s1), len(s2), 8 + 4 * seed))
    for trim in (5, 2):
        want = oracle.order_conflicts(hits, len(g1), len(g2), trim)
        got = pt.check_hits_order_conflict(len(g1), len(g2), hits.tolist(), trim)
        assert np.array_equal(np.array(got, np.int32).reshape(-1, 4), want), "order conflicts, trim %d" % trim
        assert 0 < want.shape[0] <= hits.shape[0]
        for width in (15, 4):
            ob = oracle.tunnel_from_hits(want, g1, g2, width)
            up, lo = pt.define_tunnel(want.tolist(), g1, g2, width)
            assert np.array_equal(np.array(up, np.int32), ob.upper), "upper bound, width %d" % width
            assert np.array_equal(np.array(lo, np.int32), ob.lower), "lower bound, width %d" % width


@pytest.mark.parametrize("seed", range(16))
def test_tunnel_with_overlapping_hits(oracle, pg, seed):
    rng = np.random.default_rng(200 + seed)
    g1, g2 = gapped(rng, 500 + 40 * seed, 0.02 * (seed % 3)), gapped(rng, 520 + 30 * seed, 0.02 * ((seed + 1) % 3))
    n1, n2 = len(g1.replace("-", "")), len(g2.replace("-", ""))
    hits = random_hits(rng, n1, n2, 6 + 3 * seed)
    good = oracle.eliminate_bad_hits(hits, 50, 400)
    for width in (15, 4):
        ob, oblocks = oracle.tunnel_overlapping(good, g1, g2, width)
        pb, pblocks = host.define_tunnel_overlapping(good, g1, g2, width)
        up, lo, blocks = pt.define_tunnel_with_overlapping_hits(good.tolist(), g1, g2, width)
        assert np.array_equal(np.array(up, np.int32), ob.upper) and np.array_equal(np.array(lo, np.int32), ob.lower)
        assert np.array_equal(np.array(up, np.int32), pb.upper) and np.array_equal(np.array(lo, np.int32), pb.lower)
        assert np.array_equal(np.array(blocks, np.int32).reshape(-1, 4), oblocks) and np.array_equal(oblocks, pblocks)


def test_the_whole_prefix_anchor_chain_on_real_sequences(oracle, pg):
    """prefix anchors (product) -> order conflicts (second reading) -> define_tunnel (second reading) = the product's and the
    oracle's band for the same two sequences"""
    from pagan2_msa_amd import synth
    _, seqs, _ = synth.evolve_balanced(2, 5000, branch=0.02, sub=0.02, indel_start=0.002, mean_len=5, seed=11)
    a, b = seqs[0], seqs[1]
    hits = host.prefix_hits(a, b, 30)
    assert hits.shape[0] > 10
    kept = pt.check_hits_order_conflict(len(a), len(b), hits.tolist(), 5)
    up, lo = pt.define_tunnel(kept, a, b, 15)
    band, n = host.define_tunnel(a, b, a, b, 30, 5, 15)
    assert n == len(kept)
    assert np.array_equal(np.array(up, np.int32), band.upper) and np.array_equal(np.array(lo, np.int32), band.lower)
