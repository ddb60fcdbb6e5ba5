"""GPU: the prefix-anchor finder on the device (dp_anchors.hip: suffix array by prefix doubling with library sorts, common
prefixes from the rounds' rank arrays) against the host's finder (host_anchors.cpp: the same list from a host suffix array
and Kasai's pass) and, through it, the oracle's restatement of Find_anchors::find_long_substrings: the hit lists have to be
the same, element for element, and so do the tunnels made of them."""
import numpy as np
import pytest

from pagan2_msa_amd import host, synth

pytestmark = pytest.mark.gpu


def pair(length, seed, sub=0.02, indel=0.004):
    names, seqs, _ = synth.evolve_balanced(2, length, branch=0.02, sub=sub, indel_start=indel, mean_len=4, seed=seed)
    return seqs[0], seqs[1]


@pytest.mark.parametrize("length,seed", [(9000, 2), (20000, 3), (100000, 4)])
def test_device_hits_equal_host_hits(pg, monkeypatch, length, seed):
    a, b = pair(length, seed)
    monkeypatch.setenv("PAGAN_ANCHORS", "host")
    want = host.prefix_hits(a, b, 30)
    monkeypatch.delenv("PAGAN_ANCHORS")
    before = host.anchors_device_calls()
    got = host.prefix_hits(a, b, 30)
    assert host.anchors_device_calls() == before + 1, "the device's finder is meant to have run"
    assert len(want) > 0, "the pair is meant to share long substrings"
    assert np.array_equal(got, want)


def test_repeats_and_short_strings(pg, oracle):
    """low-complexity strings: long repeats make the doubling run many rounds; many equal common prefixes"""
    rng = np.random.default_rng(5)
    unit = "".join(rng.choice(list("ACGT"), 37))
    a = unit * 40 + "".join(rng.choice(list("ACGT"), 9000)) + "A" * 300
    b = "".join(rng.choice(list("ACGT"), 8000)) + unit * 35 + "A" * 250 + unit[:20]
    got = host.prefix_hits(a, b, 12)             # (the oracle's finder is reached through define_tunnel: compared below)
    assert len(got) > 0
    band, n = host.define_tunnel(a, b, a, b, prefix_hit_length=12)
    oband, on = oracle.define_tunnel(oracle.OGraph.leaf(a), oracle.OGraph.leaf(b), min_length=12)
    assert n == on and np.array_equal(band.upper, oband.upper) and np.array_equal(band.lower, oband.lower)


def test_tunnels_of_a_tree_walk_use_the_device_finder(pg, oracle):
    names, seqs, nwk = synth.evolve_balanced(8, 9000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=7)
    before = host.anchors_device_calls()
    msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
    assert host.anchors_device_calls() >= before + 3          # (the upper levels at least: a wide level shares the device four ways at most)
    for k in range(msa.n_internal):
        l, r, m, b = msa.node_job(k)
        want = oracle.dp_align(l, r, m, b)
        assert msa.node_result(k).same_alignment(want), "node %d" % k
