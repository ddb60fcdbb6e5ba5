"""CPU: the tunnel from possibly overlapping hits (the reference's BLAST branch from the hit list onwards) and the
--force-gap rescue: product (csrc/host_anchors.cpp) against the oracle's literal restatement (oracle_host.cpp), plus
the invariants the aligner needs from a tunnel (monotone, holds both corners)."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, host, synth


def random_hits(rng, l1, l2, n):
    """Hits along a noisy diagonal, overlapping and crossing now and then, sorted by score like BLAST output."""
    hits = []
    for _ in range(n):
        ln = int(rng.integers(8, 60))
        s1 = int(rng.integers(0, max(1, l1 - ln)))
        s2 = int(np.clip(s1 + rng.integers(-40, 41) + (0 if rng.random() < 0.85 else rng.integers(-300, 301)), 0, l2 - ln))
        hits.append((s1, s2, ln, ln * 2 - int(rng.integers(0, 5))))
    hits.sort(key=lambda h: -h[3])
    return np.array(hits, np.int32)


def gapped(rng, n, frac):
    s = np.array(list("ACGT"))[rng.integers(0, 4, n)]
    out = []
    for c in s:
        out.append(c)
        if rng.random() < frac:
            out.extend("-" * int(rng.integers(1, 6)))
    return "".join(out)


def check_band(band, l1, l2):
    up, lo = band.upper, band.lower
    assert up.shape[0] == l1 + 1 and up[0] == 0 and lo[l1] == l2
    assert np.all(np.diff(up) >= 0) and np.all(np.diff(lo) >= 0) and np.all(up <= lo)
    assert up.min() >= 0 and lo.max() <= l2


@pytest.mark.parametrize("seed", range(12))
def test_overlapping_tunnel_matches_oracle(oracle, pg, seed):
    rng = np.random.default_rng(seed)
    g1, g2 = gapped(rng, 500 + 40 * seed, 0.02 * (seed % 3)), gapped(rng, 520 + 30 * seed, 0.02 * ((seed + 1) % 3))
    n1, n2 = len(g1.replace("-", "")), len(g2.replace("-", ""))
    hits = random_hits(rng, n1, n2, 6 + 3 * seed)
    a, b = host.drop_bad_hits(hits, 50, 400), oracle.eliminate_bad_hits(hits, 50, 400)
    assert np.array_equal(a, b) and 0 < a.shape[0] <= hits.shape[0]
    for width in (15, 4):
        (pb, pblocks), (ob, oblocks) = host.define_tunnel_overlapping(a, g1, g2, width), oracle.tunnel_overlapping(a, g1, g2, width)
        assert np.array_equal(pb.upper, ob.upper) and np.array_equal(pb.lower, ob.lower)
        assert np.array_equal(pblocks, oblocks)
        sizes = (pblocks[:, 2] - pblocks[:, 0]).astype(np.int64) * (pblocks[:, 3] - pblocks[:, 1])
        assert np.all(np.diff(sizes) >= 0)                       # ascending: the largest block is the last
        # --force-gap until nothing is left to replace, both ways
        for wide in (False, True):
            band_p, band_o, bl_p, bl_o = pb, ob, pblocks, oblocks
            rounds = 0
            while True:
                dp, band_p, bl_p = host.force_gap(band_p, bl_p, threshold=200, width=width, wide=wide)
                do, band_o, bl_o = oracle.force_gap(band_o, bl_o, threshold=200, width=width, wide=wide)
                assert dp == do
                assert np.array_equal(band_p.upper, band_o.upper) and np.array_equal(band_p.lower, band_o.lower)
                if not dp:
                    break
                rounds += 1
            assert rounds <= pblocks.shape[0]


def test_real_anchors_give_a_valid_tunnel_and_forced_gaps_shrink_it(oracle, pg):
    _, seqs, _ = synth.evolve_balanced(2, 6000, branch=0.02, sub=0.02, indel_start=0.002, mean_len=6, seed=3)
    rng = np.random.default_rng(8)
    junk = lambda n: "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])
    # 800 unrelated bases in the middle of each: a large empty block between the anchored flanks
    a, b = seqs[0][:2500] + junk(800) + seqs[0][3300:], seqs[1][:2500] + junk(800) + seqs[1][3300:]
    hits = host.prefix_hits(a, b, 20)
    assert hits.shape[0] > 20
    good = host.drop_bad_hits(hits)
    assert np.array_equal(good, oracle.eliminate_bad_hits(hits))
    band, blocks = host.define_tunnel_overlapping(good, a, b)
    ob, oblocks = oracle.tunnel_overlapping(good, a, b)
    assert np.array_equal(band.upper, ob.upper) and np.array_equal(band.lower, ob.lower) and np.array_equal(blocks, oblocks)
    check_band(band, len(a), len(b))
    assert blocks.shape[0] >= 1
    big = blocks[-1]
    assert (big[2] - big[0]) >= 600 and (big[3] - big[1]) >= 600   # the unrelated stretch is the largest empty block
    cells = lambda bd: int(np.sum(np.minimum(bd.lower[:-1], len(b)) - np.maximum(bd.upper[:-1], 0) + 1))
    before = cells(band)
    done, forced, rest = host.force_gap(band, blocks, threshold=40000)
    assert done and rest.shape[0] == blocks.shape[0] - 1
    assert cells(forced) < before - 100000                       # the block's interior left the tunnel
    od, oforced, _ = oracle.force_gap(ob, oblocks, threshold=40000)
    assert od and np.array_equal(forced.upper, oforced.upper) and np.array_equal(forced.lower, oforced.lower)
    check_band(forced, len(a), len(b))
    # below the threshold nothing is replaced
    assert not host.force_gap(band, blocks, threshold=10 ** 9)[0]
