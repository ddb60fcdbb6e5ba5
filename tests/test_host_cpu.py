"""CPU tests of the host-side pieces (graph builder, anchors, model) against the oracle's
restatement.  No GPU: the alignment paths fed to the builders come from the oracle DP."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, host, synth


def same_graph(hg, og, what=""):
    a, b = hg.flatten(), og.flatten()
    for f in ("state", "bwd_off", "bwd_src", "bwd_eid"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), "%s: %s differs" % (what, f)
    assert a.bwd_logw.tobytes() == b.bwd_logw.tobytes(), what + ": bwd_logw bits differ"
    assert a.n_edges == b.n_edges
    (sa, sd, ea, ef), (sb, sdb, eb, efb) = hg.attrs(), og.attrs()
    assert np.array_equal(sa, sb), what + ": site attributes differ"
    assert sd.tobytes() == sdb.tobytes(), what + ": site distances differ"
    assert np.array_equal(ea, eb), what + ": edge attributes differ"
    assert ef.tobytes() == efb.tobytes(), what + ": edge weights differ"
    (fo, fe), (fob, feb) = hg.fwd(), og.fwd()
    assert np.array_equal(fo, fob) and np.array_equal(fe, feb), what + ": fwd lists differ"


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_leaf_graphs(oracle, pg, flags):
    for s in ["ACGT", "A", "AAAAAAAACCCCCGTTTGGGGGGGGGGGGAC", "ACGTTTTTTACGGGGACCCCCCCCATTTAGGAN" * 3]:
        same_graph(host.HGraph.leaf(s, flags=flags), oracle.OGraph.leaf(s, flags=flags), "leaf %s/%d" % (s[:8], flags))


def walk(tree, seqs_by_name, oracle, bf, flags=0, check=None, band=False, protein=False):
    """Post-order progressive alignment with the oracle DP; both graph builders in lockstep."""
    stats = {"nodes": 0, "skips": 0, "nonreal": 0, "multi": 0, "states": set()}
    leaf_alpha, anc_alpha = host.alphabets(2 if protein else 1)
    o_leaf_alpha = oracle.protein_leaf_alphabet() if protein else oracle.DNA_ALPHABET
    char_as = 20 if protein else 4

    def rec(t):
        if t[0] == "leaf":
            s = seqs_by_name[t[1]]
            return (host.HGraph.leaf(s, leaf_alpha), oracle.OGraph.leaf(s, o_leaf_alpha),
                    min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
        hl, ol, dl = rec(t[1])
        hr, orr, dr = rec(t[2])
        if protein:
            model, pars = host.protein_model(dl + dr)
            opars = oracle.protein_model(dl + dr)[1]
        else:
            model, pars = host.dna_model(bf, dl + dr)
            opars = oracle.dna_parsimony()
        gl, gr = hl.flatten(), hr.flatten()
        b = None
        if band:
            b, _ = host.define_tunnel(hl.string(False, anc_alpha), hr.string(False, anc_alpha),
                                      hl.string(True, anc_alpha), hr.string(True, anc_alpha))
            ob, _ = oracle.define_tunnel(ol, orr, alphabet=anc_alpha)
            assert np.array_equal(b.upper, ob.upper) and np.array_equal(b.lower, ob.lower)
        res = oracle.dp_align(gl, gr, model, b)
        assert res.status == 0
        hp = host.HGraph.parent(hl, hr, res, dl, dr, pars, char_as, flags)
        op = oracle.OGraph.parent(ol, orr, res, dl, dr, opars, char_as, flags)
        stats["states"].update(hp.flatten().state.tolist())
        same_graph(hp, op, "node %d" % stats["nodes"])
        stats["nodes"] += 1
        sa = hp.attrs()[0]
        stats["skips"] += int(np.isin(sa[:, 2], (5, 6)).sum())
        stats["nonreal"] += int((sa[:, 1] == 5).sum())
        stats["multi"] += int((np.diff(hp.flatten().bwd_off) > 1).sum())
        if check:
            check(hp, res)
        d = t[3]
        return hp, op, (0.001 if d <= 0 else min(d, 0.2))
    rec(tree)
    return stats


def base_freq(seqs):
    c = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    return c / c.sum()


def test_progressive_graphs_balanced(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(16, 160, branch=0.05, sub=0.05, indel_start=0.012, mean_len=6, seed=3)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 15 and st["skips"] > 20 and st["multi"] > 20


def test_progressive_graphs_banded_and_flags(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(8, 400, branch=0.02, sub=0.015, indel_start=0.004, mean_len=5, seed=5)
    for flags in (0, 1, 2):
        st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs), flags=flags, band=True)
        assert st["nodes"] == 7


def test_progressive_graphs_caterpillar_deletes_ranges(oracle, pg):
    """Deep caterpillar: skipped-edge limits drop edges and the deletion pass leaves non_real sites."""
    names, seqs, nwk = synth.evolve_caterpillar(14, 150, seed=2)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 13 and st["nonreal"] > 0


def test_progressive_graphs_protein(oracle, pg):
    """WAG / 211-letter alphabet: upper nodes carry X and pair codes from the parsimony table."""
    aa = "ARNDCQEGHILKMFPSTWYV"
    names, seqs, nwk = synth.evolve_balanced(16, 120, branch=0.05, sub=0.08, indel_start=0.012, mean_len=4, seed=9,
                                             alphabet=aa)
    seqs[3] = seqs[3][:40] + "X" + seqs[3][41:]
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, None, protein=True)
    assert st["nodes"] == 15 and st["skips"] > 5
    assert max(st["states"]) > 20                                # pair codes reached internal nodes (X yields to a residue)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, None, protein=True, band=True)
    assert st["nodes"] == 15


def test_define_tunnel_matches_oracle_on_leaves(oracle, pg):
    for seed in range(4):
        _, seqs, _ = synth.evolve_balanced(2, 2500, branch=0.01, sub=0.01 + 0.01 * seed, indel_start=0.003, seed=seed)
        ol, orr = oracle.OGraph.leaf(seqs[0]), oracle.OGraph.leaf(seqs[1])
        want, nw = oracle.define_tunnel(ol, orr)
        got, ng = host.define_tunnel(seqs[0], seqs[1], seqs[0], seqs[1])
        assert ng == nw and np.array_equal(got.upper, want.upper) and np.array_equal(got.lower, want.lower)
        assert np.all(np.diff(got.upper) >= 0) and np.all(np.diff(got.lower) >= 0)
    # repeats and identical tails: equal suffixes across the two strings, many equal-length hits
    a = "ACGTACGTAGCTAGCTAGGATCGATCGATTTAGCGCGATATCGCGAT" * 9 + "GGGTTTCACAC"
    b = a[:150] + "TT" + a[150:300] + a[310:]
    want, nw = oracle.define_tunnel(oracle.OGraph.leaf(a), oracle.OGraph.leaf(b), min_length=12)
    got, ng = host.define_tunnel(a, b, a, b, prefix_hit_length=12)
    assert ng == nw and np.array_equal(got.upper, want.upper) and np.array_equal(got.lower, want.lower)


def test_dna_model_against_numpy(pg):
    bf = np.array([0.31, 0.19, 0.22, 0.28], np.float32)
    for dist in (0.002, 0.1, 0.4):
        model, pars = host.dna_model(bf, dist)
        pi = bf.astype(np.float64)
        ka, piR, piY = 1.0, pi[0] + pi[2], pi[1] + pi[3]
        beta = 1 / (2 * piR * piY * (1 + ka))
        aY = (piR * piY * ka - pi[0] * pi[2] - pi[1] * pi[3]) / ((2 + 2 * ka) * (piY * pi[0] * pi[2] + piR * pi[1] * pi[3]))
        aR = aY
        Q = np.array([[0, beta * pi[1], aR * pi[2] / piR + beta * pi[2], beta * pi[3]],
                      [beta * pi[0], 0, beta * pi[2], aY * pi[3] / piY + beta * pi[3]],
                      [aR * pi[0] / piR + beta * pi[0], beta * pi[1], 0, beta * pi[3]],
                      [beta * pi[0], aY * pi[1] / piY + beta * pi[1], beta * pi[2], 0]])
        Q -= np.diag(Q.sum(1))
        w, v = np.linalg.eig(Q)
        P = (v @ np.diag(np.exp(w * dist)) @ np.linalg.inv(v)).real
        lo = 0.5 * (pi[:, None] + pi[None, :]) * P / (pi[:, None] * pi[None, :])
        assert np.allclose(model.log_score[:4, :4], np.log(lo), rtol=0, atol=2e-6)
        t = 1 - np.exp(-0.5 * 0.02 * dist)
        assert np.allclose(model.params, [np.log(t), np.log(0.8), np.log(0.95), np.log(1 - 2 * t)], atol=1e-6)
        # ambiguity rows are the max over member bases; N-N is the overall max
        assert model.log_score[14, 14] == model.log_score[:4, :4].max()
        assert model.log_score[4, 0] == max(model.log_score[0, 0], model.log_score[2, 0])   # R vs A
    assert np.array_equal(pars, __import__("oracle").dna_parsimony())
