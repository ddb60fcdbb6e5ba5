"""CPU: the parent-graph builder read a second time (tests/pycheck_parent.py) against the oracle's restatement
(oracle/oracle_host.cpp, struct Builder), field by field -- sites (state, type, path state, children, skip counts and
distances, ambiguity), edges (ends, weights and their logs as float bits, skip counts, distances, whether still linked) and
the bwd / fwd lists in their iteration order -- at every internal node of progressive alignments: balanced and 64-leaf
trees, a caterpillar (the deletion pass leaves non_real sites), banded alignments under the three settings, protein,
leaves with homopolymer edges and the reads settings.  The children the Python builder gets are imported from the oracle's
dump of its own graphs; alignment paths come from the oracle's DP."""
import numpy as np
import pytest

import pycheck_parent as pp
from pagan2_msa_amd import host, synth
from test_host_cpu import base_freq


def same_dump(py_seq, og, what):
    (state, off, src, lw, eid), (sa, sd, ea, ef), (fo, fe) = py_seq.dump()
    flat = og.flatten()
    assert np.array_equal(state, flat.state), what + ": states"
    assert np.array_equal(off, flat.bwd_off) and np.array_equal(src, flat.bwd_src) and np.array_equal(eid, flat.bwd_eid), what + ": bwd lists"
    assert lw.tobytes() == flat.bwd_logw.tobytes(), what + ": log weights (bits)"
    osa, osd, oea, oef = og.attrs()
    assert np.array_equal(sa, osa), what + ": site attributes"
    assert sd.tobytes() == osd.tobytes(), what + ": site distances (bits)"
    assert np.array_equal(ea, oea), what + ": edge attributes"
    assert ef.tobytes() == oef.tobytes(), what + ": edge weights / distances (bits)"
    ofo, ofe = og.fwd()
    assert np.array_equal(fo, ofo) and np.array_equal(fe, ofe), what + ": fwd lists"


def walk(tree, seqs_by_name, oracle, bf, flags=0, leaf_flags=0, band=False, protein=False):
    """Post-order progressive alignment on the oracle's graphs and DP; at every node the Python builder works on an import
    of the oracle's children and its parent is compared with the oracle's."""
    stats = {"nodes": 0, "nonreal": 0, "skips": 0, "multi": 0, "edges_unlinked": 0}
    _leaf_alpha, anc_alpha = host.alphabets(2 if protein else 1)
    o_leaf_alpha = oracle.protein_leaf_alphabet() if protein else oracle.DNA_ALPHABET
    char_as = 20 if protein else 4

    def rec(t):
        if t[0] == "leaf":
            return oracle.OGraph.leaf(seqs_by_name[t[1]], o_leaf_alpha, leaf_flags), (min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
        ol, dl = rec(t[1])
        orr, dr = rec(t[2])
        if protein:
            model, _ = host.protein_model(dl + dr)
            opars = oracle.protein_model(dl + dr)[1]
        else:
            model, _ = host.dna_model(bf, dl + dr)
            opars = oracle.dna_parsimony()
        b = oracle.define_tunnel(ol, orr, alphabet=anc_alpha)[0] if band else None
        res = oracle.dp_align(ol.flatten(), orr.flatten(), model, b)
        assert res.status == 0
        # the Python side's children: imported BEFORE the oracle marks the used edges (it marks its own)
        pl = pp.Sequence.from_dump(ol.flatten(), ol.attrs(), ol.fwd())
        pr = pp.Sequence.from_dump(orr.flatten(), orr.attrs(), orr.fwd())
        same_dump(pl, ol, "import of the left child")                # (the import itself loses nothing)
        py_parent = pp.build_parent(pl, pr, res.cols, res.left_used, res.right_used, dl, dr, opars, char_as, flags)
        op = oracle.OGraph.parent(ol, orr, res, dl, dr, opars, char_as, flags)
        same_dump(py_parent, op, "node %d" % stats["nodes"])
        sa, _sd, ea, _ef = op.attrs()
        stats["nodes"] += 1
        stats["nonreal"] += int((sa[:, 1] == pp.non_real).sum())
        stats["skips"] += int(np.isin(sa[:, 2], (pp.xskipped, pp.yskipped)).sum())
        stats["multi"] += int((np.diff(op.flatten().bwd_off) > 1).sum())
        stats["edges_unlinked"] += int((ea[:, 5] == 0).sum())
        d = t[3]
        return op, (0.001 if d <= 0 else min(d, 0.2))
    rec(tree)
    return stats


def test_balanced_tree(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(16, 160, branch=0.05, sub=0.05, indel_start=0.012, mean_len=6, seed=3)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 15 and st["skips"] > 20 and st["multi"] > 20


def test_sixty_four_leaves(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(64, 120, branch=0.04, sub=0.04, indel_start=0.015, mean_len=4, seed=11)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 63 and st["multi"] > 100


def test_caterpillar_deletes_ranges(oracle, pg):
    """Deep caterpillar: the skip limits drop edges, the deletion pass unlinks edges and leaves non_real sites."""
    names, seqs, nwk = synth.evolve_caterpillar(14, 150, seed=2)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 13 and st["nonreal"] > 0 and st["edges_unlinked"] > 0


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_banded_alignments_under_the_settings(oracle, pg, flags):
    """flags: 1 the reads settings (no skip limits, no penalty), 2 --no-reduced-terminal-penalties"""
    names, seqs, nwk = synth.evolve_balanced(8, 400, branch=0.02, sub=0.015, indel_start=0.004, mean_len=5, seed=5)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs), flags=flags, band=True)
    assert st["nodes"] == 7


def test_protein(oracle, pg):
    aa = "ARNDCQEGHILKMFPSTWYV"
    names, seqs, nwk = synth.evolve_balanced(16, 120, branch=0.05, sub=0.08, indel_start=0.012, mean_len=4, seed=9, alphabet=aa)
    seqs[3] = seqs[3][:40] + "X" + seqs[3][41:]
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, None, protein=True)
    assert st["nodes"] == 15 and st["skips"] > 5


@pytest.mark.parametrize("leaf_flags,flags", [(2, 0), (1, 1)])
def test_homopolymer_leaves(oracle, pg, leaf_flags, flags):
    """leaves with skip-back edges over homopolymer runs (multi-edge sites from the first level on)"""
    names, seqs, nwk = synth.evolve_balanced(8, 300, branch=0.02, sub=0.02, indel_start=0.006, seed=23)
    seqs = [s.replace("AC", "AAAAC", 25).replace("GT", "GGGGT", 15) for s in seqs]
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs), flags=flags, leaf_flags=leaf_flags)
    assert st["nodes"] == 7 and st["multi"] > 10
