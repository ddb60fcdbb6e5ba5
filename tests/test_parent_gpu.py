"""SURVEY.md s.8 row f1 on the device: the parent sequence graph built by csrc/dp_parent.hip
(Basic_alignment::build_ancestral_sequence, basic_alignment.cpp:36-653) against the oracle's restatement, field by field --
sites, edges in creation order, weights and their logarithms as float bits, skip histories, bwd and fwd lists in iteration
order -- at every internal node of progressive alignments that reach the builder's rules: balanced trees (skip columns,
multi-edge sites), a caterpillar (skip limits, deleted ranges, non_real sites), homopolymer leaves (weights 0.25 / 0.9 below
the parents), banded walks under the three settings, protein states.  The paths come from the oracle's DP so that both
builders see the same columns; the walk itself (tests/test_msa_gpu.py, test_baseline_sizes_gpu.py) runs the device builder
behind the device aligner."""
import numpy as np
import pytest

from pagan2_msa_amd import host, synth
from test_host_cpu import base_freq, same_graph

pytestmark = pytest.mark.gpu


def walk(tree, seqs_by_name, oracle, bf, flags=0, band=False, protein=False, leaf_flags=0):
    """Post-order progressive alignment with the oracle DP; the device builder and the oracle's builder in lockstep (the
    children handed to the device builder are its own earlier outputs: they are resident on the device)."""
    stats = {"nodes": 0, "skips": 0, "nonreal": 0, "multi": 0, "rounds": 0, "runs": 0, "outside": 0, "patched": 0, "light": 0}
    leaf_alpha, anc_alpha = host.alphabets(2 if protein else 1)
    o_leaf_alpha = oracle.protein_leaf_alphabet() if protein else oracle.DNA_ALPHABET
    char_as = 20 if protein else 4

    def rec(t):
        if t[0] == "leaf":
            s = seqs_by_name[t[1]]
            return (host.HGraph.leaf(s, leaf_alpha, flags=leaf_flags), oracle.OGraph.leaf(s, o_leaf_alpha, flags=leaf_flags),
                    min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
        hl, ol, dl = rec(t[1])
        hr, orr, dr = rec(t[2])
        if protein:
            model, pars = host.protein_model(dl + dr)
            opars = oracle.protein_model(dl + dr)[1]
        else:
            model, pars = host.dna_model(bf, dl + dr)
            opars = oracle.dna_parsimony()
        b = None
        if band:
            b, _ = host.define_tunnel(hl.string(False, anc_alpha), hr.string(False, anc_alpha),
                                      hl.string(True, anc_alpha), hr.string(True, anc_alpha))
        res = oracle.dp_align(hl.flatten(), hr.flatten(), model, b)
        assert res.status == 0
        hp = host.HGraph.parent_device(hl, hr, res, dl, dr, pars, char_as, flags)
        op = oracle.OGraph.parent(ol, orr, res, dl, dr, opars, char_as, flags)
        same_graph(hp, op, "node %d" % stats["nodes"])
        runs, rounds, deleted, outside, patched = hp.build_info
        sa, _, _, ef = hp.attrs()
        assert deleted == int((sa[:, 1] == 5).sum())
        stats["nodes"] += 1
        stats["skips"] += int(np.isin(sa[:, 2], (5, 6)).sum())
        stats["nonreal"] += deleted
        stats["multi"] += int((np.diff(hp.flatten().bwd_off) > 1).sum())
        stats["rounds"] = max(stats["rounds"], rounds)
        stats["runs"] += runs
        stats["outside"] += outside
        stats["patched"] += patched
        stats["light"] += int((ef[:, 0] != 1.0).sum())
        d = t[3]
        return hp, op, (0.001 if d <= 0 else min(d, 0.2))
    rec(tree)
    return stats


def test_balanced_tree(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(16, 160, branch=0.05, sub=0.05, indel_start=0.012, mean_len=6, seed=3)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 15 and st["skips"] > 20 and st["multi"] > 20 and st["runs"] > 10
    assert st["light"] > 0 and st["outside"] == 0 and st["patched"] == 0      # weights < 1 came out of the table


def test_deeper_balanced_tree(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(64, 150, branch=0.03, sub=0.03, indel_start=0.02, mean_len=3, seed=21)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
    assert st["nodes"] == 63 and st["outside"] == 0


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_banded_walk_under_the_three_settings(oracle, pg, flags):
    names, seqs, nwk = synth.evolve_balanced(8, 400, branch=0.02, sub=0.015, indel_start=0.004, mean_len=5, seed=5)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs), flags=flags, band=True)
    assert st["nodes"] == 7


def test_caterpillar_deletes_ranges(oracle, pg):
    """Deep caterpillar: the skip limits drop edges and the boundary pass deletes ranges (non_real sites); later runs of
    skipped sites see lists that earlier deletions shortened, so the fixpoint needs more than one round somewhere."""
    for n, seed in ((14, 2), (18, 6)):
        names, seqs, nwk = synth.evolve_caterpillar(n, 150, seed=seed)
        st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs))
        assert st["nodes"] == n - 1 and st["nonreal"] > 0
        assert st["rounds"] >= 2


def test_homopolymer_leaves(oracle, pg):
    """--homopolymer / --454 leaves: multi-edge leaf sites with weights 0.25 and 0.9, carried into the parents."""
    names, seqs, nwk = synth.evolve_balanced(8, 200, branch=0.03, sub=0.02, indel_start=0.01, mean_len=3, seed=8)
    seqs = [s.replace("AC", "AAAC", 3).replace("GT", "GGGGGT", 2) for s in seqs]
    for leaf_flags in (1, 2):
        st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, base_freq(seqs), leaf_flags=leaf_flags)
        assert st["nodes"] == 7 and st["light"] > 0 and st["outside"] == 0


def test_protein_states(oracle, pg):
    aa = "ARNDCQEGHILKMFPSTWYV"
    names, seqs, nwk = synth.evolve_balanced(16, 120, branch=0.05, sub=0.08, indel_start=0.012, mean_len=4, seed=9, alphabet=aa)
    st = walk(synth.parse_newick(nwk), dict(zip(names, seqs)), oracle, None, protein=True)
    assert st["nodes"] == 15


def test_the_walk_builds_its_long_parents_on_the_device(oracle, pg, monkeypatch):
    """PAGAN_PARENTS=device: every parent of a tree walk comes from the device builder, children resident from the level
    below; the alignment and every node's graph equal the host builder's walk."""
    names, seqs, nwk = synth.evolve_balanced(8, 1500, branch=0.02, sub=0.02, indel_start=0.004, mean_len=4, seed=31)
    monkeypatch.setenv("PAGAN_PARENTS", "host")
    a = host.Msa(names, seqs, nwk, use_anchors=1).align()
    before = host._lib().pagan_parents_device_calls()
    monkeypatch.setenv("PAGAN_PARENTS", "device")
    b = host.Msa(names, seqs, nwk, use_anchors=1).align()
    assert host._lib().pagan_parents_device_calls() - before == 7
    assert a.alignment() == b.alignment()
    n = len(names)
    for node in range(n, 2 * n - 1):
        same_graph(b.node_graph(node), a.node_graph(node), "node %d" % node)


def test_mostcommon_walk_with_every_parent_from_the_device(oracle, pg, monkeypatch):
    """--mostcommon (Node::fix_ambiguous_states, node.cpp:1610-1690): the parents come from the device builder as in any
    walk, the state rewriting runs on the host behind each build and the node's device copy takes the new states over (the
    build one level up reads its children's states there).  Every node's graph and the alignment equal the host builder's
    --mostcommon walk -- and the rule did change states, or the test would prove nothing."""
    names, seqs, nwk = synth.evolve_balanced(16, 400, branch=0.08, sub=0.12, indel_start=0.008, mean_len=3, seed=41)
    monkeypatch.setenv("PAGAN_PARENTS", "host")
    plain = host.Msa(names, seqs, nwk, use_anchors=0).align()
    a = host.Msa(names, seqs, nwk, use_anchors=0, mostcommon=1).align()
    before = host._lib().pagan_parents_device_calls()
    monkeypatch.setenv("PAGAN_PARENTS", "device")
    b = host.Msa(names, seqs, nwk, use_anchors=0, mostcommon=1).align()
    assert host._lib().pagan_parents_device_calls() - before == 15
    assert a.alignment() == b.alignment()
    n = len(names)
    changed = 0
    for node in range(n, 2 * n - 1):
        same_graph(b.node_graph(node), a.node_graph(node), "node %d" % node)
        pa, pb = plain.node_graph(node).attrs()[0], b.node_graph(node).attrs()[0]
        changed += int((pa[:, 0] != pb[:, 0]).sum()) if pa.shape == pb.shape else 1
    assert changed > 0
