"""CPU (oracle DP behind the test seam): the output side of the tree walk -- ancestors' rows (Node::get_alignment with
include_internal_nodes, src/main/node.cpp:537-555, 779-834) and --mostcommon with Node::fix_ambiguous_states
(node.cpp:1610-1690) -- against restatements written here in plain Python over the graphs' attributes."""
import numpy as np
import pytest

from pagan2_msa_amd import host, synth

from test_workqueue_cpu import walk

DNA = "ACGTRYMKWSBDHVN"


def tree_of(msa):
    n = msa.n
    kids = {}
    for k in range(msa.n_internal):
        info = msa.node_info(k)
        kids[info.node] = (info.left, info.right)
    return n, kids


def column_rows(msa, alphabet):
    """get_alignment_column_at, recursively, for every root column: {node id: row}."""
    n, kids = tree_of(msa)
    attrs = {v: msa.node_graph(v).attrs()[0] for v in kids}
    root = 2 * n - 2
    rows = {v: [] for v in range(2 * n - 1)}
    seqs = msa._seqs

    def below(v):
        return [v] if v < n else below(kids[v][0]) + [v] + below(kids[v][1])

    def fill(v, j):
        if v < n:
            rows[v].append(seqs[v][j - 1])
            return
        a = attrs[v][j]
        l, r = kids[v]
        if a[3] >= 0:
            fill(l, a[3])
        else:
            for u in below(l):
                rows[u].append("-")
        rows[v].append("-" if a[2] in (5, 6) or a[1] == 5 else alphabet[a[0]])
        if a[4] >= 0:
            fill(r, a[4])
        else:
            for u in below(r):
                rows[u].append("-")
    for j in range(1, attrs[root].shape[0] - 1):
        fill(root, j)
    return {v: "".join(x) for v, x in rows.items()}, below(root)


def test_ancestor_rows_and_fasta_with_internal_nodes(oracle, pg, tmp_path):
    names, seqs, nwk = synth.evolve_caterpillar(9, 140, seed=6)
    msa = walk(oracle, names, seqs, nwk, use_anchors=0).align()
    msa._seqs = seqs
    rows = msa.alignment_all()
    want, order = column_rows(msa, DNA)
    assert len(rows) == 2 * len(seqs) - 1
    for v, r in enumerate(rows):
        assert r == want[v], "node %d" % v
    assert any("-" in rows[v] for v in range(len(seqs), len(rows)))
    out = tmp_path / "all.fas"
    msa.write_fasta(out, chars_by_line=70, include_internal=True)
    text = out.read_text().split(">")[1:]
    got_names = [t.split("\n", 1)[0] for t in text]
    n = len(seqs)
    assert got_names == [names[v] if v < n else "#%d#" % (v - n + 1) for v in order]      # left subtree, node, right subtree
    for t, v in zip(text, order):
        assert "".join(t.split("\n")[1:]) == rows[v]
    leaf_only = tmp_path / "leaves.fas"
    msa.write_fasta(leaf_only)
    assert [t.split("\n", 1)[0] for t in leaf_only.read_text().split(">")[1:]] == [names[v] for v in order if v < n]


def py_fix_ambiguous(states, amb, kids, attrs, n, node):
    """Node::fix_ambiguous_states / get_ambiguous_states / set_ambiguous_state on dict-of-arrays (node.cpp:1610-1690)."""
    def get(v, pos, out):
        if not amb[v][pos]:
            out.add(int(states[v][pos]))
            return
        if v < n:
            return
        a = attrs[v][pos]
        if a[3] >= 0:
            get(kids[v][0], a[3], out)
        if a[4] >= 0:
            get(kids[v][1], a[4], out)

    def put(v, pos, st):
        if not amb[v][pos]:
            return int(states[v][pos]) == st
        if v < n:
            return False
        a = attrs[v][pos]
        cont = True
        if a[3] >= 0 and put(kids[v][0], a[3], st):
            states[v][pos] = st
            cont = False
        if a[4] >= 0 and cont and put(kids[v][1], a[4], st):
            states[v][pos] = st
        return False
    for j in range(1, attrs[node].shape[0] - 1):
        a = attrs[node][j]
        ls, rs = set(), set()
        if a[3] >= 0:
            get(kids[node][0], a[3], ls)
        if a[4] >= 0:
            get(kids[node][1], a[4], rs)
        both = ls & rs
        if len(both) == 1 and len(ls) + len(rs) > 2:
            put(node, j, next(iter(both)))


def test_mostcommon_walk_matches_the_restated_chain(oracle, pg):
    """The whole --mostcommon walk again from the oracle's pieces: DP, parent graph, then Node::fix_ambiguous_states in
    Python (which rewrites states in the node AND in the ambiguous sites below it, so later DPs see them)."""
    names, seqs, nwk = synth.evolve_balanced(8, 120, branch=0.08, sub=0.12, indel_start=0.01, mean_len=3, seed=8)
    plain = walk(oracle, names, seqs, nwk, use_anchors=0).align()
    mc = walk(oracle, names, seqs, nwk, use_anchors=0, mostcommon=1).align()
    n, kids = tree_of(mc)
    bf = np.array([sum(s.count(x) for s in seqs) for x in "ACGT"], np.float32)
    bf /= bf.sum()
    og = {v: oracle.OGraph.leaf(seqs[v]) for v in range(n)}
    states, amb, attrs = {}, {}, {}
    for v in range(n):
        a = og[v].attrs()[0]
        states[v], amb[v], attrs[v] = a[:, 0].copy(), a[:, 6].copy(), a
    table = oracle.dna_parsimony()                       # the DNA most-common table is the parsimony table (model_factory.cpp:218-224)
    for k in range(mc.n_internal):
        info = mc.node_info(k)
        node, (l, r) = info.node, kids[info.node]
        model = oracle.dna_model(bf, info.dist)
        res = oracle.dp_align(og[l].flatten(), og[r].flatten(), model)
        assert res.same_alignment(mc.node_result(k)), "node %d: the product's walk took another path" % node
        og[node] = oracle.OGraph.parent(og[l], og[r], res, info.dist / 2, info.dist / 2, table, 4)
        a = og[node].attrs()[0]
        states[node], amb[node], attrs[node] = a[:, 0].copy(), a[:, 6].copy(), a
        py_fix_ambiguous(states, amb, kids, attrs, n, node)
        for v in (x for x in states if x >= n):          # write the fixed states back: the next DP reads them
            cur = og[v].attrs()[0][:, 0]
            for pos in np.nonzero(cur != states[v])[0]:
                og[v].set_state(pos, states[v][pos])
    changed = 0
    for v in range(n, 2 * n - 1):
        got = mc.node_graph(v).attrs()[0][:, 0]
        assert np.array_equal(got, states[v]), "node %d states" % v
        pa = plain.node_graph(v).attrs()[0]
        changed += int((pa[:, 0] != got).sum()) if pa.shape[0] == got.shape[0] else 1
    assert changed > 0                                    # the rule did resolve something parsimony alone leaves ambiguous
    for r, s in zip(mc.alignment(), seqs):
        assert r.replace("-", "") == s
