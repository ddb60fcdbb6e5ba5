"""Deterministic synthetic inputs (the reference ships no data).

sequences down a balanced binary tree, random sequence graphs with multi-edge sites, and
small model tables.  Used by tests/ and bench.py to build identical inputs for the HIP path
and the oracle.
"""
import math

import numpy as np

from .abi import Graph, Model

DNA = "ACGT"
DNA_FULL = "ACGTRYMKWSBDHVN"


def evolve_balanced(n_leaves, length, branch=0.05, sub=0.04, indel_start=0.004, mean_len=4.0, seed=0,
                    alphabet=DNA):
    """Root iid uniform over `alphabet`, evolved down a balanced binary tree; per site per
    branch: substitution w.p. `sub`, deletion start w.p. `indel_start` (geometric length,
    mean `mean_len`), insertion start w.p. `indel_start` (same law, iid residues).
    Returns (names, sequences, newick) with the true tree as guide tree."""
    rng = np.random.default_rng(seed)
    A = len(alphabet)
    depth = int(round(math.log2(n_leaves)))
    assert 2 ** depth == n_leaves
    p_geo = 1.0 / mean_len

    def mutate(seq):
        n = seq.shape[0]
        out = seq.copy()
        subs = rng.random(n) < sub
        k = int(subs.sum())
        if k:
            out[subs] = (out[subs] + rng.integers(1, A, k)) % A
        keep = np.ones(n, bool)
        for s in np.nonzero(rng.random(n) < indel_start)[0]:
            keep[s:s + int(rng.geometric(p_geo))] = False
        ins_at = np.nonzero(rng.random(n) < indel_start)[0]
        pieces, prev = [], 0
        for s in ins_at:
            pieces.append(out[prev:s][keep[prev:s]])
            pieces.append(rng.integers(0, A, int(rng.geometric(p_geo))).astype(out.dtype))
            prev = s
        pieces.append(out[prev:][keep[prev:]])
        return np.concatenate(pieces)

    level = [rng.integers(0, A, length).astype(np.int8)]
    for _ in range(depth):
        nxt = []
        for s in level:
            nxt.append(mutate(s))
            nxt.append(mutate(s))
        level = nxt
    names = ["S%03d" % k for k in range(n_leaves)]
    seqs = ["".join(alphabet[c] for c in s) for s in level]
    nodes = ["%s:%g" % (nm, branch) for nm in names]
    while len(nodes) > 1:
        nodes = ["(%s,%s):%g" % (nodes[k], nodes[k + 1], branch) for k in range(0, len(nodes), 2)]
    newick = nodes[0].rsplit(":", 1)[0] + ";"
    return names, seqs, newick


def _mutator(rng, A, sub, indel_start, mean_len):
    p_geo = 1.0 / mean_len

    def mutate(seq):
        n = seq.shape[0]
        out = seq.copy()
        subs = rng.random(n) < sub
        k = int(subs.sum())
        if k:
            out[subs] = (out[subs] + rng.integers(1, A, k)) % A
        keep = np.ones(n, bool)
        for s in np.nonzero(rng.random(n) < indel_start)[0]:
            keep[s:s + int(rng.geometric(p_geo))] = False
        ins_at = np.nonzero(rng.random(n) < indel_start)[0]
        pieces, prev = [], 0
        for s in ins_at:
            pieces.append(out[prev:s][keep[prev:s]])
            pieces.append(rng.integers(0, A, int(rng.geometric(p_geo))).astype(out.dtype))
            prev = s
        pieces.append(out[prev:][keep[prev:]])
        return np.concatenate(pieces)
    return mutate


def evolve_caterpillar(n_leaves, length, branch=0.03, sub=0.03, indel_start=0.02, mean_len=3.0, seed=0,
                       alphabet=DNA):
    """Same substitution/indel process down a caterpillar tree (((S0,S1),S2),S3)...: the deep
    pair's private insertions stay skipped for many levels, which is what drives the
    reference's skipped-edge limits (basic_alignment.cpp:370-489,587-592)."""
    rng = np.random.default_rng(seed)
    mutate = _mutator(rng, len(alphabet), sub, indel_start, mean_len)
    anc = rng.integers(0, len(alphabet), length).astype(np.int8)
    leaves = [None] * n_leaves
    for k in range(n_leaves - 1, 1, -1):
        leaves[k] = mutate(anc)
        anc = mutate(anc)
    leaves[0], leaves[1] = mutate(anc), mutate(anc)
    names = ["S%03d" % k for k in range(n_leaves)]
    seqs = ["".join(alphabet[c] for c in s) for s in leaves]
    tree = "(%s:%g,%s:%g)" % (names[0], branch, names[1], branch)
    for k in range(2, n_leaves):
        tree = "(%s:%g,%s:%g)" % (tree, branch, names[k], branch)
    return names, seqs, tree + ";"


def parse_newick(newick):
    """Tiny rooted-binary Newick reader for tests: returns nested (left, right, dist) /
    (name, dist) tuples in the order the guide tree is walked."""
    pos = [0]
    s = newick.strip().rstrip(";")

    def node():
        if s[pos[0]] == "(":
            pos[0] += 1
            left = node()
            assert s[pos[0]] == ","
            pos[0] += 1
            right = node()
            assert s[pos[0]] == ")"
            pos[0] += 1
            return ("internal", left, right, dist())
        j = pos[0]
        while pos[0] < len(s) and s[pos[0]] not in ":,()":
            pos[0] += 1
        return ("leaf", s[j:pos[0]], dist())

    def dist():
        if pos[0] < len(s) and s[pos[0]] == ":":
            j = pos[0] + 1
            pos[0] = j
            while pos[0] < len(s) and s[pos[0]] not in ",()":
                pos[0] += 1
            return float(s[j:pos[0]])
        return 0.0
    return node()


def chain_graph(seq, alphabet=DNA_FULL):
    """Plain leaf graph of a sequence string (Sequence::create_default_sequence, no 454 edges)."""
    return Graph.chain(np.array([alphabet.index(c) for c in seq], np.int32))


def random_graph(n_real, n_states, seed, p_extra=0.3, max_deg=4, max_span=6, p_dead=0.0):
    """Random sequence graph: every site keeps the edge from its predecessor (unless `dead`),
    a fraction get extra bwd edges of random span in random list order with random
    log-weights -- exercises multi-edge ordering, long edges and predecessor-less sites
    (what delete_edge_range leaves behind, basic_alignment.cpp:491-508)."""
    rng = np.random.default_rng(seed)
    n = n_real + 2
    state = np.full(n, -1, np.int32)
    state[1:-1] = rng.integers(0, n_states, n_real)
    off, src, lw, eid = [0, 0], [], [], []
    next_eid = 1
    for s in range(1, n):
        srcs = []
        dead = (1 < s < n - 1) and rng.random() < p_dead
        if not dead:
            srcs.append(s - 1)
            if rng.random() < p_extra:
                k = int(rng.integers(1, max_deg))
                cand = [s - d for d in range(2, max_span + 2) if s - d >= 0]
                rng.shuffle(cand)
                srcs.extend(cand[:k])
                rng.shuffle(srcs)
        for p in srcs:
            src.append(p)
            w = 1.0 if rng.random() < 0.5 else float(rng.choice([0.9, 0.81, 0.25, 0.729]))
            lw.append(np.log(np.float32(w)))
            eid.append(next_eid)
            next_eid += 1
        off.append(len(src))
    return Graph(state, np.array(off, np.int32), np.array(src, np.int32), np.array(lw, np.float32),
                 np.array(eid, np.int32), n_edges=next_eid)


def random_model(n_states, seed, dist=0.1):
    """A log-odds-like table with many exact ties (values on a coarse float grid) plus the
    indel parameters of Model_factory::alignment_model (model_factory.cpp:1898-1925)."""
    rng = np.random.default_rng(seed)
    t = (rng.integers(-40, 8, (n_states, n_states)) / 8.0).astype(np.float32)
    t = np.minimum(t, t.T)
    t[np.arange(n_states), np.arange(n_states)] = (rng.integers(4, 16, n_states) / 8.0).astype(np.float32)
    return Model(t, *indel_params(dist))


def indel_params(dist, ins_rate=0.01, del_rate=0.01, ext=0.8, end_ext=0.95):
    """log_id_prob, log_ext_prob, log_end_ext_prob, log_match_prob as floats
    (model_factory.cpp:1898-1925; DNA defaults model_factory.cpp:1303-1306)."""
    t = 1.0 - math.exp(-0.5 * (np.float32(ins_rate) + np.float32(del_rate)) * dist)
    return (np.float32(math.log(t)), np.log(np.float32(ext)), np.log(np.float32(end_ext)),
            np.float32(math.log(1.0 - 2 * t)))


def jc_like_dna_model(dist=0.1, n_states=15):
    """15-state DNA table built the way alignment_model does (log-odds of a symmetric
    substitution process, ambiguity rows = max over members, model_factory.cpp:1944-2016).
    An INPUT for parity tests, not a restatement of the reference's eigen path."""
    sets = ["A", "C", "G", "T", "AG", "CT", "AC", "GT", "AT", "CG", "CGT", "AGT", "ACT", "ACG", "ACGT"]
    p_same = 0.25 + 0.75 * math.exp(-4.0 * dist / 3.0)
    p_diff = 0.25 - 0.25 * math.exp(-4.0 * dist / 3.0)
    core = np.full((4, 4), p_diff)
    np.fill_diagonal(core, p_same)
    pr = np.zeros((n_states, n_states), np.float64)
    lo = (0.5 * (0.25 + 0.25) * core / (0.25 * 0.25)).astype(np.float32)
    pr[:4, :4] = lo
    logpr = np.zeros((n_states, n_states), np.float64)
    logpr[:4, :4] = np.log(lo)                      # float log of a float
    for i in range(n_states):
        for j in range(n_states):
            if i < 4 and j < 4:
                continue
            mx = max(pr[DNA.index(a), DNA.index(b)] for a in sets[i] for b in sets[j])
            pr[i, j] = mx
            logpr[i, j] = math.log(mx)
    return Model(logpr.astype(np.float32), *indel_params(dist))
