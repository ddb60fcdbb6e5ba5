"""CPU tests of the codon side: the empirical codon model's producer (Model_factory::define_codon_alphabet / codon_model /
alignment_model, 61 sense codons + NNN + 1830 pair codes = 1892 states) -- the product's restatement
(csrc/host_model.cpp) against the oracle's literal one (oracle/oracle_model.cpp) bit for bit and both against
numpy/scipy and rules written out here --, codon leaves (Sequence::create_codon_sequence), parent graphs over the 1892-state
parsimony table, and the tree walk on DNA read as codons (data_type 3) against the same walk made from the oracle's pieces."""
import os
import re

import numpy as np
import pytest
from scipy.linalg import expm

from pagan2_msa_amd import abi, host, synth

from test_host_cpu import same_graph
from test_workqueue_cpu import walk as msa_walk

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
S = 1892
# the model's codon order (model_factory.cpp:841): every triplet over ACGT in lexical order without TAA, TAG, TGA
CODONS = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in ("TAA", "TAG", "TGA")]


def codon_constants():
    src = open(os.path.join(ROOT, "oracle", "codon_data.h")).read()
    nums = [float(x) for x in re.findall(r"-?\d+\.\d+", src.split("kCodonPi[61]")[1])]
    return np.array(nums[:61]), np.array(nums[61:61 + 3721]).reshape(61, 61)


def pair_codes():
    code, members = {}, {}
    k = 62
    for i in range(60):
        for j in range(i + 1, 61):
            code[(i, j)] = code[(j, i)] = k
            members[k] = (i, j)
            k += 1
    assert k == S
    return code, members


def test_constants_are_a_rate_matrix_as_tabulated():
    """Six printed decimals: rows sum to zero and detailed balance holds to the printing precision, not exactly -- the
    eigen solution symmetrises from the lower triangle (eigen.cpp:66-71), which both restatements do alike."""
    pi, Q = codon_constants()
    assert abs(pi.sum() - 1) < 5e-6
    assert np.abs(Q.sum(1)).max() < 2e-5
    assert np.abs(pi[:, None] * Q - (pi[:, None] * Q).T).max() < 1e-6
    assert (Q - np.diag(np.diag(Q)) >= 0).all() and (np.diag(Q) < 0).all()
    a = open(os.path.join(ROOT, "oracle", "codon_data.h")).read().split("namespace")[1].split("{", 1)[1]
    b = open(os.path.join(ROOT, "pagan2-msa_amd", "csrc", "codon_data.h")).read().split("namespace")[1].split("{", 1)[1]
    assert a.rsplit("}", 1)[0] == b.rsplit("}", 1)[0]


def test_eigen_solution_of_the_codon_matrix(oracle, pg):
    pi, Q = codon_constants()
    r1, U1, V1 = host.eigen_qrev(Q, pi)
    r2, U2, V2 = oracle.eigen_qrev(Q, pi)
    assert r1.tobytes() == r2.tobytes() and U1.tobytes() == U2.tobytes() and V1.tobytes() == V2.tobytes()
    assert r1[0] == 0 and np.all(np.diff(r1) <= 0)
    assert np.abs(U1 @ V1 - np.eye(61)).max() < 1e-12
    assert np.abs(U1 @ np.diag(r1) @ V1 - Q).max() < 1e-4          # Q as tabulated is reversible to ~1e-6 of pi*Q only


@pytest.mark.parametrize("dist", [0.002, 0.05, 0.1, 0.4, 1.3])
def test_codon_model_bits_match_oracle_and_scipy(oracle, pg, dist):
    m1, p1 = host.codon_model(dist)
    m2, p2 = oracle.codon_model(dist)
    assert m1.n_states == S
    assert m1.table.tobytes() == m2.table.tobytes(), "1892 x 1892 log-odds tables differ in some bit"
    assert np.array(m1.params).tobytes() == np.array(m2.params).tobytes()
    assert np.array_equal(p1, p2)
    pi, Q = codon_constants()
    T = m1.log_score

    def log_odds(rates):
        P = expm(rates * dist)
        return np.log(0.5 * (pi[:, None] + pi[None, :]) * P / (pi[:, None] * pi[None, :]))
    # the tabulated Q is reversible to its six decimals only, and the eigen solution reads its lower triangle
    # (eigen.cpp:66-71): against that matrix the table agrees to 1e-3 in the logs (of probabilities down to 1e-7),
    # against Q as printed to 1.5e-2
    sq = np.sqrt(pi)
    low = np.tril(Q, -1) * sq[:, None] / sq[None, :]
    sym = low + low.T + np.diag(np.diag(Q))
    assert np.allclose(T[:61, :61], log_odds(sym * sq[None, :] / sq[:, None]), rtol=0, atol=1e-3)
    assert np.allclose(T[:61, :61], log_odds(Q), rtol=0, atol=1.5e-2)
    t = 1 - np.exp(-0.5 * 0.02 * dist)                               # ins = del = 0.01, gap extension 0.5 / 0.75 at the ends
    assert np.allclose(m1.params, [np.log(t), np.log(0.5), np.log(0.75), np.log(1 - 2 * t)], atol=1e-5)
    # NNN = best over all codons; a pair code = best of its members (model_factory.cpp:2026-2090)
    code, _ = pair_codes()
    assert T[61, 3] == T[:61, 3].max() and T[5, 61] == T[5, :61].max() and T[61, 61] == T[:61, :61].max()
    c = code[(2, 47)]
    assert T[c, 4] == max(T[2, 4], T[47, 4]) and T[4, c] == max(T[4, 2], T[4, 47])
    c2 = code[(0, 60)]
    assert T[c, c2] == max(T[2, 0], T[2, 60], T[47, 0], T[47, 60])
    assert T[61, c] == T[:61, c].max() and T[c, 61] == T[c, :61].max()
    assert np.isfinite(T).all()
    # the probability-space view the forward/backward pass takes
    q1, q2 = host.model_prob(3, dist), oracle.model_prob(3, dist)
    assert q1.table.tobytes() == q2.table.tobytes()
    assert (q1.gap_open, q1.gap_ext, q1.non_gap) == (q2.gap_open, q2.gap_ext, q2.non_gap)


def test_codon_parsimony_table_rules(pg):
    _, pars = host.codon_model(0.1)
    P = pars.reshape(S, S).T                      # P[i, j] = table(i, j)
    _, Q = codon_constants()
    code, members = pair_codes()
    d = np.arange(S)
    assert np.array_equal(P[d, d], d) and np.array_equal(P[61, :], d) and np.array_equal(P[:, 61], d)   # NNN yields to anything
    for i in range(61):
        for j in range(61):
            if i != j:
                assert P[i, j] == code[(i, j)]                            # two codons -> their pair code
    for c, (a, b) in members.items():
        assert P[a, c] == a and P[c, a] == a and P[b, c] == b and P[c, b] == b   # codon inside a pair -> the codon
    # disjoint and overlapping pairs: the member pair with the largest rate; the running maximum is a float that starts at
    # the first pair's rate and only a strictly larger rate replaces it (:1044-1088)
    rng = np.random.default_rng(5)
    cases = [(3, code[(5, 9)]), (code[(0, 1)], code[(2, 3)]), (code[(4, 7)], code[(7, 11)]), (code[(10, 12)], 6),
             (code[(59, 60)], code[(0, 60)])]
    cases += [(int(a), int(b)) for a, b in rng.integers(62, S, size=(3000, 2))]
    cases += [(int(a), int(b)) for a, b in zip(rng.integers(0, 61, 500), rng.integers(62, S, 500))]
    for (i, j) in cases:
        mi, mj = members.get(i, (i,)), members.get(j, (j,))
        if i == j or (len(mi) == 1 and mi[0] in mj) or (len(mj) == 1 and mj[0] in mi):
            continue
        order = [(mi[0], mj[0])] + ([(mi[0], mj[1])] if len(mj) == 2 else []) + ([(mi[1], mj[0])] if len(mi) == 2 else []) \
            + ([(mi[1], mj[1])] if len(mi) == 2 and len(mj) == 2 else [])
        best, pick = np.float32(Q[order[0]]), order[0]
        for (a, b) in order[1:]:
            if Q[a, b] > best:
                best, pick = np.float32(Q[a, b]), (a, b)
        assert pick[0] != pick[1] and P[i, j] == code[pick], (i, j)


def test_codon_names_states_and_mostcommon(pg, oracle):
    names, mc = host.codon_alphabet()
    onames, omc = oracle.codon_alphabet()
    assert len(names) == 3 * S and [names[3 * k:3 * k + 3] for k in range(S)] == onames
    assert np.array_equal(mc, omc)
    assert [names[3 * k:3 * k + 3] for k in range(61)] == CODONS and names[183:186] == "NNN"
    iupac = {frozenset("A"): "A", frozenset("C"): "C", frozenset("G"): "G", frozenset("T"): "T", frozenset("AC"): "M",
             frozenset("AG"): "R", frozenset("CG"): "S", frozenset("AT"): "W", frozenset("CT"): "Y", frozenset("GT"): "K"}
    _, members = pair_codes()
    for c, (a, b) in members.items():
        assert names[3 * c:3 * c + 3] == "".join(iupac[frozenset((CODONS[a][p], CODONS[b][p]))] for p in range(3))
    pi, _ = codon_constants()
    M = mc.reshape(61, 61).T
    for i in range(61):
        for j in range(61):
            assert M[i, j] == (i if pi[i] > pi[j] else j)
    # leaf states: one per triplet; stop codons, ambiguity letters and a last partial triplet are NNN
    assert host.codon_states("AAAAACTAGTTTGG").tolist() == [0, 1, 61, 60, 61]
    assert host.codon_states("").tolist() == [] and host.codon_states("AC").tolist() == [61]
    assert host.codon_states("".join(CODONS)).tolist() == list(range(61))
    assert host.codon_states("TAATGAANAATGRYA").tolist() == [61, 61, 61, CODONS.index("ATG"), 61]
    rng = np.random.default_rng(2)
    for _ in range(300):
        s = "".join(rng.choice(list("ACGTN"), p=[0.24, 0.24, 0.24, 0.24, 0.04], size=int(rng.integers(0, 60))))
        assert np.array_equal(host.codon_states(s), oracle.codon_states(s)), s


# ---- graphs ---------------------------------------------------------------------------------------------------------
PSEUDO = "".join(chr(64 + k) for k in range(62))      # one letter per leaf state, for the oracle's default leaf builder


def oracle_codon_leaf(oracle, nucleotides):
    """create_codon_sequence builds the plain chain create_default_sequence builds (sequence.cpp:306-359 vs 152-303
    without the 454 / homopolymer modes); the states come from the oracle's restatement of the triplet lookup."""
    st = oracle.codon_states(nucleotides)
    return oracle.OGraph.leaf("".join(PSEUDO[k] for k in st), PSEUDO, 0)


def evolve_codons(n, length, seed, shape="balanced", **kw):
    """Sequences evolved over a 61-letter alphabet, written out as codons; a few stop codons / N put in afterwards."""
    letters = "".join(chr(64 + k) for k in range(61))
    if shape == "balanced":
        names, seqs, nwk = synth.evolve_balanced(n, length, seed=seed, alphabet=letters, **kw)
    else:
        names, seqs, nwk = synth.evolve_caterpillar(n, length, seed=seed, alphabet=letters, **kw)
    return names, ["".join(CODONS[ord(c) - 64] for c in s) for s in seqs], nwk


def test_codon_leaf_graphs(oracle, pg):
    for s in ["ATGGCC", "ATG", "ATGAAAAAAAAATAGGCNCCCTT", "".join(CODONS) * 2 + "A"]:
        h = host.HGraph.codon_leaf(s)
        same_graph(h, oracle_codon_leaf(oracle, s), "codon leaf " + s[:9])
        names, _ = host.codon_alphabet()
        want = "".join(s[i:i + 3] if s[i:i + 3] in CODONS else "NNN" for i in range(0, len(s), 3))
        assert h.string(False, names) == want and h.string(True, names) == want


def test_progressive_codon_graphs(oracle, pg):
    """Parent graphs over the 1892-state parsimony table, both builders in lockstep on the oracle's alignments; the
    ancestors' strings print three letters per site ('---' where a site is skipped)."""
    names, seqs, nwk = evolve_codons(8, 60, seed=4, branch=0.08, sub=0.12, indel_start=0.02, mean_len=3)
    seqs[2] = seqs[2][:30] + "TAG" + seqs[2][33:]
    seqs[5] = seqs[5][:9] + "ANN" + seqs[5][12:] + "G"
    by_name = dict(zip(names, seqs))
    anc, _ = host.codon_alphabet()
    stats = {"nodes": 0, "states": set(), "gaps": 0}

    def rec(t):
        if t[0] == "leaf":
            s = by_name[t[1]]
            return host.HGraph.codon_leaf(s), oracle_codon_leaf(oracle, s), (min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
        hl, ol, dl = rec(t[1])
        hr, orr, dr = rec(t[2])
        model, pars = host.codon_model(dl + dr)
        opars = oracle.codon_model(dl + dr)[1]
        res = oracle.dp_align(hl.flatten(), hr.flatten(), model, None)
        assert res.status == 0
        hp = host.HGraph.parent(hl, hr, res, dl, dr, pars, 61, 0)
        op = oracle.OGraph.parent(ol, orr, res, dl, dr, opars, 61, 0)
        same_graph(hp, op, "node %d" % stats["nodes"])
        sa = hp.attrs()[0]
        want = "".join("---" if a[2] in (5, 6) or a[1] == 5 else anc[3 * a[0]:3 * a[0] + 3] for a in sa[1:-1])
        assert hp.string(True, anc) == want and hp.string(False, anc) == want.replace("---", "")
        stats["gaps"] += want.count("---")
        stats["nodes"] += 1
        stats["states"].update(hp.flatten().state.tolist())
        d = t[3]
        return hp, op, (0.001 if d <= 0 else min(d, 0.2))
    rec(synth.parse_newick(nwk))
    assert stats["nodes"] == 7 and max(stats["states"]) > 61 and stats["gaps"] > 0


def codon_strings(oracle, og, names):
    """Sequence::get_sequence_string of a codon graph, without and with gaps (sequence.cpp:704-740), from the oracle graph's
    attributes: three letters per site, '---' where a site is skipped or was deleted."""
    sa = og.attrs()[0][1:-1]
    parts = [None if (a[2] in (5, 6) or a[1] == 5) else names[a[0]] for a in sa]
    return "".join(p for p in parts if p), "".join(p or "---" for p in parts)


def test_translation_of_codon_strings(oracle, pg):
    """Codon_translation::gapped_DNA_to_protein: every triplet over the IUPAC letters and '-', the table's duplicate (CTR is
    listed under V before L, and a map insert keeps the first), stop codons, partial triplets."""
    letters = "ACGTRYMKWSBDHVN-"
    for a in letters:
        for b in letters:
            s = "".join(a + b + c for c in letters)
            assert host.codon_translate(s) == oracle.codon_translate(s), s
    assert host.codon_translate("ATGCTRGTRTAA---NNNAAMTTYCCNAC") == "MVXX-XXFPX"
    assert host.codon_translate("") == "" and host.codon_translate("A") == "X"
    assert host.codon_translate("".join(CODONS)) == oracle.codon_translate("".join(CODONS))
    assert set(host.codon_translate("".join(CODONS))) == set("ARNDCQEGHILKMFPSTWYV")


def oracle_codon_walk(oracle, names, seqs, nwk, anchors=False, hit_length=6, trim=5, offset=4):
    """The codon walk made from the oracle's pieces: leaves by triplet, model per node, anchors found in the translated
    codon strings (Viterbi_alignment::define_tunnel for codon data), DP, parent graphs.  -> (results, root graph, how many
    nodes got a band narrower than 80 % of their matrix)."""
    by_name = dict(zip(names, seqs))
    cnames, _ = oracle.codon_alphabet()
    results, banded = [], [0]
    i32p = oracle.C.POINTER(oracle.C.c_int32)

    def rec(t):
        if t[0] == "leaf":
            return oracle_codon_leaf(oracle, by_name[t[1]]), (min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
        ol, dl = rec(t[1])
        orr, dr = rec(t[2])
        model, opars = oracle.codon_model(dl + dr)
        band = None
        if anchors:
            (s1, g1), (s2, g2) = codon_strings(oracle, ol, cnames), codon_strings(oracle, orr, cnames)
            p1, p2, q1, q2 = (oracle.codon_translate(x) for x in (s1, s2, g1, g2))
            up, lo = np.zeros(len(q1) + 1, np.int32), np.zeros(len(q1) + 1, np.int32)
            oracle.lib().oracle_define_tunnel(p1.encode(), p2.encode(), q1.encode(), q2.encode(), hit_length, trim, offset,
                                              up.ctypes.data_as(i32p), lo.ctypes.data_as(i32p))
            band = abi.Band(up, lo)
            banded[0] += int((lo - up).sum() < 0.8 * len(q1) * len(q2))
        res = oracle.dp_align(ol.flatten(), orr.flatten(), model, band)
        results.append(res)
        d = t[3]
        return oracle.OGraph.parent(ol, orr, res, dl, dr, opars, 61, 0), (0.001 if d <= 0 else min(d, 0.2))
    root, _ = rec(synth.parse_newick(nwk))
    return results, root, banded[0]


@pytest.mark.parametrize("shape,anchors", [("balanced", 0), ("balanced", 1), ("caterpillar", 1)])
def test_tree_walk_on_codons(oracle, pg, shape, anchors):
    """data_type 3: the whole walk (model per node, leaves by triplet, anchors found in the translation of the codon
    strings, DP behind the test seam, parents, rows of three characters per column) against the same walk made here from
    the oracle's pieces."""
    if shape == "balanced":
        names, seqs, nwk = evolve_codons(8, 120, seed=7, branch=0.03, sub=0.04, indel_start=0.012, mean_len=2)
    else:
        names, seqs, nwk = evolve_codons(6, 100, seed=3, shape="caterpillar", branch=0.02, sub=0.02, indel_start=0.01)
    seqs[1] = seqs[1][:12] + "TGA" + seqs[1][15:]
    msa = msa_walk(oracle, names, seqs, nwk, data_type=3, use_anchors=anchors, prefix_hit_length=6, anchors_offset=4).align()
    assert msa.data_type == 3
    rows = msa.alignment()
    assert len({len(r) for r in rows}) == 1 and len(rows[0]) % 3 == 0
    for r, s in zip(rows, seqs):
        cod = [r[i:i + 3] for i in range(0, len(r), 3)]
        assert all(c == "---" or "-" not in c for c in cod)
        assert "".join(c for c in cod if c != "---") == "".join(s[i:i + 3] if s[i:i + 3] in CODONS else "NNN" for i in range(0, len(s), 3))
    # the same walk from the oracle's pieces
    results, root, banded = oracle_codon_walk(oracle, names, seqs, nwk, anchors=bool(anchors), hit_length=6, trim=5, offset=4)
    scores = [(r.score, r.cols) for r in results]
    assert msa.n_internal == len(scores)
    for k, (sc, cols) in enumerate(scores):
        r = msa.node_result(k)
        assert r.score == sc and np.array_equal(r.cols, cols), "node %d" % k
        _l, _r, _m, b = msa.node_job(k)
        assert (b is not None) == bool(anchors)
    if anchors:
        assert banded >= len(scores) // 2                         # the anchors found in the translation do narrow the matrices
    same_graph(msa.node_graph(2 * len(seqs) - 2), root, "root")
    # ancestors' rows: three letters per column
    anc, _ = host.codon_alphabet()
    allrows = msa.alignment_all()
    sa = msa.node_graph(2 * len(seqs) - 2).attrs()[0]
    want = "".join("---" if a[2] in (5, 6) or a[1] == 5 else anc[3 * a[0]:3 * a[0] + 3] for a in sa[1:-1])
    assert allrows[2 * len(seqs) - 2] == want
