"""Two independent readings of the reference must agree: the oracle (oracle/oracle_dp.cpp, CSR arrays, its own loop
structure) against tests/pycheck.py (linked edge lists with the reference's cursor, Matrix_pointer cells, the reference's
row-major fill order, its functions one by one) -- on every golden case and on 200 randomized graph pairs with
multi-edge sites, dead sites, tunnels and both option bits.  Then hand-worked vectors for the rules ties hinge on, with
the arithmetic written out."""
import numpy as np
import pytest

import _golden
import pycheck
from pagan2_msa_amd import abi, synth

f32 = np.float32
NEG = float("-inf")


@pytest.mark.parametrize("name", _golden.NAMES)
def test_golden_cases_two_readings(oracle, name):
    left, right, model, band, flags, d = _golden.load(name)
    py = pycheck.align(left, right, model, band, flags)
    res = oracle.dp_align(left, right, model, band, flags=flags)
    assert pycheck.same(py, res), name
    assert py["status"] == int(d["status"])
    if py["status"] == 0:
        assert np.float64(py["score"]).tobytes() == np.float64(d["score"]).tobytes()
        assert np.array_equal(py["cols"], d["cols"])


def random_band(rng, Lx, Ly, width):
    """A monotone tunnel around a random walk from (0,0) to (Lx-1, Ly-1)."""
    centre = np.round(np.linspace(0, Ly - 1, Lx) + np.cumsum(rng.integers(-1, 2, Lx))).astype(int)
    centre = np.maximum.accumulate(np.clip(centre, 0, Ly - 1))
    up = np.maximum.accumulate(np.clip(centre - width, 0, Ly - 1))
    lo = np.maximum.accumulate(np.clip(centre + width, 0, Ly - 1))
    up[0] = 0
    lo[-1] = Ly - 1
    return abi.Band(up.astype(np.int32), lo.astype(np.int32))


@pytest.mark.parametrize("block", range(8))
def test_fuzz_two_readings(oracle, block):
    """200 cases in 8 blocks of 25."""
    ties = 0
    for case in range(25):
        seed = 1000 * block + case
        rng = np.random.default_rng(seed)
        n_states = int(rng.choice([4, 15, 23]))
        left = synth.random_graph(int(rng.integers(3, 45)), n_states, seed, p_extra=float(rng.choice([0.0, 0.3, 0.6])),
                                  max_deg=4, max_span=int(rng.integers(2, 9)), p_dead=float(rng.choice([0.0, 0.05])))
        right = synth.random_graph(int(rng.integers(3, 45)), n_states, seed + 7, p_extra=float(rng.choice([0.0, 0.3, 0.6])),
                                   max_deg=4, max_span=int(rng.integers(2, 9)), p_dead=float(rng.choice([0.0, 0.05])))
        model = synth.jc_like_dna_model(0.1) if (n_states == 15 and rng.random() < 0.5) else synth.random_model(n_states, seed)
        flags = int(rng.integers(0, 4))
        band = None
        if rng.random() < 0.5:
            band = random_band(rng, left.n_sites - 1, right.n_sites - 1, int(rng.integers(2, 12)))
        py = pycheck.align(left, right, model, band, flags)
        res = oracle.dp_align(left, right, model, band, flags=flags)
        assert pycheck.same(py, res), "seed %d" % seed
        ties += int(py["status"] == 0)
    assert ties > 5


# ---- hand-worked vectors ----------------------------------------------------------------------------------------------
def chain(states):
    """A plain sequence: start site, the given states, stop site; one bwd edge per site, weight 1 (log 0)."""
    n = len(states) + 2
    st = np.array([-1] + list(states) + [-1], np.int32)
    off = np.array([0] + list(range(0, n)), np.int32)          # site 0 has no bwd edge
    src = np.arange(0, n - 1, dtype=np.int32)
    return abi.Graph(st, off, src, np.zeros(n - 1, np.float32), np.arange(1, n, dtype=np.int32), n_edges=n)


def model2(match=1.0, mism=-2.0, go=-3.0, ge=-0.5, gE=-0.25, ng=-0.125):
    t = np.full((2, 2), mism, np.float32)
    np.fill_diagonal(t, match)
    return abi.Model(t, go, ge, gE, ng)


def both(oracle, left, right, model, band=None, flags=0):
    py = pycheck.align(left, right, model, band, flags)
    res = oracle.dp_align(left, right, model, band, flags=flags)
    assert pycheck.same(py, res)
    return py, res


def test_hand_single_match_and_the_open_penalty_at_the_start(oracle):
    """A vs A (Lx = Ly = 2).  M[1][1] = M[0][0] + (2*ng + s) = 0 + (2*(-0.125) + 1) = 0.75; the end corner adds ng:
    0.75 - 0.125 = 0.625.  The alternative X then Y: X[1][0] opens from M[0][0] with open_pen(p == 0) = 0 (reduced
    terminal penalties): 0 + ng + 0 = -0.125; Y[1][1] = X[1][0] + 0 + go = -3.125; closes at -3.125 < 0.625."""
    py, _ = both(oracle, chain([0]), chain([0]), model2())
    assert py["score"] == 0.625 and py["cols"].tolist() == [[1, 1, 2]]
    # without the reduced terminal penalty the X opening costs go as well: X[1][0] = 0 + (-0.125) + (-3) = -3.125
    va = pycheck.ViterbiAlignment(chain([0]), chain([0]), model2(), flags=2)
    va.align()
    assert va.xgap.at(1, 0).score == -3.125 and pycheck.ViterbiAlignment(chain([0]), chain([0]), model2()).align() is not None
    va0 = pycheck.ViterbiAlignment(chain([0]), chain([0]), model2())
    va0.align()
    assert va0.xgap.at(1, 0).score == -0.125 and va0.xgap.at(1, 0).matrix == pycheck.M_MAT


def test_hand_tie_between_gap_orders_goes_to_the_first_candidate(oracle):
    """A vs C with a mismatch so bad that two gaps win: the path is X then Y or Y then X.  Lx = Ly = 2.
    X[1][0] = M[0][0] + ng + 0 = -0.125 (end gap row j = 0, open at p = 0 free); Y[0][1] likewise -0.125.
    Y[1][1] candidates over the right edge (q = 0), in order ext, double, open:
        Y[1][0] + gE = -inf;  X[1][0] + 0 + go = -0.125 - 3 = -3.125;  M[1][0] + ... = -inf      -> from X, -3.125
    X[1][1] likewise from Y[0][1]: -3.125.  End corner, in order: M-term (M[1][1] = 0 + (-0.25 - 20) = -20.25, + ng =
    -20.375), X-close X[1][1] + 0 = -3.125 (strictly bigger: taken), Y-close Y[1][1] + 0 = -3.125 -- a TIE, and
    first_is_bigger is strict, so the X-close stays: last column is an X gap (left residue), before it a Y gap."""
    py, res = both(oracle, chain([0]), chain([1]), model2(mism=-20.0))
    assert py["score"] == -3.125
    assert py["end"][0] == pycheck.X_MAT
    assert py["cols"].tolist() == [[-1, 1, 4], [1, -1, 3]]


def test_hand_pair_order_row_major_with_equal_candidates(oracle):
    """Both sites before the match have two bwd edges each, all four predecessor cells hold the same M score: the pair
    (l0, r0) is evaluated first (viterbi_alignment.cpp:1396-1433: first pair, then the right site's further edges with
    l0, then each further left edge with r0 and the right site's further edges) and strict > keeps it.
    Left: start, A, A, A, stop with an extra edge site1 -> site3 (skips site 2); right the same.  All emissions equal."""
    def skip_graph():
        st = np.array([-1, 0, 0, 0, -1], np.int32)
        # bwd lists: site1 <- 0 ; site2 <- 1 ; site3 <- 2 (first), <- 1 (second) ; stop <- 3
        off = np.array([0, 0, 1, 2, 4, 5], np.int32)
        src = np.array([0, 1, 2, 1, 3], np.int32)
        eid = np.array([1, 2, 3, 5, 4], np.int32)
        return abi.Graph(st, off, src, np.zeros(5, np.float32), eid, n_edges=6)
    t = np.zeros((1, 1), np.float32)
    model = abi.Model(t, -3.0, -0.5, -0.25, -0.125)
    left, right = skip_graph(), skip_graph()
    FLAGS = 2       # --no-reduced-terminal-penalties: a gap at the very start pays go like any other (else it is free and ties)
    va = pycheck.ViterbiAlignment(left, right, model, flags=FLAGS)
    path = va.align()
    c = va.match.at(3, 3)
    # tM = 2*ng + s = -0.25.  M[1][1] = -0.25 (one match), M[2][2] = -0.5 (two).  X[1][0] = 0 + ng + go = -3.125, so
    # M[2][1] = X[1][0] + (0 + ng + s) = -3.25.  Candidates at (3,3) in the reference's order:
    #   (l0,r0) = (2,2): M[2][2] + tM = -0.5 - 0.25 = -0.75
    #   (l0,r1) = (2,1): M[2][1] + tM = -3.5
    #   (l1,r0) = (1,2): -3.5
    #   (l1,r1) = (1,1): M[1][1] + tM = -0.25 - 0.25 = -0.5   <- strictly bigger: wins although it comes last
    assert va.match.at(1, 1).score == -0.25 and va.match.at(2, 2).score == -0.5
    assert c.score == -0.5 and (c.x_ind, c.y_ind) == (1, 1) and (c.x_edge_ind, c.y_edge_ind) == (5, 5)
    cols = va.columns(path).tolist()
    # the skipped sites come out as skip columns between the two matches: right one first, then left (insert_preexisting_gap
    # pushes x skips then y skips onto a stack that is reversed at the end, viterbi_alignment.h:146-193)
    assert cols == [[1, 1, 2], [-1, 2, 6], [2, -1, 5], [3, 3, 2]]
    res = oracle.dp_align(left, right, model, flags=FLAGS)
    assert pycheck.same(pycheck.align(left, right, model, flags=FLAGS), res)
    assert res.left_used.tolist() == [1, 4, 5] and res.right_used.tolist() == [1, 4, 5]
    # now make the two routes exactly equal: log weight -0.125 on each skip edge gives
    # (l1,r1) = ((M[1][1] + tM) + lw) + rw = ((-0.25 - 0.25) - 0.125) - 0.125 = -0.75 == (l0,r0): the FIRST pair stays.
    def skip_w():
        g = skip_graph()
        lw = np.array([0, 0, 0, -0.125, 0], np.float32)
        return abi.Graph(g.state, g.bwd_off, g.bwd_src, lw, g.bwd_eid, n_edges=6)
    va = pycheck.ViterbiAlignment(skip_w(), skip_w(), model, flags=FLAGS)
    va.align()
    c = va.match.at(3, 3)
    assert c.score == -0.75 and (c.x_ind, c.y_ind) == (2, 2)          # tie: the earlier pair stays
    assert pycheck.same(pycheck.align(skip_w(), skip_w(), model, flags=FLAGS), oracle.dp_align(skip_w(), skip_w(), model, flags=FLAGS))


def test_hand_float_promotion_points(oracle):
    """2*log_non_gap is a FLOAT product promoted afterwards (viterbi_alignment.cpp:1364): with ng = fl32(-0.1) the double
    product 2 * (double)ng equals the float product here (a power-of-two factor is exact), but 0.0f + ng and the sum with
    the float table entry are where a double evaluation would differ: s + ng + ng in doubles versus (2*ng as float) + s."""
    ng, s = f32(-0.1), f32(0.3)
    model = abi.Model(np.array([[s]], np.float32), -3.0, -0.5, -0.25, ng)
    va = pycheck.ViterbiAlignment(chain([0]), chain([0]), model)
    va.align()
    want = 0.0 + (float(f32(2) * ng) + float(s))                      # ((M[0][0] + tM) + lw) + rw with lw = rw = 0
    assert va.match.at(1, 1).score == want
    assert want != float(2 * -0.1 + 0.3)                              # not the decimal arithmetic
    assert np.float64(oracle.dp_align(chain([0]), chain([0]), model).score).tobytes() == np.float64(want + float(ng)).tobytes()
