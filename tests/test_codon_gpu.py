"""GPU: the aligner under the real codon model (Kosiol & Goldman's empirical matrix, 1892 states, a 14 MB score table;
the producer is checked bit for bit on the CPU in test_codon_cpu.py) -- the tree walk on DNA read as codons, and the banded
and tiled kernels on graphs whose states come from such a walk, against the oracle."""
import numpy as np
import pytest

from pagan2_msa_amd import host, synth

from test_codon_cpu import CODONS, evolve_codons, oracle_codon_walk
from test_host_cpu import same_graph
from test_pipe_gpu import banded_job, same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("anchors", [0, 1])
def test_tree_walk_on_codons_on_the_device(pg, oracle, anchors):
    """16 leaves of ~250 codons: every node's alignment (the 1892 x 1892 table stays in HBM / L2: the tiled kernel stages
    scores out of it, the banded kernel's assist waves gather them) against the same walk made from the oracle's pieces;
    with anchors the bands come from the translated codon strings."""
    names, seqs, nwk = evolve_codons(16, 250, seed=12, branch=0.03, sub=0.04, indel_start=0.012, mean_len=3)
    seqs[4] = seqs[4][:60] + "TAA" + seqs[4][63:]
    seqs[9] = seqs[9] + "AC"                                            # a last partial triplet
    msa = host.Msa(names, seqs, nwk, data_type=3, use_anchors=anchors, prefix_hit_length=8, anchors_offset=6).align()
    assert msa.data_type == 3
    results, root, banded = oracle_codon_walk(oracle, names, seqs, nwk, anchors=bool(anchors), hit_length=8, trim=5, offset=6)
    assert msa.n_internal == len(results) == 15
    states, routes = set(), set()
    for k, want in enumerate(results):
        got = msa.node_result(k)
        assert got.same_alignment(want), "node %d differs" % k
        left, right, model, band = msa.node_job(k)
        assert (band is not None) == bool(anchors) and model.n_states == 1892
        routes.add(pg.debug_route(left, right, model, band)[0])
        states.update(left.state.tolist())
    assert max(states) > 61                                              # pair codes entered later alignments
    assert routes <= {"pg_fill_pipe (large table)", "pg_fill_tiles_flow"}
    if anchors:
        assert banded >= 8 and "pg_fill_pipe (large table)" in routes
    same_graph(msa.node_graph(30), root, "root")
    for r, s in zip(msa.alignment(), seqs):
        cod = [r[i:i + 3] for i in range(0, len(r), 3)]
        assert "".join(c for c in cod if c != "---") == "".join(s[i:i + 3] if s[i:i + 3] in CODONS else "NNN" for i in range(0, len(s), 3))


def codon_like_states(rng, n):
    """mostly codons, some pair codes and NNN -- what upper nodes of a codon walk carry"""
    st = rng.integers(0, 61, n)
    pairs = rng.random(n) < 0.25
    st[pairs] = rng.integers(62, 1892, int(pairs.sum()))
    st[rng.random(n) < 0.02] = 61
    return st.astype(np.int32)


@pytest.mark.parametrize("seed", [0, 2])
def test_the_real_codon_table_on_the_banded_kernel(pg, oracle, seed):
    left, right, _, band = banded_job(seed, max_span=40 if seed < 2 else 8)
    rng = np.random.default_rng(300 + seed)
    left.state[1:-1] = codon_like_states(rng, left.n_sites - 2)
    right.state[1:-1] = codon_like_states(rng, right.n_sites - 2)
    model, _ = host.codon_model(0.12 + 0.1 * seed)
    assert model.table.tobytes() == oracle.codon_model(0.12 + 0.1 * seed)[0].table.tobytes()
    assert pg.debug_route(left, right, model, band)[0] == "pg_fill_pipe (large table)"
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


def test_the_real_codon_table_on_the_tiled_kernel(pg, oracle):
    rng = np.random.default_rng(9)
    left, right = synth.random_graph(320, 15, 5, p_extra=0.2, max_span=12), synth.random_graph(290, 15, 6, p_extra=0.2, max_span=12)
    left.state[1:-1] = codon_like_states(rng, left.n_sites - 2)
    right.state[1:-1] = codon_like_states(rng, right.n_sites - 2)
    model, _ = host.codon_model(0.3)
    assert pg.debug_route(left, right, model, None)[0] == "pg_fill_tiles_flow"
    same(pg.align(left, right, model), oracle.dp_align(left, right, model), "codon model")
