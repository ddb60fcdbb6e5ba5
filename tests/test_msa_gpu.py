"""GPU: the host tree walk end to end -- every internal node's alignment (fill + traceback on
the device) is compared with the oracle on the node's own inputs; upper nodes are
graph-vs-graph alignments with multi-edge sites, skip columns and bands over gapped strings."""
import numpy as np
import pytest

from pagan2_msa_amd import host, synth

pytestmark = pytest.mark.gpu


def check_tree(msa, seqs, oracle, flags=0):
    kinds = set()
    for k in range(msa.n_internal):
        left, right, model, band = msa.node_job(k)
        want = oracle.dp_align(left, right, model, band, flags=flags)
        got = msa.node_result(k)
        assert got.same_alignment(want), "node %d differs" % k
        kinds.update(got.cols[:, 2].tolist())
        info = msa.node_info(k)
        assert info.cells == want.cells and info.score == want.score
    rows = msa.alignment()
    assert len({len(r) for r in rows}) == 1
    for r, s in zip(rows, seqs):
        assert r.replace("-", "") == s
    return kinds


def test_banded_tree_16x1500(pg, oracle):
    names, seqs, nwk = synth.evolve_balanced(16, 1500, branch=0.01, sub=0.012, indel_start=0.003, mean_len=5, seed=21)
    msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
    kinds = check_tree(msa, seqs, oracle)
    assert {2, 3, 4, 5, 6} <= kinds            # matched, gapped and skipped columns all occur
    assert max(msa.node_info(k).level for k in range(15)) == 3
    t = msa.timing()
    assert t["dp_fill_dev_s"] > 0 and t["total_s"] >= t["dp_wall_s"]


def test_full_matrix_tree_8x300_with_option_bits(pg, oracle):
    names, seqs, nwk = synth.evolve_balanced(8, 300, branch=0.05, sub=0.04, indel_start=0.01, mean_len=4, seed=22)
    for flags in (0, 1, 2):
        msa = host.Msa(names, seqs, nwk, use_anchors=0, dp_flags=flags).align()
        check_tree(msa, seqs, oracle, flags)


def test_caterpillar_with_deleted_ranges(pg, oracle):
    names, seqs, nwk = synth.evolve_caterpillar(14, 150, seed=2)
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
    check_tree(msa, seqs, oracle)
    types = np.concatenate([msa.node_graph(14 + k).attrs()[0][:, 1] for k in range(13)])
    assert (types == 5).any()                  # non_real sites entered later DPs as dead rows


def test_homopolymer_leaves_tree(pg, oracle):
    names, seqs, nwk = synth.evolve_balanced(4, 400, branch=0.02, sub=0.02, indel_start=0.004, seed=23)
    seqs = [s.replace("AC", "AAAC", 20) for s in seqs]
    msa = host.Msa(names, seqs, nwk, use_anchors=0, leaf_flags=2).align()
    check_tree(msa, seqs, oracle)


def test_tree_errors(pg):
    with pytest.raises(pg.PaganError) as e:
        host.Msa(["a", "b", "c"], ["ACGT", "ACGT", "ACGT"], "(a:0.1,b:0.1,c:0.1);")
    assert e.value.code == host.PAGAN_E_TREE
    with pytest.raises(pg.PaganError):
        host.Msa(["a", "b"], ["ACGT", "ACGT"], "(a:0.1,x:0.1);")


def test_fasta_output_in_tree_order(pg, tmp_path):
    """pagan_msa_write_fasta: Fasta_reader::write_fasta over the leaf rows, leaves left to right."""
    names, seqs, _ = synth.evolve_balanced(4, 200, branch=0.02, sub=0.02, indel_start=0.01, mean_len=3, seed=5)
    nwk = "((%s:0.02,%s:0.02):0.02,(%s:0.02,%s:0.02):0.02);" % (names[2], names[0], names[3], names[1])
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
    out = tmp_path / "aligned.fas"
    msa.write_fasta(out, chars_by_line=50)
    rows = msa.alignment()
    lines = out.read_text().split("\n")
    assert lines[-1] == ""
    entries, name = {}, None
    order = []
    for ln in lines[:-1]:
        if ln.startswith(">"):
            name = ln[1:]; order.append(name); entries[name] = []
        else:
            assert 0 < len(ln) <= 50
            entries[name].append(ln)
    assert order == [names[2], names[0], names[3], names[1]]
    for k, nm in enumerate(names):
        assert "".join(entries[nm]) == rows[k]
        assert all(len(x) == 50 for x in entries[nm][:-1])


AA = "ARNDCQEGHILKMFPSTWYV"


def check_graphs(msa, seqs, oracle, leaf_alpha, char_as):
    """Every node's graph as the product built it against the oracle's builder fed the same paths.  The two
    branches under a node have the same length in these synthetic trees, so each is half the node's dist."""
    n = len(seqs)
    og = [oracle.OGraph.leaf(s, leaf_alpha) for s in seqs] + [None] * (n - 1)
    for k in range(msa.n_internal):
        info = msa.node_info(k)
        pars = oracle.protein_model(info.dist)[1] if char_as == 20 else oracle.dna_parsimony()
        og[info.node] = oracle.OGraph.parent(og[info.left], og[info.right], msa.node_result(k), info.dist / 2,
                                             info.dist / 2, pars, char_as)
        a, b = msa.node_graph(info.node).flatten(), og[info.node].flatten()
        for f in ("state", "bwd_off", "bwd_src", "bwd_eid"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), "node %d: %s" % (info.node, f)
        assert a.bwd_logw.tobytes() == b.bwd_logw.tobytes()


def test_protein_tree_16x200(pg, oracle):
    """BASELINE config 3 scaled down: WAG, 211-letter alphabet, no anchors -> full matrices (tiled kernel)."""
    names, seqs, nwk = synth.evolve_balanced(16, 200, branch=0.05, sub=0.08, indel_start=0.01, mean_len=4, seed=31,
                                             alphabet=AA)
    msa = host.Msa(names, seqs, nwk, use_anchors=0).align()          # data type guessed from the residues
    assert msa.data_type == 2
    check_tree(msa, seqs, oracle)
    left, right, model, band = msa.node_job(msa.n_internal - 1)
    assert model.n_states == 211 and band is None
    assert max(left.state.max(), right.state.max()) > 20              # pair codes enter the upper DPs
    want, _ = oracle.protein_model(msa.node_info(0).dist)
    assert msa.node_job(0)[2].table.tobytes() == want.table.tobytes()
    check_graphs(msa, seqs, oracle, oracle.protein_leaf_alphabet(), 20)


def test_protein_tree_banded_large_table_on_the_banded_kernel(pg, oracle):
    names, seqs, nwk = synth.evolve_balanced(8, 700, branch=0.02, sub=0.03, indel_start=0.004, mean_len=4, seed=32,
                                             alphabet=AA)
    msa = host.Msa(names, seqs, nwk, use_anchors=1, data_type=2, prefix_hit_length=12).align()
    check_tree(msa, seqs, oracle)
    assert all(msa.node_job(k)[3] is not None for k in range(msa.n_internal))


def test_work_queue_over_devices_cfg5_reduced(pg, oracle):
    """BASELINE config 5 scaled down (64 x 1 kb, anchored): the ready-queue walk over min(2, #GPUs) devices."""
    ndev = min(2, pg.device_count())
    names, seqs, nwk = synth.evolve_balanced(64, 1000, branch=0.02, sub=0.016, indel_start=0.0016, mean_len=4, seed=33)
    msa = host.Msa(names, seqs, nwk, use_anchors=1, n_devices=ndev, first_device=0).align()
    check_tree(msa, seqs, oracle)
    assert {msa.node_device(k) for k in range(msa.n_internal)} == set(range(ndev))
    # the round-at-a-time API (one process per GPU drives it through dist.align_sharded) gives the same walk
    from pagan2_msa_amd import dist as pdist
    again = host.Msa(names, seqs, nwk, use_anchors=1)
    rounds = pdist.align_sharded(again, host.assign_units)
    assert [r[0] for r in rounds] == [32, 16, 8, 4, 2, 1]
    assert again.alignment() == msa.alignment()
    assert all(again.node_result(k).same_alignment(msa.node_result(k)) for k in range(63))


def test_force_gapped_tunnel(pg, oracle):
    """--force-gap (node.cpp:124-141 + replace_largest_tunnel_block_with_gap_tunnel): a pair with an unalignable
    stretch; the tunnel from overlapping hits, then with its largest empty block replaced by a gap -- both aligned on
    the GPU against the oracle, and the tree walk taking the same route under a small memory budget."""
    _, seqs, _ = synth.evolve_balanced(2, 5000, branch=0.02, sub=0.02, indel_start=0.002, mean_len=5, seed=51)
    rng = np.random.default_rng(9)
    junk = lambda n: "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])
    a, b = seqs[0][:2000] + junk(700) + seqs[0][2700:], seqs[1][:2000] + junk(700) + seqs[1][2700:]
    gl, gr = host.HGraph.leaf(a).flatten(), host.HGraph.leaf(b).flatten()
    band, blocks = host.define_tunnel_overlapping(host.drop_bad_hits(host.prefix_hits(a, b, 20)), a, b)
    done, forced, _ = host.force_gap(band, blocks)
    assert done
    model, _ = host.dna_model([0.25] * 4, 0.04)
    cells = []
    for bd in (band, forced):
        got, want = pg.align(gl, gr, model, bd), oracle.dp_align(gl, gr, model, bd)
        assert got.same_alignment(want) and got.status == 0
        cells.append(got.cells)
    assert cells[1] < cells[0] - 200000
    # the forced tunnel aligns the junk as one long gap pair instead of scattered matches
    names = ["a", "b"]
    nwk = "(a:0.02,b:0.02);"
    free = host.Msa(names, [a, b], nwk, anchor_mode=1, prefix_hit_length=20).align()
    assert free.node_info(0).n_forced_gaps == 0
    need = pg.lib().pagan_dp_predict_bytes(gl.n_sites, gr.n_sites, __import__("ctypes").byref(band.c))
    tight = host.Msa(names, [a, b], nwk, anchor_mode=1, prefix_hit_length=20, force_gap=1, device_mem_budget=int(need * 0.8)).align()
    info = tight.node_info(0)
    assert info.n_forced_gaps == 1 and info.cells == cells[1]
    left, right, m2, b2 = tight.node_job(0)
    assert np.array_equal(b2.upper, forced.upper) and np.array_equal(b2.lower, forced.lower)
    assert tight.node_result(0).same_alignment(oracle.dp_align(left, right, m2, b2))
    with pytest.raises(pg.PaganError) as e:                       # without --force-gap the node does not fit: the reference exits
        host.Msa(names, [a, b], nwk, anchor_mode=1, prefix_hit_length=20, device_mem_budget=int(need * 0.8)).align()
    assert e.value.code == host.PAGAN_E_MEMCAP
    with pytest.raises(pg.PaganError) as e:                       # nothing large enough to replace
        host.Msa(names, [a, b], nwk, anchor_mode=1, prefix_hit_length=20, force_gap=1, force_gap_threshold=10 ** 9, device_mem_budget=int(need * 0.8)).align()
    assert e.value.code == host.PAGAN_E_MEMCAP
