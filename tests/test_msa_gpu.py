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
