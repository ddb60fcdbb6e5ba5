"""GPU: BASELINE config 1 (`--queryfile reads --pileup-alignment --homopolymer`) as plumbing over the GPU aligner: a
reference read + 9 reads (one unrelated), every step's alignment -- homopolymer leaves with multi-edge sites against the
growing root graph, reads settings -- against the oracle, and the chain's accept / reject decisions against the chain
spelled out from the oracle's pieces (tests/test_pileup_cpu.py)."""
import numpy as np
import pytest

from pagan2_msa_amd import host

from test_pileup_cpu import make_reads, oracle_chain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("anchors", [0, 1])
def test_pileup_chain_on_the_gpu(pg, oracle, anchors):
    names, seqs = make_reads(5)
    p = host.Pileup(names, seqs, leaf_flags=2, min_overlap=0.85, use_anchors=anchors, prefix_hit_length=12).align()
    multi = 0
    for k in range(p.n_steps):
        left, right, model, band = p.step_job(k)
        assert (band is not None) == bool(anchors) or p.step(k).status == 0
        multi += int((np.diff(right.bwd_off) > 1).sum())
        assert p.step_result(k).same_alignment(oracle.dp_align(left, right, model, band))
    assert multi > 0                                             # homopolymer runs gave the read leaves skip-back edges
    if not anchors:
        want, _ = oracle_chain(oracle, seqs, min_overlap=np.float32(0.85))
        for k, (i, ok, ov, idn, res) in enumerate(want):
            s = p.step(k)
            assert (s.read, bool(s.accepted)) == (i, ok) and np.float32(s.overlap) == np.float32(ov)
            assert p.step_result(k).same_alignment(res)
    rows = p.alignment()
    for k, (r, s) in enumerate(zip(rows, seqs)):
        assert r == "" or r.replace("-", "") == s
    assert rows[4] == "" and sum(1 for r in rows if r) == len(seqs) - 1
