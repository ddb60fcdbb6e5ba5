"""Generates tests/golden/*.npz.

PROVENANCE: the reference (pagan2-msa @ 2024_08_07) ships no vectors and cannot be built in this
image, so these are NOT reference outputs.  They are inputs plus the outputs of
OUR oracle restatement at the commit that created them, frozen so that later rounds notice any
drift of the oracle or of the GPU path (and so the GPU box needs nothing but this repo).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pagan2_msa_amd import abi, synth  # noqa: E402
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pack_graph(prefix, g):
    return {prefix + k: getattr(g, k) for k in ("state", "bwd_off", "bwd_src", "bwd_logw", "bwd_eid")} | {
        prefix + "n_edges": np.int32(g.n_edges)}


def save(name, left, right, model, band, flags):
    r = oracle.dp_align(left, right, model, band, flags=flags)
    d = {}
    d.update(pack_graph("l_", left))
    d.update(pack_graph("r_", right))
    d["table"] = model.log_score
    d["params"] = np.array(model.params, np.float32)
    d["flags"] = np.int32(flags)
    if band is not None:
        d["upper"], d["lower"] = band.upper, band.lower
    d["status"], d["score"] = np.int32(r.status), np.float64(r.score)
    d["end"] = np.array(r.end, np.int32)
    d["cols"], d["left_used"], d["right_used"] = r.cols, r.left_used, r.right_used
    d["cells"] = np.int64(r.cells)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "cells", r.cells, "score", r.score, "cols", r.cols.shape[0])


def main():
    oracle.build()
    # 1: plain DNA leaves, full matrix
    _, seqs, _ = synth.evolve_balanced(2, 180, sub=0.08, indel_start=0.03, seed=101)
    save("leaves_full", synth.chain_graph(seqs[0]), synth.chain_graph(seqs[1]), synth.jc_like_dna_model(0.1), None, 0)
    # 2: same with both option bits
    save("leaves_full_flags3", synth.chain_graph(seqs[0]), synth.chain_graph(seqs[1]), synth.jc_like_dna_model(0.1), None, 3)
    # 3: prefix-anchor band on 2 kb leaves
    _, seqs, _ = synth.evolve_balanced(2, 2000, branch=0.01, sub=0.012, indel_start=0.003, seed=102)
    ol, orr = oracle.OGraph.leaf(seqs[0]), oracle.OGraph.leaf(seqs[1])
    band, _ = oracle.define_tunnel(ol, orr)
    save("leaves_banded", ol.flatten(), orr.flatten(), synth.jc_like_dna_model(0.02), band, 0)
    # 4: homopolymer leaves (multi-edge leaf sites)
    s1, s2 = "ACGTTTTTTACGGGGACCCCCCCCATTTAGGA" * 4, "ACGTTTTTACGGGGGACCCCCCATTAGGA" * 4
    save("homopolymer", oracle.OGraph.leaf(s1, flags=2).flatten(), oracle.OGraph.leaf(s2, flags=2).flatten(),
         synth.jc_like_dna_model(0.1), None, 0)
    # 5: random multi-edge graphs with long edges, dead sites, tie-rich 23-state table, banded
    rng = np.random.default_rng(7)
    left = synth.random_graph(260, 23, 103, p_extra=0.35, max_span=22, p_dead=0.02)
    right = synth.random_graph(240, 23, 104, p_extra=0.35, max_span=22)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    centre = np.linspace(0, Ly - 1, Lx)
    up = np.maximum.accumulate(np.clip(centre - rng.integers(4, 30, Lx), 0, None)).astype(np.int32)
    lo = np.maximum.accumulate(np.clip(centre + rng.integers(4, 30, Lx), 0, Ly + 5)).astype(np.int32)
    up[0] = 0
    save("graphs_banded_23", left, right, synth.random_model(23, 5), abi.Band(up, lo), 0)
    # 6: unreachable end corner
    g = synth.chain_graph("ACGTACGTACGTACGTACGT")
    save("unreachable", g, g, synth.jc_like_dna_model(0.1), abi.Band(np.zeros(21, np.int32), np.full(21, 3, np.int32)), 0)


if __name__ == "__main__":
    main()
