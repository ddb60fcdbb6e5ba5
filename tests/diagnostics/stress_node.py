"""Diagnostic: re-run one node alignment of a 32 x 100 kb tree many times with poisoned output
buffers and compare every run with the oracle (hunts nondeterminism in the kernels)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host
import oracle

node = int(sys.argv[1]) if len(sys.argv) > 1 else 29
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
names, seqs, nwk = synth.evolve_balanced(32, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=5)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
job = msa.node_job(node)
want = oracle.dp_align(*job)
print("node", node, "cells", want.cells, "score", want.score, "cols", want.cols.shape[0], flush=True)
batch = pg.Batch([job])
bad = 0
for r in range(reps):
    pg.lib().pagan_batch_debug_poison(batch._h)
    batch.run()
    got = batch.fetch()[0]
    if not got.same_alignment(want):
        bad += 1
        ncol = min(got.cols.shape[0], want.cols.shape[0])
        diff = np.nonzero((got.cols[:ncol] != want.cols[:ncol]).any(axis=1))[0]
        print("rep", r, "MISMATCH status", got.status, "score", got.score, "dscore", got.score - want.score,
              "ncols", got.cols.shape[0], "first col diff", diff[:1], "n diff", diff.size, flush=True)
print("bad runs: %d of %d" % (bad, reps))
