"""Diagnostic: the node alignments of the bench tree through pg_fill_ring (the tree walk runs on it) and pg_fill_pipe,
cell-by-cell score comparison of every node; prints the first differing cell with its diagonal's class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
os.environ["PAGAN_DP_FILL"] = "ring"
names, seqs, nwk = synth.evolve_balanced(leaves, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
for k in range(msa.n_internal - 1, -1, -1):
    if msa.node_info(k).level < 1:
        continue
    job = msa.node_job(k)
    l, r, m, b = job
    os.environ["PAGAN_DP_FILL"] = "ring"; A = pg.Batch([job]); A.run(); A.sync(); sa = A.debug_scores(0)
    os.environ["PAGAN_DP_FILL"] = "pipe"; B = pg.Batch([job])
    try:
        B.run(); B.sync()
    except Exception as e:
        print("node", k, "pipe run failed:", e, flush=True)
    sb = B.debug_scores(0)
    same = (sa.view(np.int64) == sb.view(np.int64)).all(axis=1)
    if same.all():
        print("node", k, "level", msa.node_info(k).level, "identical", flush=True)
        continue
    Lx, Ly = l.n_sites - 1, r.n_sites - 1
    up = np.maximum(b.upper[:Lx].astype(np.int64), 0); lw = np.minimum(b.lower[:Lx].astype(np.int64), Ly - 1)
    ii = np.arange(Lx); nd = Lx + Ly - 1; d = np.arange(nd)
    imin = np.searchsorted(ii + lw, d, side="left"); imax = np.searchsorted(ii + up, d, side="right") - 1
    off = np.concatenate([[0], np.cumsum(np.maximum(imax - imin + 1, 0))])
    cls, _ = pg.debug_plan(l, r, b)
    bad = np.nonzero(~same)[0]
    first = bad[0]; dd = int(np.searchsorted(off, first, side="right") - 1); i = int(imin[dd] + first - off[dd]); j = dd - i
    print("node", k, "level", msa.node_info(k).level, "DIFFERENT: %d cells; first d=%d i=%d j=%d lo %d hi %d class %d (prev classes %s)" %
          (bad.size, dd, i, j, imin[dd], imax[dd], cls[dd], cls[dd - 6:dd].tolist()), "ring", sa[first], "pipe", sb[first], flush=True)
    for side, g, s in (("L", l, i), ("R", r, j)):
        e0, e1 = g.bwd_off[s], g.bwd_off[s + 1]
        print("   ", side, "site", s, "edges from", (s - g.bwd_src[e0:e1]).tolist(), "w", g.bwd_logw[e0:e1].tolist())
    break
