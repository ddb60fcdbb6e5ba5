"""Diagnostic (GPU): the node alignments of the headline workload (walk on the ring kernel), each through pg_fill_ring and
pg_fill_pipe, cell-by-cell score comparison; the first differing cells of the first differing node."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAGAN_DP_FILL"] = "ring"
os.environ["PAGAN_DP_COMPACT"] = "0"
os.environ["PAGAN_DP_SCORE_CHECK"] = "0"
os.environ["PAGAN_DP_RERUN"] = "0"
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host
from dbg_pipe import diag_index

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
names, seqs, nwk = synth.evolve_balanced(leaves, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
order = sorted(range(msa.n_internal), key=lambda k: -msa.node_info(k).level)
for k in order:
    if msa.node_info(k).level == 0:
        continue
    job = msa.node_job(k)
    l, r, m, b = job
    os.environ["PAGAN_DP_FILL"] = "ring"; A = pg.Batch([job]); A.run(); A.sync(); sa = A.debug_scores(0); A.close()
    os.environ["PAGAN_DP_FILL"] = "pipe"; B = pg.Batch([job]); B.run(); B.sync(); sb = B.debug_scores(0); B.close()
    same = (sa.view(np.int64) == sb.view(np.int64)).all(axis=1)
    print("node", k, "level", msa.node_info(k).level, "cells", same.size, "different", int((~same).sum()), flush=True)
    if not same.all():
        n, hfl, hfr, hb, cls = pg.debug_far(l, r, b)
        imin, imax, off = diag_index(l.n_sites - 1, r.n_sites - 1, b)
        bad = np.nonzero(~same)[0]
        dd = np.searchsorted(off, bad, side="right") - 1
        def edges(g, s_):
            return [int(s_ - x) for x in g.bwd_src[g.bwd_off[s_]:g.bwd_off[s_ + 1]]]
        for c, d in list(zip(bad, dd))[:10]:
            i = int(imin[d] + c - off[d]); j = int(d - i)
            print("  cell d=%d i=%d j=%d (lo %d hi %d width %d) class %d flags %d | row edges %s col edges %s | ring %s pipe %s" %
                  (d, i, j, imin[d], imax[d], imax[d] - imin[d] + 1, cls[d], hb[d], edges(l, i), edges(r, j), sa[c], sb[c]))
        ud = np.unique(dd)
        print("  bad diagonals:", ud[:20], "... classes", np.bincount(cls[ud], minlength=6).tolist())
        # the wide run the first bad diagonal belongs to / follows
        d0 = int(ud[0])
        t = d0
        while t > 0 and cls[t] >= 4: t -= 1
        print("  first bad diagonal %d; class %d; classes around: %s" % (d0, cls[d0], cls[max(0, d0 - 8): d0 + 8].tolist()))
        break
