import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
import pagan2_msa_amd as pg
from pagan2_msa_amd import host
wl = sys.argv[1]
cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = bench.WORKLOADS[wl]
names, seqs, newick = bench.make_inputs(wl)
msa = host.Msa(names, seqs, newick, use_anchors=anchors).align()
by = {}
for k in range(msa.n_internal):
    by.setdefault(msa.node_info(k).level, []).append(k)
for lv in sorted(by):
    k = by[lv][0]
    l, r, m, b = msa.node_job(k)
    out = []
    for g in (l, r):
        off = np.asarray(g.bwd_off); ne = np.diff(off)[1:-1]; n = len(ne)
        src = np.asarray(g.bwd_src); lw = np.asarray(g.bwd_logw)
        first = off[1:-2]
        adj1 = (ne == 1)
        w1 = adj1 & (lw[np.minimum(first, len(lw)-1)] != 0)
        out.append((n, int((ne == 0).sum()), int(w1.sum()), int((ne == 2).sum()), int((ne >= 3).sum())))
    print("level", lv, "node", k, "sites/no-edge/one-weighted/two/three+:", out, pg.debug_route(l, r, m, b))
