"""Diagnostic: band-width and edge-span statistics of the bench workload's node alignments
(decides the ring geometry of the fill kernel).  Needs the GPU (the tree walk aligns on it)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pagan2_msa_amd import synth, host

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
names, seqs, nwk = synth.evolve_balanced(leaves, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0,
                                         seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()


def site_stats(g):
    off = g.bwd_off.astype(np.int64)
    n = g.n_sites
    ne = off[1:] - off[:-1]
    span = np.zeros(n, np.int64)
    idx = np.repeat(np.arange(n), ne)
    np.maximum.at(span, idx, idx - g.bwd_src)
    simple = (ne == 1) & (span == 1) & (np.add.reduceat(np.abs(g.bwd_logw), off[:-1].clip(max=len(g.bwd_logw) - 1)) * (ne > 0) == 0)
    return ne, span, simple


for k in range(msa.n_internal):
    info = msa.node_info(k)
    l, r, m, b = msa.node_job(k)
    Lx, Ly = l.n_sites - 1, r.n_sites - 1
    up = np.maximum(b.upper[:Lx].astype(np.int64), 0)
    lw = np.minimum(b.lower[:Lx].astype(np.int64), Ly - 1)
    ii = np.arange(Lx)
    nd = Lx + Ly - 1
    d = np.arange(nd)
    imin = np.searchsorted(ii + lw, d, side="left")
    imax = np.searchsorted(ii + up, d, side="right") - 1
    w = imax - imin + 1
    neL, spL, siL = site_stats(l)
    neR, spR, siR = site_stats(r)
    line = "node %2d level %d nd %d cells %d meanw %.1f" % (k, info.level, nd, int(w.sum()), w.mean())
    line += " | w>64 %.3f >112 %.3f >176 %.3f >240 %.3f" % tuple((w > t).mean() for t in (64, 112, 176, 240))
    line += " | multiL %.4f multiR %.4f ne>2 %.5f" % ((~siL[1:Lx]).mean(), (~siR[1:Ly]).mean(), ((neL > 2).sum() + (neR > 2).sum()) / (Lx + Ly))
    jlo, jhi = d - imax, d - imin
    for name, flagL, flagR in (("nonsimple", ~siL, ~siR), ("ne>2", neL > 2, neR > 2), ("span>=8", spL >= 8, spR >= 8),
                               ("span>=16", spL >= 16, spR >= 16), ("span>=24", spL >= 24, spR >= 24),
                               ("span>=36", spL >= 36, spR >= 36), ("span>=60", spL >= 60, spR >= 60)):
        cl = np.concatenate([[0], np.cumsum(flagL[:Lx])])
        cr = np.concatenate([[0], np.cumsum(flagR[:Ly])])
        anyd = ((cl[imax + 1] - cl[imin]) + (cr[jhi + 1] - cr[jlo])) > 0
        line += " | d:%s %.4f" % (name, anyd.mean())
    print(line, flush=True)

# ---- shapes of the non-simple sites (what a register-carried skip-edge step would have to cover) ----
for k in (msa.n_internal - 1, msa.n_internal - 2, msa.n_internal // 2):
    l, r, m, b = msa.node_job(k)
    for side, g in (("L", l), ("R", r)):
        off = g.bwd_off.astype(np.int64); n = g.n_sites - 1
        ne = off[1:n + 1] - off[:n]
        two = np.nonzero(ne == 2)[0]
        d0 = two - g.bwd_src[off[two]]; d1 = two - g.bwd_src[off[two] + 1]
        w0 = g.bwd_logw[off[two]]; w1 = g.bwd_logw[off[two] + 1]
        one = np.nonzero(ne == 1)[0]
        one_d = one - g.bwd_src[off[one]]; one_w = g.bwd_logw[off[one]]
        print("node %d %s: sites %d | ne==1 %d (of which dist>1 %d, weight!=0 %d) | ne==2 %d: adjacent first %d, adjacent second %d, "
              "no adjacent %d, both weights 0 %d | ne>2 %d" % (k, side, n, one.size, int((one_d > 1).sum()), int((one_w != 0).sum()),
              two.size, int((d0 == 1).sum()), int(((d1 == 1) & (d0 != 1)).sum()), int(((d0 != 1) & (d1 != 1)).sum()),
              int(((w0 == 0) & (w1 == 0)).sum()), int((ne > 2).sum())), flush=True)
