"""Diagnostic: why the diagonals of the bench workload's upper nodes are class 2 (dp_abi.hip, classify_diagonals):
a site that is not 'easy' (more than two edges / no previous-site edge) or operands beyond the ring's reach.  Needs the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host

names, seqs, nwk = synth.evolve_balanced(32, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
REACH = 15


def feats(g, n):
    off = g.bwd_off.astype(np.int64)
    ne = off[1:n + 1] - off[:n]
    idx = np.repeat(np.arange(n), ne)
    dist = idx - g.bwd_src[:off[n]]
    span = np.zeros(n, np.int64); np.maximum.at(span, idx, dist)
    nadj = np.zeros(n, np.int64); np.add.at(nadj, idx, (dist == 1).astype(np.int64))
    easy = (np.arange(n) > 0) & ((ne == 1) | (ne == 2)) & (nadj == 1)
    three = (ne == 3) & (nadj == 1)
    return ne, span, easy, three


for k in (msa.n_internal - 1, msa.n_internal - 2, 21, 5):
    l, r, m, b = msa.node_job(k)
    Lx, Ly = l.n_sites - 1, r.n_sites - 1
    cls, _ = pg.debug_plan(l, r, b)
    up = np.maximum(b.upper[:Lx].astype(np.int64), 0); lw = np.minimum(b.lower[:Lx].astype(np.int64), Ly - 1)
    ii = np.arange(Lx); nd = Lx + Ly - 1; d = np.arange(nd)
    imin = np.searchsorted(ii + lw, d, side="left"); imax = np.searchsorted(ii + up, d, side="right") - 1
    neL, spL, eL, tL = feats(l, Lx); neR, spR, eR, tR = feats(r, Ly)
    def anyd(fl, fr):
        cl = np.concatenate([[0], np.cumsum(fl)]); cr = np.concatenate([[0], np.cumsum(fr)])
        jlo, jhi = d - imax, d - imin
        return ((cl[imax + 1] - cl[imin]) + (cr[jhi + 1] - cr[jlo])) > 0
    c2 = cls == 2
    noteasy = anyd(~eL & (neL > 0), ~eR & (neR > 0))
    three_only = anyd(tL, tR)
    worse = anyd(~eL & ~tL & (neL > 0), ~eR & ~tR & (neR > 0))
    far14 = anyd(spL >= REACH - 1, spR >= REACH - 1)
    print("node %d level %d: nd %d | class counts %s | class 2: %d; of them not-easy site %d (three-edge-with-adjacent only %d, worse %d), span>=14 site %d, neither (pair sums) %d"
          % (k, msa.node_info(k).level, nd, np.bincount(cls, minlength=6).tolist(), c2.sum(), (c2 & noteasy).sum(),
             (c2 & three_only & ~worse).sum(), (c2 & worse).sum(), (c2 & far14).sum(), (c2 & ~noteasy & ~far14).sum()), flush=True)
    for side, g, n in (("L", l, Lx), ("R", r, Ly)):
        ne, sp, e, t = feats(g, n)
        print("   %s: sites %d ne==2 %d ne==3 %d ne>3 %d | span>=14: %d, span>=8: %d" % (side, n, (ne == 2).sum(), (ne == 3).sum(), (ne > 3).sum(), (sp >= 14).sum(), (sp >= 8).sum()))
