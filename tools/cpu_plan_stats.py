"""Diagnostic, no GPU: walks the bench workload (cfg4 by default) with the oracle's DP behind the test seam
(pagan_msa_set_batch_backend), then reports what the banded kernel's planner (pagan_dp_debug_plan) makes of the upper
nodes: diagonals per class, band widths, and the shapes / spans of the multi-edge sites the classes hinge on.
    python tools/cpu_plan_stats.py [leaves] [length] [nodes...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host
import oracle

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
length = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
oracle.build()
OL = oracle.lib()


def backend(n, jobs, opts, out, user):
    for k in range(n):
        j = jobs[k]
        rc = OL.oracle_dp_align(j.left, j.right, j.model, j.band if j.band else None, opts, C.byref(out[k]))
        if rc != 0:
            return rc
    return 0


names, seqs, nwk = synth.evolve_balanced(leaves, length, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1)
msa.set_batch_backend(backend)
t0 = time.time()
msa.align()
print("walk on the CPU: %.1f s, %d internal nodes" % (time.time() - t0, msa.n_internal), flush=True)


def feats(g, n):
    off = g.bwd_off.astype(np.int64)
    ne = off[1:n + 1] - off[:n]
    idx = np.repeat(np.arange(n), ne)
    dist = idx - g.bwd_src[:off[n]]
    span = np.zeros(n, np.int64)
    np.maximum.at(span, idx, dist)
    nadj = np.zeros(n, np.int64)
    np.add.at(nadj, idx, (dist == 1).astype(np.int64))
    return ne, span, nadj


nodes = [int(a) for a in sys.argv[3:]] or [msa.n_internal - 1, msa.n_internal - 2, msa.n_internal - 4, msa.n_internal - 8, 0]
for k in nodes:
    l, r, m, b = msa.node_job(k)
    Lx, Ly = l.n_sites - 1, r.n_sites - 1
    cls, _ = pg.debug_plan(l, r, b)
    n_served, _hfl, _hfr, hb_, cls_far = pg.debug_far(l, r, b)       # the batch path's plan: with the far histories and the third pass
    print("node %d level %d: classes as planned for the kernel %s | served far sites %d, diagonals with a history reader / writer %d, with a third pass %d"
          % (k, msa.node_info(k).level, np.bincount(cls_far & 15, minlength=6).tolist(), n_served, int((hb_ & 1).sum()), int(((hb_ >> 1) & 1).sum())), flush=True)
    up = np.maximum(b.upper[:Lx].astype(np.int64), 0)
    lw = np.minimum(b.lower[:Lx].astype(np.int64), Ly - 1)
    ii = np.arange(Lx)
    nd = Lx + Ly - 1
    d = np.arange(nd)
    imin = np.searchsorted(ii + lw, d, side="left")
    imax = np.searchsorted(ii + up, d, side="right") - 1
    w = imax - imin + 1
    print("node %d level %d: nd %d cells %d | classes without far histories / third pass (pagan_dp_debug_plan) %s | width mean %.1f p50 %d p90 %d p99 %d max %d | >241: %d, >352: %d"
          % (k, msa.node_info(k).level, nd, int(w.sum()), np.bincount(cls & 15, minlength=6).tolist(), w.mean(),
             np.percentile(w, 50), np.percentile(w, 90), np.percentile(w, 99), w.max(), (w > 241).sum(), (w > 352).sum()), flush=True)
    jlo, jhi = d - imax, d - imin
    for side, g, n in (("L", l, Lx), ("R", r, Ly)):
        ne, sp, nadj = feats(g, n)
        print("   %s: sites %d | ne 1/2/3/4+: %d %d %d %d | no adjacent edge %d | span>=8 %d >=14 %d >=18 %d >=22 %d >=30 %d >=46 %d max %d"
              % (side, n, (ne == 1).sum(), (ne == 2).sum(), (ne == 3).sum(), (ne > 3).sum(), ((nadj == 0) & (ne > 0)).sum(),
                 (sp >= 8).sum(), (sp >= 14).sum(), (sp >= 18).sum(), (sp >= 22).sum(), (sp >= 30).sum(), (sp >= 46).sum(), sp.max()))
    neL, spL, adL = feats(l, Lx)
    neR, spR, adR = feats(r, Ly)

    def anyd(fl, fr):
        cl = np.concatenate([[0], np.cumsum(fl)])
        cr = np.concatenate([[0], np.cumsum(fr)])
        return ((cl[imax + 1] - cl[imin]) + (cr[jhi + 1] - cr[jlo])) > 0

    narrow = w <= 241
    line = "   diagonals (narrow ones) holding a site with"
    for name, fl, fr in (("ne>=3", neL >= 3, neR >= 3), ("ne>=4", neL >= 4, neR >= 4), ("no adjacent", (adL == 0) & (neL > 0), (adR == 0) & (neR > 0)),
                         ("span>=14", spL >= 14, spR >= 14), ("span>=18", spL >= 18, spR >= 18), ("span>=22", spL >= 22, spR >= 22),
                         ("span>=30", spL >= 30, spR >= 30), ("span>=46", spL >= 46, spR >= 46)):
        line += " | %s %d" % (name, (anyd(fl, fr) & narrow).sum())
    print(line, flush=True)
    # multi-edge cells per diagonal: how many lanes of a step have work
    multiL = ~((neL == 1) & (spL == 1))
    multiR = ~((neR == 1) & (spR == 1))
    cl = np.concatenate([[0], np.cumsum(multiL)])
    cr = np.concatenate([[0], np.cumsum(multiR)])
    nm = (cl[imax + 1] - cl[imin]) + (cr[jhi + 1] - cr[jlo])
    print("   multi-edge sites per diagonal: mean %.1f p50 %d p90 %d max %d" % (nm.mean(), np.percentile(nm, 50), np.percentile(nm, 90), nm.max()), flush=True)
