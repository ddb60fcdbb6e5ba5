"""Diagnostic: one job through pg_fill_ring and pg_fill_pipe, cell-by-cell score comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth

def diag_index(Lx, Ly, band):
    nd = Lx + Ly - 1
    d = np.arange(nd)
    if band is None:
        imin = np.maximum(0, d - (Ly - 1)); imax = np.minimum(d, Lx - 1)
    else:
        up = np.maximum(band.upper[:Lx].astype(np.int64), 0); lw = np.minimum(band.lower[:Lx].astype(np.int64), Ly - 1)
        ii = np.arange(Lx)
        imin = np.searchsorted(ii + lw, d, side="left"); imax = np.searchsorted(ii + up, d, side="right") - 1
    w = np.maximum(imax - imin + 1, 0)
    return imin, imax, np.concatenate([[0], np.cumsum(w)])

def compare(job, what):
    l, r, m, b = job
    os.environ["PAGAN_DP_FILL"] = "ring"; A = pg.Batch([job]); A.run(); A.sync(); sa = A.debug_scores(0)
    os.environ["PAGAN_DP_FILL"] = "pipe"; B = pg.Batch([job]); B.run(); B.sync(); sb = B.debug_scores(0)
    same = (sa.view(np.int64) == sb.view(np.int64)).all(axis=1)
    if same.all():
        print(what, "identical", sa.shape[0], "cells", flush=True); return
    imin, imax, off = diag_index(l.n_sites - 1, r.n_sites - 1, b)
    bad = np.nonzero(~same)[0]
    first = bad[0]; d = int(np.searchsorted(off, first, side="right") - 1); i = int(imin[d] + first - off[d])
    print(what, "DIFFERENT: %d of %d cells; first at cell %d: d=%d i=%d j=%d (diag lo %d hi %d)" %
          (bad.size, same.size, first, d, i, d - i, imin[d], imax[d]), "ring", sa[first], "pipe", sb[first], flush=True)
    dd = np.searchsorted(off, bad, side="right") - 1
    print("   bad diagonals:", np.unique(dd)[:20], "rows", np.unique(imin[dd] + bad - off[dd])[:20])

for length in [int(a) for a in sys.argv[1:]] or [21, 30, 64]:
    _, seqs, _ = synth.evolve_balanced(2, length, seed=1, sub=0.1, indel_start=0.03)
    job = (synth.chain_graph(seqs[0]), synth.chain_graph(seqs[1]), synth.jc_like_dna_model(0.1), None)
    compare(job, "len %d" % length)
