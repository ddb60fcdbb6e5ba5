"""Diagnostic: what the planner makes of the largest node alignments of a workload -- class histogram of
the banded kernel's diagonals, width percentiles, bwd-edge counts per site (needs the GPU for the tree walk):
   python tools/probe_plan.py cfg5_512x10kb_dna_anchored 6"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_512x10kb_dna_anchored"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
leaves, length, branch, sub, indel, mean_len, anchors = bench.WORKLOADS[name]
names, seqs, nwk = synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel, mean_len=mean_len,
                                         seed=20240807 + int(name[3]))
msa = host.Msa(names, seqs, nwk, use_anchors=anchors).align()


def cells_of(job):
    left, right, _, band = job
    return pg.lib().pagan_dp_count_cells(left.n_sites, right.n_sites, C.byref(band.c) if band is not None else None)


sized = sorted(((cells_of(msa.node_job(k)), k) for k in range(msa.n_internal)), reverse=True)[:top]
for cells, k in sized:
    l, r, m, b = msa.node_job(k)
    cls, waves = pg.debug_plan(l, r, b)
    hist = np.bincount(cls, minlength=6)
    Lx, Ly = l.n_sites - 1, r.n_sites - 1
    up = np.maximum(b.upper[:Lx].astype(np.int64), 0)
    lo = np.minimum(b.lower[:Lx].astype(np.int64), Ly - 1)
    ii = np.arange(Lx)
    d = np.arange(Lx + Ly - 1)
    w = (np.searchsorted(ii + up, d, side="right") - 1) - np.searchsorted(ii + lo, d, side="left") + 1
    ne = np.diff(l.bwd_off.astype(np.int64))
    print("node %d level %d cells %d diagonals %d classes %s width p50/p90/p99/max %d/%d/%d/%d left sites with 1/2/3/4+ bwd edges %.1f/%.1f/%.1f/%.1f %%"
          % (k, msa.node_info(k).level, cells, len(cls), hist.tolist(), *np.percentile(w, [50, 90, 99, 100]).astype(int),
             *(100.0 * np.array([(ne == 1).mean(), (ne == 2).mean(), (ne == 3).mean(), (ne >= 4).mean()]))), flush=True)
