"""Diagnostic: kernel times of single unbanded node alignments (first and last node of a 16 x 2 kb tree,
or of the tree given as `leaves length`); set PAGAN_DP_WIDE=wavefront to time the one-workgroup kernel."""
import ctypes as C
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 16
length = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
names, seqs, nwk = synth.evolve_balanced(leaves, length, branch=0.05, sub=0.04, indel_start=0.004, mean_len=4.0, seed=20240807 + 2)
msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
for k in (0, msa.n_internal - 1):
    job = msa.node_job(k)
    b = pg.Batch([job])
    if os.environ.get("PG_STAMPS"):
        pg.lib().pagan_batch_debug_poison(b._h)      # the tile kernel's counters (trace buffer's tail) start at -1
    for rep in range(1 if os.environ.get("PG_STAMPS") else 3):
        b.run(); b.sync()
    if os.environ.get("PG_STAMPS"):
        l, r = job[0], job[1]
        n_int = 3 * (l.n_sites + r.n_sites - 2)
        raw = np.zeros(n_int, np.int32)
        pg.lib().pagan_batch_debug_trace(b._h, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes)
        a = raw[(n_int - 64) & ~1:][:18].view(np.uint64)
        a = (a + np.uint64(1)).astype(np.float64)
        tiles = max(a[0], 1.0)
        print("tiles %d: cycles per tile %.0f, prologue %.0f; steps per tile simple %.1f near %.1f general %.1f; cycles per step %.0f %.0f %.0f"
              % (a[0], a[8] / tiles, a[1] / tiles, a[2] / tiles, a[3] / tiles, a[4] / tiles,
                 a[5] / max(a[2], 1), a[6] / max(a[3], 1), a[7] / max(a[4], 1)))
    print("node", k, "level", msa.node_info(k).level, "cells", b.cells, "ms", b.last_ms(), flush=True)
    b.close()
