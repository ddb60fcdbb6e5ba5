"""Diagnostic: per-node kernel times of the largest node alignments of a bench workload
   (python tools/probe_nodes.py cfg5_512x10kb_dna_anchored 12); PAGAN_DP_FILL=tiles / ring and
   PAGAN_DP_WIDE=wavefront select the other fill kernels."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_512x10kb_dna_anchored"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
leaves, length, branch, sub, indel, mean_len, anchors = bench.WORKLOADS[name]
seed = 20240807 + int(name[3]) if name.startswith("cfg") else 20240807
names, seqs, nwk = synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel, mean_len=mean_len, seed=seed)
msa = host.Msa(names, seqs, nwk, use_anchors=anchors).align()
def cells_of(job):
    left, right, _, band = job
    return pg.lib().pagan_dp_count_cells(left.n_sites, right.n_sites, C.byref(band.c) if band is not None else None)


sized = sorted(((cells_of(msa.node_job(k)), k) for k in range(msa.n_internal)), reverse=True)[:top]
for cells, k in sized:
    job = msa.node_job(k)
    b = pg.Batch([job])
    for rep in range(2):
        b.run(); b.sync()
    l, r = job[0], job[1]
    nd = l.n_sites + r.n_sites - 3
    print("node %d level %d cells %d diagonals %d mean width %.0f fill %.2f ms trace %.2f ms" %
          ((k, msa.node_info(k).level, cells, nd, cells / nd) + tuple(b.last_ms())), flush=True)
    b.close()
