"""Diagnostic: forward/backward sweeps of ONE pair (two 2 kb leaves / the cfg2 root pair), kernel times by group count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pagan2_msa_amd as pg
from pagan2_msa_amd import host
names, seqs, newick = bench.make_inputs("cfg2_16x2kb_dna_full")
msa = host.Msa(names, seqs, newick, use_anchors=0).align()
bf = np.array([sum(sq.count(x) for sq in seqs) for x in "ACGT"], np.float32); bf /= bf.sum()
for k in (0, msa.n_internal - 1):
    left, right, _m, band = msa.node_job(k)
    mp = host.model_prob(1, msa.node_info(k).dist, base_freq=bf)
    for groups in ("1", "8", "16", "32"):
        os.environ["PAGAN_FB_GROUPS"] = groups
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            fb = pg.FullProbability(left, right, mp, band)
            wall = time.perf_counter() - t0
            cur = (fb.forward_ms, fb.backward_ms, wall * 1e3)
            fb.close()
            best = cur if best is None or cur[0] < best[0] else best
        print("node", k, "sites", left.n_sites, right.n_sites, "groups", groups, "fwd %.2f ms bwd %.2f ms wall %.1f ms" % best, flush=True)
