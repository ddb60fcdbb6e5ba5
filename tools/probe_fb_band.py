"""Diagnostic: forward/backward sweeps of ONE long tunnel (two 100 kb leaves inside define_tunnel's band), kernel times by
group count; with a -DPG_FB_STATS build (PAGAN_DP_LIB=...) the kernel's own counters come out on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import host, synth
length = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
_, seqs, _ = synth.evolve_balanced(2, length, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240811)
gl, gr = (host.HGraph.leaf(s).flatten() for s in seqs)
band, _ = host.define_tunnel(seqs[0], seqs[1], seqs[0], seqs[1])
mp = host.model_prob(1, 0.02, base_freq=[0.25] * 4)
for groups in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "3", "4")):
    if groups == "ring":                       # (the default for plain sequences: the LDS-ring sweeps)
        os.environ.pop("PAGAN_FB_GROUPS", None)
    else:
        os.environ["PAGAN_FB_GROUPS"] = groups
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        fb = pg.FullProbability(gl, gr, mp, band)
        wall = time.perf_counter() - t0
        cur = (fb.forward_ms, fb.backward_ms, wall * 1e3, fb.cells)
        fb.close()
        best = cur if best is None or cur[0] < best[0] else best
    print("sites", gl.n_sites, gr.n_sites, "groups", groups, "fwd %.2f ms bwd %.2f ms wall %.1f ms cells %d" % best, flush=True)
