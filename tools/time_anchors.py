import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pagan2_msa_amd import host, synth
import pagan2_msa_amd as pg
for length in (3000, 100000):
    names, seqs, _ = synth.evolve_balanced(2, length, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4, seed=3)
    a, b = seqs
    for mode in ("host", "device"):
        if mode == "host": os.environ["PAGAN_ANCHORS"] = "host"
        else: os.environ.pop("PAGAN_ANCHORS", None)
        host.prefix_hits(a, b, 30)
        t0 = time.perf_counter()
        for _ in range(5): h = host.prefix_hits(a, b, 30)
        print(length, mode, "%.2f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3), len(h), "hits")
