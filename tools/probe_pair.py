import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host, abi
names, seqs, nwk = synth.evolve_balanced(2, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=5)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
print("timing", msa.timing())
l, r, m, b = msa.node_job(0)
for rep in range(2):
    batch = pg.Batch([(l, r, m, b)])
    for k in range(3):
        batch.run(); batch.sync()
        print("cells", batch.cells, "ms", batch.last_ms(), "nd", l.n_sites + r.n_sites - 3)
    batch.close()
