"""Timing experiments (GPU): level-by-level fill times of the headline workload for whatever library PAGAN_DP_LIB names --
including experiment builds with WRONG RESULTS (tools/build_exp.sh): the tree walk that produces the node jobs runs on the
older ring kernel (PAGAN_DP_FILL=ring, untouched by the experiments), only the timed level batches use the banded kernel.
    PAGAN_DP_LIB=$PWD/pagan2-msa_amd/libpagan_dp_exp_b.so python tools/exp_levels.py [leaves] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAGAN_DP_FILL"] = "ring"
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names, seqs, nwk = synth.evolve_balanced(leaves, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=20240807 + 4)
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
os.environ["PAGAN_DP_FILL"] = "pipe"
os.environ.setdefault("PAGAN_DP_RERUN", "0")
levels = {}
for k in range(msa.n_internal):
    levels.setdefault(msa.node_info(k).level, []).append(k)
out = []
for lv in sorted(levels):
    jobs = [msa.node_job(k) for k in levels[lv]]
    b = pg.Batch(jobs)
    best = 1e9
    for rep in range(reps):
        b.run(); b.sync()
        best = min(best, b.last_ms()[0])
    out.append(best)
    b.close()
print(os.path.basename(os.environ.get("PAGAN_DP_LIB", "libpagan_dp.so")), "fill ms by level", [round(x, 1) for x in out], "sum %.1f" % sum(out), flush=True)
