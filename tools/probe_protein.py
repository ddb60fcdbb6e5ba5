"""Diagnostic: fill time of 32 plain 500-site alignments with a 211-state table (cfg3-shaped DP jobs; the
protein model itself is not built, the table is random) next to the same jobs with a 15-state table."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth

for states in (15, 211):
    jobs = []
    for k in range(32):
        left = synth.random_graph(500, states, 100 + k, p_extra=0.0)
        right = synth.random_graph(500, states, 200 + k, p_extra=0.0)
        jobs.append((left, right, synth.random_model(states, 3), None))
    b = pg.Batch(jobs)
    for rep in range(3):
        b.run(); b.sync()
    print("states %d: 32 jobs, cells %d, fill %.3f ms, trace %.3f ms" % ((states, b.cells) + tuple(b.last_ms())), flush=True)
    b.close()
