"""Diagnostic: host phases of every level's batch during the tree walk of a workload (PAGAN_DP_VERBOSE):
   python tools/probe_e2e.py cfg4_32x100kb_dna_anchored"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAGAN_DP_VERBOSE"] = "1"
from pagan2_msa_amd import synth, host
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_32x100kb_dna_anchored"
leaves, length, branch, sub, indel, mean_len, anchors = bench.WORKLOADS[name]
names, seqs, nwk = synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel, mean_len=mean_len,
                                         seed=20240807 + int(name[3]))
for rep in range(2):
    t0 = time.time()
    msa = host.Msa(names, seqs, nwk, use_anchors=anchors).align()
    print("wall %.3f s" % (time.time() - t0), msa.timing(), flush=True)
