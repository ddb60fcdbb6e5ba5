"""Diagnostic: the cfg4 walk with PAGAN_DP_VERBOSE; the library's per-batch lines and the walk's own totals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pagan2_msa_amd import host
w = sys.argv[1] if len(sys.argv) > 1 else "cfg4_32x100kb_dna_anchored"
names, seqs, newick = bench.make_inputs(w)
for rep in range(2):
    t0 = time.perf_counter()
    msa = host.Msa(names, seqs, newick, use_anchors=bench.WORKLOADS[w][7])
    t1 = time.perf_counter()
    msa.align()
    t2 = time.perf_counter()
    print("walk %d: create (tree, leaves, model factory) %.1f ms, align %.1f ms; %s" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), msa.timing()), file=sys.stderr)
