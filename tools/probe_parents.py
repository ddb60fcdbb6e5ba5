"""Diagnostic: the cfg4 walk with PAGAN_DP_VERBOSE, host vs device parents (per-node timings on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pagan2_msa_amd import host
names, seqs, newick = bench.make_inputs("cfg4_32x100kb_dna_anchored")
for mode in ("host", "device", "device"):
    os.environ["PAGAN_PARENTS"] = mode
    t0 = time.perf_counter()
    msa = host.Msa(names, seqs, newick, use_anchors=1).align()
    print(mode, "walk %.3f s" % (time.perf_counter() - t0), msa.timing(), file=sys.stderr)
