import sys, time, os
sys.path.insert(0, os.getcwd())
import bench
from pagan2_msa_amd import host
names, seqs, newick = bench.make_inputs("cfg4_32x100kb_dna_anchored")
for rep in range(2):
    t0 = time.time()
    msa = host.Msa(names, seqs, newick, use_anchors=1)
    t1 = time.time()
    msa.align()
    t2 = time.time()
    print("construct %.3f align %.3f" % (t1 - t0, t2 - t1), msa.timing(), flush=True)
