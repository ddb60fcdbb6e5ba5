"""Diagnostic: time the fill kernel of one saved node alignment with whatever library PAGAN_DP_LIB names -- also builds
whose results are wrong on purpose (timing experiments).  `save N` walks an N x 100 kb tree with the product library and
stores the root alignment's inputs under $TMPDIR/pagan_root_job.npz; `run` loads them and times the resident batch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

PATH = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pagan_root_job.npz")
if sys.argv[1] == "save":
    from pagan2_msa_amd import host, synth
    names, seqs, nwk = synth.evolve_balanced(int(sys.argv[2]), 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=5)
    msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
    l, r, m, b = msa.node_job(msa.n_internal - 1)
    np.savez(PATH, ls=l.state, lo=l.bwd_off, lsrc=l.bwd_src, lw=l.bwd_logw, le=l.bwd_eid, ln=l.n_edges,
             rs=r.state, ro=r.bwd_off, rsrc=r.bwd_src, rw=r.bwd_logw, re=r.bwd_eid, rn=r.n_edges,
             table=m.log_score, params=np.array(m.params), up=b.upper, low=b.lower)
else:
    import pagan2_msa_amd as pg
    from pagan2_msa_amd import abi
    d = np.load(PATH)
    l = abi.Graph(d["ls"], d["lo"], d["lsrc"], d["lw"], d["le"], n_edges=int(d["ln"]))
    r = abi.Graph(d["rs"], d["ro"], d["rsrc"], d["rw"], d["re"], n_edges=int(d["rn"]))
    m = abi.Model(d["table"], *d["params"])
    b = abi.Band(d["up"], d["low"])
    batch = pg.Batch([(l, r, m, b)])
    for rep in range(3):
        batch.run(); batch.sync()
        print("cells", batch.cells, "ms", batch.last_ms())
