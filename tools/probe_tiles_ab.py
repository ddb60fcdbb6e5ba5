"""Diagnostic: fill time of single tiled jobs under the loaded library (PAGAN_DP_LIB).  The jobs come from a walk with the
PRODUCT library in a child process (a timing variant gives wrong results and cannot walk a tree): `--dump` writes them to
/tmp/pg_jobs.npz, the default mode loads them."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
WANT = (("cfg2_16x2kb_dna_full", (0, 8, 12, 14)),) + ((("cfg5_512x10kb_dna_anchored", (510, 509, 500)),) if not os.environ.get("PROBE_SMALL") else ())
if "--dump" in sys.argv:
    import bench
    from pagan2_msa_amd import host
    out = {}
    for w, nodes in WANT:
        names, seqs, newick = bench.make_inputs(w)
        msa = host.Msa(names, seqs, newick, use_anchors=bench.WORKLOADS[w][7]).align()
        for k in nodes:
            left, right, model, band = msa.node_job(k)
            key = "%s/%d/" % (w, k)
            for side, g in (("l", left), ("r", right)):
                for f in ("state", "bwd_off", "bwd_src", "bwd_logw", "bwd_eid"):
                    out[key + side + f] = getattr(g, f)
                out[key + side + "ne"] = np.array([g.n_edges])
            out[key + "table"] = model.log_score
            out[key + "params"] = np.array(model.params, np.float32)
            if band is not None:
                out[key + "up"] = band.upper; out[key + "lo"] = band.lower
            out[key + "info"] = np.array([msa.node_info(k).level, msa.node_info(k).cells], np.int64)
    np.savez("/tmp/pg_jobs.npz", **out)
    sys.exit(0)
env = dict(os.environ); env.pop("PAGAN_DP_LIB", None)
subprocess.run([sys.executable, os.path.abspath(__file__), "--dump"], check=True, env=env)
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi
Z = np.load("/tmp/pg_jobs.npz")
for w, nodes in WANT:
    for k in nodes:
        key = "%s/%d/" % (w, k)
        gs = [abi.Graph(Z[key + s + "state"], Z[key + s + "bwd_off"], Z[key + s + "bwd_src"], Z[key + s + "bwd_logw"], Z[key + s + "bwd_eid"], int(Z[key + s + "ne"][0])) for s in "lr"]
        model = abi.Model(Z[key + "table"], *[float(x) for x in Z[key + "params"]])
        band = abi.Band(Z[key + "up"], Z[key + "lo"]) if key + "up" in Z else None
        b = pg.Batch([(gs[0], gs[1], model, band)])
        best = 1e9
        for rep in range(4):
            b.run(); b.sync()
            best = min(best, b.last_ms()[0])
        print(os.path.basename(pg.LIB_PATH), os.environ.get("PAGAN_DP_TILES", ""), w, "node", k, "level", int(Z[key + "info"][0]), "cells %.3g" % Z[key + "info"][1], "fill %.3f ms" % best, flush=True)
        b.close()
