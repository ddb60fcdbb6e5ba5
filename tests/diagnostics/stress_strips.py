"""Diagnostic: batches of random full-matrix / wide-band alignments as row strips on the banded kernel (dp_pipe.hip,
strip_feeder) against the oracle -- several strips per job, several jobs per XCD, bwd edges reaching into other strips,
dead sites, every option of the assist waves:
    python tests/diagnostics/stress_strips.py [rounds]"""
import os
import sys
os.environ["PAGAN_DP_WIDE"] = "strips"
os.environ["PAGAN_DP_STRIP_SITES"] = "100000"      # (every wide job as strips, the assist waves' general code included)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth
import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(os.environ.get("PG_STRESS_SEED", "54321")))   # (PG_STRESS_SEED: another stream of batches)
bad = 0
for rd in range(rounds):
    jobs = []
    for k in range(int(rng.integers(1, 12))):
        nl, nr = int(rng.integers(200, 1800)), int(rng.integers(200, 1800))
        span = int(rng.choice([6, 30, 300]))
        left = synth.random_graph(nl, 15, int(rng.integers(1 << 30)), p_extra=0.1, max_deg=int(rng.integers(2, 6)), max_span=span, p_dead=float(rng.choice([0.0, 0.02, 0.3])))
        right = synth.random_graph(nr, 15, int(rng.integers(1 << 30)), p_extra=0.1, max_deg=int(rng.integers(2, 6)), max_span=span, p_dead=float(rng.choice([0.0, 0.02, 0.3])))
        band = None
        if rng.random() < 0.5:
            Lx, Ly = left.n_sites - 1, right.n_sites - 1
            centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
            h = rng.integers(200, 500, Lx)
            upper = np.maximum.accumulate(np.maximum(centre - h, 0)); lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
            upper[0] = 0; lower[-1] = Ly - 1
            band = abi.Band(upper, lower)
        jobs.append((left, right, synth.random_model(15, int(rng.integers(1 << 30))), band))
    got = pg.align_batch(jobs)
    for k, (l, r, m, b) in enumerate(jobs):
        want = oracle.dp_align(l, r, m, b)
        ok = got[k].status == want.status and np.float64(got[k].score).tobytes() == np.float64(want.score).tobytes() and \
            np.array_equal(got[k].cols, want.cols)
        bad += not ok
        print("round %d job %d %d x %d band %s: %s" % (rd, k, l.n_sites, r.n_sites, b is not None, "OK" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
