"""Diagnostic: a randomized sweep of the forward/backward sweeps' three schedules for tunnels (LDS-ring sweeps, block schedule,
one-workgroup kernels) against the oracle: leaf pairs and graph pairs of random lengths behind random tunnels with boxes (rows whose
band jumps wider and narrower than a workgroup / a block).  Totals to 1e-9 for every case, both matrices cell by cell for the
shorter ones.  Usage: sweep_fb.py [cases] (PG_SWEEP_SEED: another seed)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, host, synth
import oracle

oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(os.environ.get("PG_SWEEP_SEED", "5000"))
TOL = 1e-9
bad = 0
used = {"ring": 0, "blocks": 0, "one": 0}


def close_logs(a, b):
    fa, fb = np.isfinite(a), np.isfinite(b)
    return np.array_equal(fa, fb) and np.allclose(a[fa], b[fb], rtol=TOL, atol=TOL)


def tunnel(rng, Lx, Ly):
    half = rng.integers(3, int(rng.choice([10, 40, 90, 200])), Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    for _ in range(int(rng.integers(0, 3))):                      # boxes
        a = int(rng.integers(5, max(6, Lx - 300))); rows = int(rng.integers(20, 300)); jump = int(rng.integers(20, 400))
        b = min(a + rows, Lx - 1)
        upper[a:b] = upper[a]; lower[a:b] = min(lower[b - 1] + jump, Ly - 1)
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    upper[0] = 0; lower[-1] = Ly - 1
    return abi.Band(upper.astype(np.int32), lower.astype(np.int32))


for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    length = int(rng.integers(120, 2500))
    kind = case % 3                                               # 0, 1: leaf pairs; 2: a graph pair (the root of a 4-leaf tree)
    if kind < 2:
        _, seqs, _ = synth.evolve_balanced(2, length, branch=0.03, sub=0.04, indel_start=0.008, mean_len=4, seed=seed0 + case)
        left, right = (host.HGraph.leaf(s).flatten() for s in seqs)
        dist = 0.06
    else:
        names, seqs, nwk = synth.evolve_balanced(4, min(length, 900), branch=0.03, sub=0.04, indel_start=0.01, mean_len=4, seed=seed0 + case)
        msa = host.Msa(names, seqs, nwk, use_anchors=0).align()
        left, right, _m, _b = msa.node_job(msa.n_internal - 1)
        dist = msa.node_info(msa.n_internal - 1).dist
    mp = host.model_prob(1, dist, base_freq=[0.3, 0.2, 0.2, 0.3])
    band = tunnel(rng, left.n_sites - 1, right.n_sites - 1) if rng.random() < 0.9 else None
    small = (left.n_sites - 1) * (right.n_sites - 1) <= 1_200_000
    lf, lb, post, logf = oracle.fb(left, right, mp, band=band, matrices=small)
    for mode, env in (("ring", {"PAGAN_FB_RING_MIN_ND": "0", "PAGAN_FB_BAND_MIN_ND": "0"}),
                      ("blocks", {"PAGAN_FB_RING": "0", "PAGAN_FB_BAND_MIN_ND": "0"}),
                      ("one", {"PAGAN_FB_RING": "0", "PAGAN_FB_BAND_MIN_ND": "off"})):
        for k in ("PAGAN_FB_RING_MIN_ND", "PAGAN_FB_BAND_MIN_ND", "PAGAN_FB_RING"):
            os.environ.pop(k, None)
        os.environ.update(env)
        fb = pg.FullProbability(left, right, mp, band)
        took = "ring" if fb.groups == 0 else ("blocks" if fb.groups > 1 else "one")
        used[took] += 1
        ok = abs(fb.log_fwd - lf) <= TOL * max(1, abs(lf)) and abs(fb.log_bwd - lb) <= TOL * max(1, abs(lb))
        if ok and small:
            ok = close_logs(fb.log_forward(), logf) and np.allclose(fb.posterior(), post, rtol=1e-7, atol=1e-12)
        fb.close()
        if not ok:
            bad += 1
            print("MISMATCH case", case, "mode", mode, "took", took, "sites", left.n_sites, right.n_sites, "band", band is not None, flush=True)
    if case % 10 == 9:
        print("case", case + 1, "of", n_cases, "schedules taken", used, "mismatches so far", bad, flush=True)
print("cases", n_cases, "seed", seed0, "schedules taken", used, "mismatches:", bad)
sys.exit(1 if bad else 0)
