"""Diagnostic: a wider randomized parity sweep of the banded fill kernel than the test suite runs
(random graphs with long edges, random bands with boxes of varying size, option bits) against the oracle."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth
import oracle

oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
hist = np.zeros(6, np.int64)
for case in range(int(os.environ.get("PG_SWEEP_FIRST", "0")), n_cases):
    rng = np.random.default_rng(int(os.environ.get("PG_SWEEP_SEED", "1000")) + case)
    n = int(rng.integers(150, 1400))
    span = int(rng.choice([4, 8, 17, 19, 25, 40, 80]))
    p_extra = float(rng.choice([0.02, 0.08, 0.3]))
    left = synth.random_graph(n, 15, 3000 + case, p_extra=p_extra, max_deg=int(rng.integers(2, 5)), max_span=span, p_dead=float(rng.choice([0, 0, 0.01])))
    right = synth.random_graph(n + int(rng.integers(-40, 60)), 15, 4000 + case, p_extra=p_extra, max_deg=int(rng.integers(2, 5)), max_span=span)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    half = rng.integers(3, 60, Lx)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    upper = np.maximum.accumulate(np.maximum(centre - half, 0))
    lower = np.maximum.accumulate(np.minimum(centre + half, Ly - 1))
    for _ in range(int(rng.integers(0, 3))):                      # boxes: some narrower, some wider than the lanes / windows
        a = int(rng.integers(10, max(11, Lx - 450))); rows = int(rng.integers(30, 440)); jump = int(rng.integers(30, 460))
        b = min(a + rows, Lx - 1)
        upper[a:b] = upper[a]; lower[a:b] = min(lower[b - 1] + jump, Ly - 1)
    upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
    upper[0] = 0; lower[-1] = Ly - 1
    band = abi.Band(upper, lower)
    model = synth.random_model(15, case)
    flags = int(rng.choice([0, 0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN]))
    cls = pg.debug_far(left, right, band)[4] & 15                # the classes as the batch path plans them (far histories, third pass)
    hist += np.bincount(cls, minlength=6)
    want = oracle.dp_align(left, right, model, band, flags=flags)
    got = pg.align(left, right, model, band, flags=flags)
    ok = got.same_alignment(want)
    bad += not ok
    print("case %2d n %4d span %2d classes %s %s" % (case, n, span, np.bincount(cls, minlength=6).tolist(), "OK" if ok else "MISMATCH"), flush=True)
print("diagonals by class:", hist.tolist(), "mismatches:", bad)
