"""Diagnostic (GPU): one case of tests/diagnostics/sweep_parity.py through pg_fill_ring and pg_fill_pipe, cell-by-cell score
comparison: the first differing cells with their rows' position in the band, the sites' edges and the diagonal's class.
    python tools/dbg_sweep_case.py CASE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAGAN_DP_COMPACT"] = "0"
os.environ["PAGAN_DP_SCORE_CHECK"] = "0"
os.environ["PAGAN_DP_RERUN"] = "0"
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth
from dbg_pipe import diag_index

case = int(sys.argv[1])
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
for _ in range(int(rng.integers(0, 3))):
    a = int(rng.integers(10, max(11, Lx - 450))); rows = int(rng.integers(30, 440)); jump = int(rng.integers(30, 460))
    b = min(a + rows, Lx - 1)
    upper[a:b] = upper[a]; lower[a:b] = min(lower[b - 1] + jump, Ly - 1)
upper = np.maximum.accumulate(upper); lower = np.maximum.accumulate(lower)
upper[0] = 0; lower[-1] = Ly - 1
band = abi.Band(upper, lower)
model = synth.random_model(15, case)
flags = int(rng.choice([0, 0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN]))
job = (left, right, model, band)
cls, _ = pg.debug_plan(left, right, band)
print("case", case, "n", n, "span", span, "flags", flags, "classes", np.bincount(cls, minlength=6).tolist())
os.environ["PAGAN_DP_FILL"] = "ring"; A = pg.Batch([job], flags=flags) if flags else pg.Batch([job]); A.run(); A.sync(); sa = A.debug_scores(0)
os.environ["PAGAN_DP_FILL"] = "pipe"; B = pg.Batch([job], flags=flags) if flags else pg.Batch([job]); B.run(); B.sync(); sb = B.debug_scores(0)
same = (sa.view(np.int64) == sb.view(np.int64)).all(axis=1)
print("cells", same.size, "different", int((~same).sum()))
if not same.all():
    imin, imax, off = diag_index(Lx, Ly, band)
    bad = np.nonzero(~same)[0]
    dd = np.searchsorted(off, bad, side="right") - 1
    d4 = np.nonzero(cls == 4)[0]
    print("first class-4 diagonal", d4[0] if len(d4) else None, "last", d4[-1] if len(d4) else None)
    for c, d in list(zip(bad, dd))[:16]:
        i = int(imin[d] + c - off[d]); j = int(d - i)
        def edges(g, s):
            a, e = g.bwd_off[s], g.bwd_off[s + 1]
            return [int(s - x) for x in g.bwd_src[a:e]]
        print("cell d=%d i=%d j=%d class %d lo %d hi %d (i-lo %d, lane %d) | row edges %s | col edges %s | ring %s pipe %s" %
              (d, i, j, cls[d], imin[d], imax[d], i - imin[d], i % 448, edges(left, i), edges(right, j), sa[c], sb[c]))
    print("bad diagonals:", np.unique(dd)[:30])
    print("classes of the bad diagonals:", np.bincount(cls[np.unique(dd)], minlength=6).tolist())
