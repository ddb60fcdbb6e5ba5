"""Diagnostic (no oracle): the scores the row strips store against the tiled kernel's, cell by cell; reports the first
differing cells by (diagonal, row, column, strip).  python tools/dbg_strips.py <case> [flags]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth


def wide_band(Lx, Ly, half, seed):
    rng = np.random.default_rng(seed)
    centre = np.arange(Lx) * (Ly - 1) // max(Lx - 1, 1)
    h = rng.integers(half // 2, half, Lx)
    upper = np.maximum.accumulate(np.maximum(centre - h, 0))
    lower = np.maximum.accumulate(np.minimum(centre + h, Ly - 1))
    upper[0] = 0
    lower[-1] = Ly - 1
    return abi.Band(upper, lower)


def case(name):
    band = None
    if name == "three":
        left = synth.random_graph(460, 15, 101, p_extra=0.15, max_deg=5, max_span=7)
        right = synth.random_graph(300, 15, 202, p_extra=0.15, max_deg=5, max_span=7)
        model = synth.random_model(15, 7)
    elif name == "band":
        left = synth.random_graph(900, 15, 10, p_extra=0.08, max_deg=4, max_span=30)
        right = synth.random_graph(860, 15, 20, p_extra=0.08, max_deg=4, max_span=30)
        band = wide_band(left.n_sites - 1, right.n_sites - 1, 400, 0)
        model = synth.random_model(15, 0)
    elif name == "opt":
        left = synth.random_graph(400, 15, 5, p_extra=0.1, max_deg=3, max_span=9)
        right = synth.random_graph(370, 15, 6, p_extra=0.1, max_deg=3, max_span=9)
        model = synth.random_model(15, 3)
    elif name == "prot":
        left = synth.random_graph(460, 211, 101, p_extra=0.12, max_deg=4, max_span=20)
        right = synth.random_graph(430, 211, 202, p_extra=0.12, max_deg=4, max_span=20)
        model = synth.random_model(211, 7)
    elif name == "protplain":
        left = synth.random_graph(400, 211, 11, p_extra=0.0)
        right = synth.random_graph(380, 211, 12, p_extra=0.0)
        model = synth.random_model(211, 3)
    elif name == "skip":
        left = synth.random_graph(480, 15, 101, p_extra=0.10, max_deg=2, max_span=6)
        right = synth.random_graph(260, 15, 202, p_extra=0.10, max_deg=2, max_span=6)
        model = synth.random_model(15, 7)
    else:
        left = synth.random_graph(300, 15, 101)
        right = synth.random_graph(330, 15, 202)
        model = synth.random_model(15, 7)
    return left, right, model, band


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "three"
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    left, right, model, band = case(name)
    sc = {}
    for kernel in ("tiles", "strips"):
        os.environ["PAGAN_DP_WIDE"] = kernel
        os.environ["PAGAN_DP_STRIP_SITES"] = "100000"
        print(kernel, pg.debug_route(left, right, model, band))
        b = pg.Batch([(left, right, model, band)], flags=flags)
        b.run(); b.sync()
        sc[kernel] = b.debug_scores(0).reshape(-1, 3)
        try:
            print(kernel, "status", b.fetch()[0].status)
        except Exception as e:
            print(kernel, "fetch:", e)
        b.close()
    a, c = sc["tiles"].view(np.int64), sc["strips"].view(np.int64)
    bad = np.nonzero((a != c).any(axis=1))[0]
    print("cells", len(a), "differing", len(bad))
    if len(bad) == 0:
        return
    # cell index -> (d, row): rebuild the diagonal index
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    lo = np.zeros(Lx, int) if band is None else np.maximum(np.asarray(band.upper[:Lx]), 0)
    hi = np.full(Lx, Ly - 1) if band is None else np.minimum(np.asarray(band.lower[:Lx]), Ly - 1)
    nd = Lx + Ly - 1
    imin = np.zeros(nd, int); imax = np.zeros(nd, int); doff = np.zeros(nd + 1, int)
    a_, b_ = -1, 0
    for d in range(nd):
        while a_ + 1 < Lx and lo[a_ + 1] + a_ + 1 <= d:
            a_ += 1
        while b_ < Lx and hi[b_] + b_ < d:
            b_ += 1
        imin[d], imax[d] = b_, a_
        doff[d + 1] = doff[d] + max(0, a_ - b_ + 1)
    seen = 0
    for ix in bad[:12]:
        d = int(np.searchsorted(doff, ix, side="right") - 1)
        row = imin[d] + (ix - doff[d])
        print("cell %d: d %d row %d col %d strip %d (row %% 192 = %d)  tiles %s strips %s" % (ix, d, row, d - row, row // 192, row % 192, sc["tiles"][ix], sc["strips"][ix]))
    ds = sorted(set(int(np.searchsorted(doff, ix, side="right") - 1) for ix in bad))
    print("first differing diagonals", ds[:20], "of", len(ds))


main()
