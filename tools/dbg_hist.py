"""Diagnostic (GPU): one banded job with far sites through pg_fill_ring and pg_fill_pipe (far histories on), cell-by-cell score
comparison; the first differing cells with their sites' flags and the diagonal's class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["PAGAN_DP_COMPACT"] = "0"
os.environ["PAGAN_DP_SCORE_CHECK"] = "0"
os.environ["PAGAN_DP_RERUN"] = "0"
import numpy as np
import pagan2_msa_amd as pg
from dbg_pipe import diag_index
from test_pipe_gpu import banded_job

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kw = dict(max_span=40 if seed < 2 else 8)
job = banded_job(seed, **kw)
l, r, m, b = job
n, hfl, hfr, hb, cls = pg.debug_far(l, r, b)
print("served", n, "classes", np.bincount(cls, minlength=6).tolist(), "H diagonals", int(hb.sum()))
os.environ["PAGAN_DP_FILL"] = "ring"; A = pg.Batch([job]); A.run(); A.sync(); sa = A.debug_scores(0)
os.environ["PAGAN_DP_FILL"] = "pipe"; B = pg.Batch([job]); B.run(); B.sync(); sb = B.debug_scores(0)
same = (sa.view(np.int64) == sb.view(np.int64)).all(axis=1)
print("cells", same.size, "different", int((~same).sum()))
if not same.all():
    imin, imax, off = diag_index(l.n_sites - 1, r.n_sites - 1, b)
    bad = np.nonzero(~same)[0]
    dd = np.searchsorted(off, bad, side="right") - 1
    for c, d in list(zip(bad, dd))[:12]:
        i = int(imin[d] + c - off[d]); j = int(d - i)
        def edges(g, s):
            a, e = g.bwd_off[s], g.bwd_off[s + 1]
            return [int(s - x) for x in g.bwd_src[a:e]]
        print("cell d=%d i=%d j=%d class %d H %d | row flag %02x edges %s | col flag %02x edges %s | ring %s pipe %s" %
              (d, i, j, cls[d], hb[d], hfl[i], edges(l, i), hfr[j], edges(r, j), sa[c], sb[c]))
    print("bad diagonals:", np.unique(dd)[:30])
    print("classes of the bad diagonals:", np.bincount(cls[np.unique(dd)], minlength=6).tolist())
