"""Diagnostic: one 2 x 100 kb banded alignment; kernel times, and with a -DPG_STAMPS build of
the library (tools/build_stamps.sh) the per-path step counts and cycle shares of pg_fill_ring."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, host, abi

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 2
names, seqs, nwk = synth.evolve_balanced(leaves, 100000, branch=0.01, sub=0.008, indel_start=0.0008, mean_len=4.0, seed=5)
if os.environ.get("PG_ALIGN_RING"):
    os.environ["PAGAN_DP_FILL"] = "ring"      # tree walk on the reference kernel; only the timed batch uses the experiment
msa = host.Msa(names, seqs, nwk, use_anchors=1).align()
if os.environ.get("PG_ALIGN_RING"):
    os.environ["PAGAN_DP_FILL"] = "pipe"
if os.environ.get("PG_NOSTORE"):
    os.environ["PAGAN_DP_DEBUG_FLAGS"] = "0x100"
print("timing", msa.timing())
k = msa.n_internal - 1
l, r, m, b = msa.node_job(k)
batch = pg.Batch([(l, r, m, b)])
for rep in range(3):
    batch.run(); batch.sync()
    print("node", k, "cells", batch.cells, "ms", batch.last_ms(), "nd", l.n_sites + r.n_sites - 3)
if os.environ.get("PG_STAMPS"):
    n_int = 3 * (l.n_sites + r.n_sites - 2)
    raw = np.zeros(n_int, np.int32)
    pg.lib().pagan_batch_debug_trace(batch._h, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes)
    a = raw[n_int - 200:]
    names = ("flow control", "lds wait + dpp", "descriptor request + hand-over", "compute", "commit", "carry + prefetch", "loop edge")
    for w in range(4):
        n = max(int(a[12 * w]), 1)
        print("wave %d: %d class-0 steps with cells; cycles/step: " % (w, n) +
              ", ".join("%s %.0f" % (names[k], 16.0 * a[12 * w + 1 + k] / n) for k in range(7)))
if os.environ.get("PG_STAMPS"):
    b2 = raw[n_int - 400:]
    for w in range(4):
        n = b2[12 * w: 12 * w + 5].astype(np.int64); t = b2[12 * w + 5: 12 * w + 10].astype(np.int64) * 256
        print("wave %d steps with cells by class (n, Mcycles, cycles/step): " % w +
              "  ".join("c%d %d %.0fM %.0f" % (c, n[c], t[c] / 1e6, t[c] / max(n[c], 1)) for c in range(5)))
if os.environ.get("PG_STAMPS"):
    b3 = raw[n_int - 600:]
    for w in range(4):
        t = b3[4 * w: 4 * w + 4].astype(np.int64) * 256
        print("wave %d wide steps, Mcycles: loader and neighbour flags %.0f, shift + records + operands off the wide ring %.0f, L2 operands %.0f, arithmetic + stores + flag %.0f" % ((w,) + tuple(t / 1e6)))
if os.environ.get("PG_STAMPS"):
    b4 = raw[n_int - 800:]
    kinds = {0: "asm loop entries / diagonals run in it (M)", 1: "loader rows", 2: "loader cols", 3: "downstream (ring row reuse)", 4: "upstream (row above)", 5: "descriptor window / asm: exits after 48 looks at the upstream flag (count), ... downstream (M)",
             6: "far: all waves 8 steps behind / asm: looks at the upstream flag (count)", 7: "rendezvous / asm: exits with no row near the band (count)", 8: "assist wave (staged multi-edge candidates)", 9: "asm: upstream waits (count) / downstream waits (M = count / 1e6)"}
    for w in range(4):
        n = b4[20 * w: 20 * w + 10].astype(np.int64); t = b4[20 * w + 10: 20 * w + 20].astype(np.int64) * 256
        print("wave %d waits (count, Mcycles): " % w + "; ".join("%s %d %.0fM" % (kinds[k], n[k], t[k] / 1e6) for k in sorted(kinds)))
if os.environ.get("PG_STAMPS"):
    b5 = raw[n_int - 1000:]
    for a_ in range(3):
        n = max(int(b5[16 * a_]), 1)
        t = b5[16 * a_ + 1: 16 * a_ + 13].astype(np.int64) * 256 / n
        print("assist %d: %d diagonals; cycles/diagonal: prepare (descriptors, loader) %.0f, wait for compute waves %.0f, compute %.0f, publish %.0f; "
              "inside prepare: scan %.0f, batch %.0f, decode %.0f, far: pool writes %.0f, polls for landed cells %.0f, addresses %.0f, L2 loads %.0f, poll for the descriptor window %.0f" %
              ((a_, n) + tuple(t[:12])))
        print("assist %d: %d of its diagonals went to the general code whole, %d had cells staged by it; %d passes held two diagonals; passes sent there for: slots %d, a site's shape %d, "
              "the other side %d, the cell's shape or an edge from site 0 %d, a recent operand off the ring %d, the pool %d" %
              ((a_, int(b5[16 * a_ + 13]), int(b5[16 * a_ + 15]), int(b5[16 * a_ + 14])) + tuple(int(x) for x in b5[16 * a_ + 3: 16 * a_ + 9])))
if os.environ.get("PG_STAMPS"):
    b6 = raw[n_int - 1100: n_int - 1092].astype(np.int64)
    print("waves 0-3 compute, 4-6 assist, 7 loader: (wave slot, SIMD, CU) =", [(int(x & 15), int((x >> 4) & 3), int((x >> 8) & 15)) for x in b6])
if os.environ.get("PG_STAMPS") and os.environ.get("PG_RUNS_OUT"):
    r0 = (n_int // 4 * 3 + 15) & ~15
    cap = (n_int - 1200 - r0 - 16) // 16
    cnt = raw[r0: r0 + 4].copy()
    recs = raw[r0 + 16: r0 + 16 + 16 * cap].reshape(cap, 4, 4)
    cls_, _w = pg.debug_plan(l, r, b)
    np.savez_compressed(os.environ["PG_RUNS_OUT"], cnt=cnt, recs=recs[: int(min(cap, cnt.max()))], cls=cls_)
    print("runs of the asm loop per wave:", cnt.tolist(), "capacity", cap)
if os.environ.get("PG_CHECK"):
    import oracle
    bad = 0
    for kk in range(msa.n_internal):
        a, b2, m2, bb = msa.node_job(kk)
        want = oracle.dp_align(a, b2, m2, bb)
        ok = msa.node_result(kk).same_alignment(want)
        bad += not ok
        print("node", kk, "level", msa.node_info(kk).level, "cells", want.cells, "OK" if ok else "MISMATCH", flush=True)
    print("mismatches:", bad)
