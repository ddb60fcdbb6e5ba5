"""Diagnostic (no oracle): one node of a bench workload as row strips -- kernel times, and with the statistics build
(tools/build_stamps.sh, PAGAN_DP_LIB) the step counts / cycles by class, the waits and the assist waves' counters of ONE strip
(PAGAN_DP_DEBUG_FLAGS bits 12-15 name it).   python tools/probe_strips.py <workload> <node | -1 = root> <strip>"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_16x2kb_dna_full"
node = int(sys.argv[2]) if len(sys.argv) > 2 else -1
strip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["PAGAN_DP_DEBUG_FLAGS"] = hex(strip << 12)
import numpy as np
import bench
import pagan2_msa_amd as pg
from pagan2_msa_amd import host

cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = bench.WORKLOADS[workload]
names, seqs, newick = bench.make_inputs(workload)
msa = host.Msa(names, seqs, newick, use_anchors=anchors).align()
k = msa.n_internal - 1 if node < 0 else node
l, r, m, b = msa.node_job(k)
info = msa.node_info(k)
print("node", k, "level", info.level, "sites", l.n_sites, r.n_sites, "cells", info.cells)
for kernel in ("tiles", "strips"):
    os.environ["PAGAN_DP_WIDE"] = kernel
    print(kernel, pg.debug_route(l, r, m, b))
    batch = pg.Batch([(l, r, m, b)])
    for rep in range(3):
        batch.run(); batch.sync()
    print(kernel, "ms (fill, trace)", batch.last_ms())
    if kernel == "tiles":
        batch.close()
stamps = "stats" in (os.environ.get("PAGAN_DP_LIB") or "")
if stamps:
    n_int = 3 * (l.n_sites + r.n_sites - 2)
    raw = np.zeros(n_int, np.int32)
    pg.lib().pagan_batch_debug_trace(batch._h, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes)
    b2 = raw[n_int - 400:]
    for w in range(4):
        n = b2[12 * w: 12 * w + 5].astype(np.int64); t = b2[12 * w + 5: 12 * w + 10].astype(np.int64) * 256
        print("wave %d steps with cells by class (n, Mcycles, cycles/step): " % w +
              "  ".join("c%d %d %.1fM %.0f" % (c, n[c], t[c] / 1e6, t[c] / max(n[c], 1)) for c in range(5)))
    b4 = raw[n_int - 800:]
    kinds = {0: "asm loop entries / diagonals run in it (M)", 1: "loader rows", 2: "loader cols", 3: "downstream (ring row reuse)", 4: "upstream (row above)", 5: "descriptor window / asm: exits after 48 looks at the upstream flag (count), ... downstream (M)",
             6: "far: all waves 8 steps behind / asm: looks at the upstream flag (count)", 7: "rendezvous / asm: exits with no row near the band (count)", 8: "assist wave (staged multi-edge candidates)", 9: "asm: upstream waits (count) / downstream waits (M = count / 1e6)"}
    for w in range(4):
        n = b4[20 * w: 20 * w + 10].astype(np.int64); t = b4[20 * w + 10: 20 * w + 20].astype(np.int64) * 256
        print("wave %d waits (count, Mcycles): " % w + "; ".join("%s %d %.2fM" % (kinds[kk], n[kk], t[kk] / 1e6) for kk in sorted(kinds)))
    b5 = raw[n_int - 1000:]
    for a_ in range(3):
        n = max(int(b5[16 * a_]), 1)
        t = b5[16 * a_ + 1: 16 * a_ + 13].astype(np.int64) * 256 / n
        print("assist %d: %d diagonals; cycles/diagonal by bucket 0..11:" % (a_, n), " ".join("%.0f" % x for x in t[:12]))
        print("assist %d: %d of its diagonals went to the general code whole, %d had cells staged by it; %d passes held two diagonals; passes sent there for: slots %d, a site's shape %d, "
              "the other side %d, the cell's shape or an edge from site 0 %d, a recent operand off the ring %d, the pool %d" %
              ((a_, int(b5[16 * a_ + 13]), int(b5[16 * a_ + 15]), int(b5[16 * a_ + 14])) + tuple(int(x) for x in b5[16 * a_ + 3: 16 * a_ + 9])))
batch.close()
