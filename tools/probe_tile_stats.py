"""Diagnostic (stats build: tools/build_stamps.sh, PAGAN_DP_LIB=.../libpagan_dp_stats.so): where a tiled job's time goes.
One full-matrix pair; prints tiles, prologue cycles per tile, steps and cycles per step by kind (simple / near / loops), the
flow waits.  Counters: dp_tiles.hip, PG_TILE_STATS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import bench
import pagan2_msa_amd as pg
from pagan2_msa_amd import host
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2_16x2kb_dna_full"
names, seqs, newick = bench.make_inputs(w)
cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = bench.WORKLOADS[w]
msa = host.Msa(names, seqs, newick, use_anchors=anchors).align()
for k in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, msa.n_internal - 1)):
    left, right, model, band = msa.node_job(k)
    b = pg.Batch([(left, right, model, band)])
    b.run(); b.sync()
    pg.lib().pagan_batch_debug_poison(b._h)       # counters (tail of the trace buffer) to all ones: they come back as x - 1
    b.run(); b.sync()
    nl_, nr_ = left.n_sites, right.n_sites
    if pg.debug_route(left, right, model, band)[1]:               # aligned on compacted graphs: the device's job has their sizes
        cp = pg.debug_compact(left, right, band)
        nl_, nr_ = len(cp["keep_left"]), len(cp["keep_right"])
    n = 3 * (nl_ - 1 + nr_ - 1)
    # multi-edge / dead site statistics of what the device aligns
    deg_l = np.diff(left.bwd_off); deg_r = np.diff(right.bwd_off)
    print("  graphs: sites %d %d (device %d %d), bwd degree 0/1/2/3+ left %s right %s" % (
        left.n_sites, right.n_sites, nl_, nr_, [int((deg_l == q).sum()) for q in (0, 1, 2)] + [int((deg_l >= 3).sum())],
        [int((deg_r == q).sum()) for q in (0, 1, 2)] + [int((deg_r >= 3).sum())]))
    raw = np.zeros(n, np.int32)
    pg.lib().pagan_batch_debug_trace(b._h, 0, raw.ctypes.data_as(C.c_void_p), raw.nbytes)
    at = (n - 64) & ~1
    c = raw[at:at + 40].view(np.uint64) + np.uint64(1)
    ms = b.last_ms()
    tiles = max(int(c[0]), 1)
    print("node", k, "sites", left.n_sites, right.n_sites, "fill %.2f ms" % ms[0])
    print("  tiles", tiles, "prologue cycles/tile %.0f" % (c[1] / tiles), "whole tile cycles/tile %.0f" % (c[8] / tiles))
    for q, name in enumerate(("simple", "near", "loops")):
        if c[2 + q]:
            print("  %-6s steps %9d  cycles/step %.0f" % (name, c[2 + q], c[5 + q] / c[2 + q]))
    if c[4]:
        print("  loops steps with a fetch from beyond the LDS window: %d (%.0f %%), such fetches one after the other per step (largest lane): %.2f; (left, right) pairs per loops step (largest lane): %.1f" %
              (c[15], 100.0 * c[15] / c[4], c[16] / max(int(c[15]), 1), c[17] / c[4]))
    print("  waits per tile: diagonals %.0f, neighbours %.0f, acquire+tile %.0f, release %.0f, lag wait %.0f, lag halo %.0f" %
          tuple(c[q] / tiles for q in (9, 10, 11, 12, 13, 14)))
    b.close()
