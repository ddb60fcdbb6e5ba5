import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import pagan2_msa_amd as pg
from pagan2_msa_amd import synth, abi
# Diagnostic: ~45 rows x 60000 columns -- every diagonal's cells are wave 0's; the pace of one compute wave with nobody to wait for
rng = np.random.default_rng(1)
left = synth.chain_graph(''.join(rng.choice(list('ACGT'), 56)))
right = synth.chain_graph(''.join(rng.choice(list('ACGT'), 60000)))
Lx, Ly = left.n_sites - 1, right.n_sites - 1
up = np.zeros(Lx, np.int32); lo_ = np.full(Lx, Ly - 1, np.int32)
up[Lx - 6:] = Ly - 30; lo_[:6] = 30                     # rows 6 .. Lx-7 run the whole length: the middle diagonals touch no edge of the matrix
band = abi.Band(up, lo_)
model = synth.random_model(15, 3)
print(pg.debug_route(left, right, model, band))
cls, waves = pg.debug_plan(left, right, band)
print("classes", np.bincount(cls, minlength=6).tolist(), "nd", len(cls))
batch = pg.Batch([(left, right, model, band)])
for rep in range(3):
    batch.run(); batch.sync()
    ms = batch.last_ms()
    print("ms", ms, "cycles/diag at 2.4 GHz: %.0f" % (ms[0] * 2.4e6 / len(cls)))
