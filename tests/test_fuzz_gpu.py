"""GPU fuzz: many small random alignments (graph shapes, bands with empty rows and wide boxes,
option bits, table sizes) in batched launches, each compared bit for bit with the oracle."""
import numpy as np
import pytest

from pagan2_msa_amd import abi, synth

pytestmark = pytest.mark.gpu


def random_band(rng, Lx, Ly):
    kind = rng.integers(0, 4)
    if kind == 0:
        return None
    centre = np.linspace(0, max(Ly - 1, 0), Lx)
    wl, wh = (1, 6) if kind == 1 else (3, 40)
    up = np.maximum.accumulate(np.clip(centre - rng.integers(wl, wh, Lx), 0, None)).astype(np.int32)
    lo = np.maximum.accumulate(np.clip(centre + rng.integers(wl, wh, Lx), 0, Ly + 3)).astype(np.int32)
    up[0] = 0
    if kind == 3 and Lx > 8:                       # a stretch of empty rows (upper > lower)
        a = int(rng.integers(2, Lx - 3))
        b = min(Lx, a + int(rng.integers(1, 4)))
        up[a:b] = np.minimum(lo[a:b] + 1, np.maximum(up[b - 1], lo[a]) + 1)
        up = np.maximum.accumulate(up)
    return abi.Band(up, lo)


def make_case(rng, k):
    S = int(rng.choice([2, 4, 15, 16, 17, 40]))
    n1, n2 = int(rng.integers(0, 90)), int(rng.integers(0, 90))
    if n1 + n2 == 0:
        n1 = 1
    left = synth.random_graph(n1, S, 5000 + 2 * k, p_extra=float(rng.choice([0.0, 0.2, 0.6])), max_deg=int(rng.integers(2, 6)),
                              max_span=int(rng.integers(1, 30)), p_dead=float(rng.choice([0.0, 0.0, 0.02])))
    right = synth.random_graph(n2, S, 5001 + 2 * k, p_extra=float(rng.choice([0.0, 0.2, 0.6])), max_deg=int(rng.integers(2, 6)),
                               max_span=int(rng.integers(1, 30)), p_dead=float(rng.choice([0.0, 0.0, 0.02])))
    band = random_band(rng, left.n_sites - 1, right.n_sites - 1)
    return left, right, synth.random_model(S, k, dist=float(rng.choice([0.002, 0.1, 0.4]))), band


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_fuzz_small_alignments(pg, oracle, flags):
    rng = np.random.default_rng(100 + flags)
    jobs = [make_case(rng, 300 * flags + k) for k in range(120)]
    want = [oracle.dp_align(l, r, m, b, flags=flags) for l, r, m, b in jobs]
    got = pg.align_batch(jobs, flags=flags)
    n_unreach = 0
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.same_alignment(w), "flags %d case %d (sizes %d x %d, band %s)" % (
            flags, k, jobs[k][0].n_sites, jobs[k][1].n_sites, "yes" if jobs[k][3] is not None else "no")
        assert g.cells == w.cells
        n_unreach += g.status
    assert n_unreach < 90                              # most cases must exercise a real traceback
