"""CPU: the oracle against hand-derived values and its own invariants, and the C-ABI library's
export table (no compute calls: this container has no GPU)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from pagan2_msa_amd import abi, host, synth


def test_identical_sequences_closed_form(oracle):
    """All-match path: score = sum_i (double(2*ng) + s(a_i,a_i)) + ng, added in that order."""
    m = synth.jc_like_dna_model(0.1)
    s = "ACGTACGGTCATTGCA"
    g = synth.chain_graph(s)
    r = oracle.dp_align(g, g, m)
    ng = m.params[3]
    want = 0.0
    for c in s:
        k = synth.DNA_FULL.index(c)
        want = want + (float(np.float32(2) * ng) + float(m.log_score[k, k]))
    want = want + float(ng)
    assert r.score == want
    assert (r.cols[:, 2] == abi.MATCHED).all() and r.cols.shape[0] == len(s)
    assert r.end == (abi.M_MAT, len(s), len(s), len(s) + 1, len(s) + 1)
    assert np.array_equal(r.left_used, np.arange(1, len(s) + 2))


def test_single_gap_closed_form(oracle):
    """One deleted residue in the middle: the Viterbi path pays open + non-gap once, and the
    terminal rules do not apply (interior gap)."""
    m = synth.jc_like_dna_model(0.1)
    a, b = "ACGTTGCAACGT", "ACGTTCAACGT"      # G at index 5 deleted
    r = oracle.dp_align(synth.chain_graph(a), synth.chain_graph(b), m)
    states = r.cols[:, 2].tolist()
    assert states.count(abi.XGAPPED) == 1 and states.count(abi.MATCHED) == len(b)
    go, ge, gE, ng = [float(x) for x in m.params]
    assert go < ge < 0 and gE > ge
    # any all-match+one-gap path has the same score up to tie order; compare against brute force
    best = -math.inf
    for gap_at in range(len(a)):
        bb = a[:gap_at] + a[gap_at + 1:]
        if bb != b:
            continue
        sc, prev_gap = 0.0, False
        ok = True
        for i, c in enumerate(a):
            k = synth.DNA_FULL.index(c)
            if i == gap_at:
                sc = (sc + ng) + go
                prev_gap = True
            else:
                t = (float(np.float32(0.0) + np.float32(ng)) if prev_gap else float(np.float32(2) * np.float32(ng))) + float(m.log_score[k, k])
                sc = sc + t
                prev_gap = False
        sc = sc + ng
        best = max(best, sc)
    assert abs(r.score - best) < 1e-9


def test_full_width_band_equals_no_band(oracle):
    left = synth.random_graph(120, 15, 1, p_extra=0.4, p_dead=0.02)
    right = synth.random_graph(100, 15, 2, p_extra=0.4)
    m = synth.random_model(15, 3)
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    full = abi.Band(np.full(Lx, -3, np.int32), np.full(Lx, Ly + 7, np.int32))   # clamped like Tunnel_matrix
    assert oracle.dp_align(left, right, m, full).same_alignment(oracle.dp_align(left, right, m))


def test_band_excluding_start_corner_is_rejected(oracle):
    g = synth.chain_graph("ACGTACGT")
    band = abi.Band(np.full(9, 1, np.int32), np.full(9, 9, np.int32))
    with pytest.raises(RuntimeError):
        oracle.dp_align(g, g, synth.jc_like_dna_model(0.1), band)


def test_skip_columns_and_used_edges(oracle):
    """A left graph with a long edge over two sites that the right sequence lacks: the path takes
    the long edge and reports the jumped sites as xskipped columns."""
    st = np.array([-1, 0, 1, 2, 2, 3, 0, -1], np.int32)          # A C G G T A
    off = np.array([0, 0, 1, 2, 3, 4, 6, 7, 8], np.int32)
    src = np.array([0, 1, 2, 3, 4, 2, 5, 6], np.int32)           # site 5 (T) also reachable from site 2 (C)
    eid = np.array([1, 2, 3, 4, 5, 8, 6, 7], np.int32)
    left = abi.Graph(st, off, src, np.zeros(8, np.float32), eid, n_edges=9)
    right = synth.chain_graph("ACTA")
    r = oracle.dp_align(left, right, synth.jc_like_dna_model(0.1))
    assert r.cols[:, 2].tolist() == [2, 2, 5, 5, 2, 2]
    assert r.cols[:, 0].tolist() == [1, 2, 3, 4, 5, 6] and r.cols[:, 1].tolist() == [1, 2, -1, -1, 3, 4]
    assert 8 in r.left_used and 3 not in r.left_used and 4 not in r.left_used and 5 not in r.left_used


def test_library_exports_every_declared_symbol():
    """include/pagan_dp.h and include/pagan_host.h: every declared entry point resolves."""
    import re
    import pagan2_msa_amd as pg
    assert os.path.exists(pg.LIB_PATH), "libpagan_dp.so not built (python __graft_entry__.py)"
    lib = C.CDLL(pg.LIB_PATH)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for hdr in ("pagan_dp.h", "pagan_host.h"):
        text = open(os.path.join(root, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(pagan_[a-z0-9_]+)\s*\(", text))
    assert declared == set(abi.EXPORTED) | set(host.HOST_EXPORTED)
    for sym in sorted(declared):
        assert getattr(lib, sym) is not None
    lib.pagan_dp_version.restype = C.c_char_p
    assert b"gfx950" in lib.pagan_dp_version()


def test_no_gpu_means_loud_failure_not_cpu_fallback(pg):
    if pg.device_count() > 0:
        pytest.skip("GPU present")
    g = synth.chain_graph("ACGT")
    with pytest.raises(pg.PaganError) as e:
        pg.align(g, g, synth.jc_like_dna_model(0.1))
    assert e.value.code == abi.PAGAN_E_NODEVICE
    assert pg.lib().pagan_dp_count_cells(6, 6, None) == 25
    assert pg.lib().pagan_dp_predict_bytes(6, 6, None) >= 25 * 36
