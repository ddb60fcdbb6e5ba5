"""BASELINE.json's configurations at their full sizes: every internal node of the progressive alignment against the
oracle -- score bits, alignment columns (skip columns included), used child edges -- and, per node, which fill kernel
the library gave it (host-only planner, the same decision pagan_batch_create takes).

    cfg2  16 x 2 kb DNA, no anchors        15 full matrices        row strips on the banded kernel; tiles where a diagonal holds too many multi-edge sites
    cfg3  64 x 500 aa, WAG, no anchors     63 full matrices        tiles (211-state table)
    cfg4  32 x 100 kb DNA, prefix anchors  31 banded alignments    register wavefront (2e5-diagonal chains, 16-bit records)
    cfg5  512 x 10 kb DNA, prefix anchors  511 alignments, ALL of them checked (root: 1.05e9 cells, compacted, tiles; the
          oracle's ~150 s of single-thread work spread over the host's cores)

The workloads are bench.py's (same generator, same seeds)."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


def walk(pg, workload):
    from pagan2_msa_amd import host
    cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = bench.WORKLOADS[workload]
    names, seqs, newick = bench.make_inputs(workload)
    msa = host.Msa(names, seqs, newick, use_anchors=anchors).align()
    rows = msa.alignment()
    assert all(r.replace("-", "") == s for r, s in zip(rows, seqs)), "the alignment's rows are not the input sequences"
    return msa


def check_nodes(pg, oracle, msa, nodes, expect, threads=1):
    """expect(k, info, route, compacted, widest) -> None or a complaint.  threads > 1: the oracle's alignments (a C call each,
    outside the interpreter's lock) side by side on that many host threads."""
    nodes = list(nodes)

    def reference(k):
        left, right, model, band = msa.node_job(k)
        return oracle.dp_align(left, right, model, band)
    if threads > 1:
        # (the oracle keeps three matrices of 32-byte cells: the root of cfg5 alone is 100 GB -- alignments of more than 2e8
        #  cells run one at a time, before the pool starts)
        from concurrent.futures import ThreadPoolExecutor
        big = [k for k in nodes if msa.node_info(k).cells > 2 * 10 ** 8]
        wanted = {k: reference(k) for k in big}
        rest = [k for k in nodes if k not in wanted]
        with ThreadPoolExecutor(threads) as pool:
            wanted.update(zip(rest, pool.map(reference, rest)))
    else:
        wanted = None
    for k in nodes:
        left, right, model, band = msa.node_job(k)
        want = wanted[k] if wanted is not None else reference(k)
        got = msa.node_result(k)
        assert got.status == want.status, "node %d" % k
        assert np.float64(got.score).tobytes() == np.float64(want.score).tobytes(), "node %d: score %r != %r" % (k, got.score, want.score)
        assert got.same_alignment(want), "node %d (level %d): columns or used edges differ" % (k, msa.node_info(k).level)
        assert got.cells == want.cells
        route, compacted, widest = pg.debug_route(left, right, model, band)
        why = expect(k, msa.node_info(k), route, compacted, widest)
        assert why is None, "node %d: %s (route %s, compacted %s, widest diagonal %d)" % (k, why, route, compacted, widest)


def test_cfg2_16x2kb_dna_full_matrices(pg, oracle):
    msa = walk(pg, "cfg2_16x2kb_dna_full")
    assert msa.n_internal == 15
    seen = set()

    def expect(k, info, route, c, w):
        seen.add(route)
        return None if route in ("pg_fill_pipe (row strips)", "pg_fill_tiles_flow") else "expected row strips or the tiled kernel"

    check_nodes(pg, oracle, msa, range(15), expect)
    assert "pg_fill_pipe (row strips)" in seen, "the leaf pairs are meant to run as row strips: %s" % seen


def test_cfg3_64x500aa_protein_full_matrices(pg, oracle):
    msa = walk(pg, "cfg3_64x500aa_protein_full")
    assert msa.n_internal == 63
    check_nodes(pg, oracle, msa, range(63), lambda k, info, route, c, w: None if route == "pg_fill_tiles_flow" else "expected the tiled kernel")


def test_cfg4_32x100kb_dna_anchored(pg, oracle):
    msa = walk(pg, "cfg4_32x100kb_dna_anchored")
    assert msa.n_internal == 31
    seen = set()

    def expect(k, info, route, compacted, widest):
        left, right, model, band = msa.node_job(k)
        cls, _ = pg.debug_plan(left, right, band)
        seen.update(int(c) for c in np.unique(cls))
        assert cls.size > 190000, "a 2 x 100 kb alignment has about 2e5 anti-diagonals"
        return None if route == "pg_fill_pipe" else "expected the banded register-wavefront kernel"

    check_nodes(pg, oracle, msa, range(31), expect)
    assert seen >= {0, 1, 2, 3, 4}, "the tree is meant to reach every class of diagonal: %s" % sorted(seen)


def test_cfg5_512x10kb_dna_anchored_every_node(pg, oracle):
    import os
    msa = walk(pg, "cfg5_512x10kb_dna_anchored")
    assert msa.n_internal == 511
    infos = [msa.node_info(k) for k in range(511)]
    top = max(i.level for i in infos)
    routes = {}

    def expect(k, info, route, compacted, widest):
        routes.setdefault(info.level, set()).add((route, compacted))
        if info.level == top:
            if info.cells < 10 ** 9:
                return "the root is meant to have more than 1e9 cells, has %d" % info.cells
            if route != "pg_fill_tiles_flow" or not compacted:
                return "the root is meant to run on the tiled kernel with its dead sites taken out"
        return None

    # largest first: the root's 1e9 cells are a fifth of the oracle's work and should not be what the pool ends on
    order = sorted(range(511), key=lambda k: -infos[k].cells)
    check_nodes(pg, oracle, msa, order, expect, threads=max(1, min(12, (os.cpu_count() or 2) - 1)))
    assert any(r == "pg_fill_pipe" for rs in routes.values() for r, _ in rs), "the lower levels are banded alignments: %s" % routes
    assert any(r == "pg_fill_pipe (row strips)" for rs in routes.values() for r, _ in rs), "the first wide levels are meant to run as row strips: %s" % routes
    assert any(r == "pg_fill_tiles_flow" for rs in routes.values() for r, _ in rs)
